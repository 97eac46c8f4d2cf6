"""The host side of the third-generation correlation forward (csrc/correlation_units.hip) on the CPU: the task tables a launch gets
(column tasks, unit lists, absent units, live / dead combinations, XCD striping) are walked exactly as the kernel's scalar decode walks
them, and every element of top[n, (q, o), y, x] must be written EXACTLY once -- by a unit's accumulator, by an absent unit's zeros, or by
a zero-fill task -- with nothing stored that no unit wrote.  No compute, no GPU: the library only has to load."""
import ctypes as C

import numpy as np
import pytest

from flownet2_amd import _lib

R, D, NBT = 10, 21, 6
MAXSEG, SEGW, MAXCOMBO, HEAD = 12, 16, 1024, 15


def plan(N, H, W, policy):
    L = _lib.lib()
    n = L.fn2_debug_correlation_units_plan(N, H, W, policy, None, 0)
    if n == 0:
        return None
    buf = (C.c_uint * n)()
    assert L.fn2_debug_correlation_units_plan(N, H, W, policy, buf, n) == n
    w = np.frombuffer(buf, dtype=np.uint32).copy()
    head = w[:HEAD].view(np.int32)
    a = dict(N=int(head[0]), H=int(head[2]), W=int(head[3]), TH=int(head[4]), TD=int(head[5]), LP=int(head[6]), DP=int(head[7]), G=int(head[8]),
             nseg=int(head[13]))
    a["seg"] = w[HEAD:HEAD + MAXSEG * SEGW].reshape(MAXSEG, SEGW)
    a["combo"] = w[HEAD + MAXSEG * SEGW:HEAD + MAXSEG * SEGW + MAXCOMBO // 2]
    a["grid"] = int(w[-1])
    return a


def unit_pattern():
    """(rowid, x within the patch) -> written, for one unit with b: [b][16 * D][8]"""
    pat = np.zeros((NBT, 16 * D, 8), np.int32)
    for b in range(NBT):
        for mi in range(4):
            for ni in range(4):
                for nj in range(4):
                    for r in range(4):
                        oo = 4 * b + nj - r
                        if 0 <= oo < D:
                            pat[b, (mi * 4 + ni) * D + oo, 2 * r:2 * r + 2] += 1
    return pat


PAT = unit_pattern()


def walk(a):
    """Every block of the grid, decoded as the kernel decodes it; returns the write count of every output element and the units per wave."""
    N, H, W = a["N"], a["H"], a["W"]
    cnt = np.zeros((N, D, D, H, W), np.int32)
    units_per_wave = []
    oo_of = np.arange(16 * D) % D
    for blk in range(a["grid"]):
        xcd, j = blk & 7, blk >> 3
        live = j < a["LP"]
        jj = j if live else j - a["LP"]
        per = a["TH"] if live else a["TD"]
        if a["G"] > 0:
            n, tt = xcd // a["G"], xcd % a["G"] + a["G"] * jj
            if tt >= per:
                continue
        else:
            t = xcd * (a["LP"] if live else a["DP"]) + jj
            if t >= N * per:
                continue
            n, tt = t // per, t % per
        ci = (0 if live else a["TH"]) + tt
        cb = (int(a["combo"][ci >> 1]) >> (16 * (ci & 1))) & 0xffff
        py, I, aa, segI = cb & 1, (cb >> 1) & 31, (cb >> 6) & 7, (cb >> 9) & 15
        assert segI < a["nseg"]
        sw = [int(v) for v in a["seg"][segI]]
        p0, npp, s0, nb = sw[0] & 255, (sw[0] >> 8) & 255, (sw[0] >> 16) & 255, sw[0] >> 24
        na, dp0, dnp = (sw[4] >> 16) & 15, (sw[4] >> 20) & 15, (sw[4] >> 24) & 15
        assert nb % 2 == 1 and na % 2 == 1 and na >= npp and 1 <= npp <= 4 and na + nb <= 12
        assert (sw[3] & 0xffff) == 65536 // (2 * nb) + 1 and (sw[4] & 0xffff) == 65536 // (2 * na) + 1
        written = np.zeros((16 * D, 8 * npp), np.int32)
        own = np.zeros((16 * D, 8 * npp), bool)            # the elements this task stores
        if live:
            nus = [(sw[2] >> (8 * wv)) & 255 for wv in range(4)]
            u0s = [(sw[1] >> (8 * wv)) & 255 for wv in range(4)]
            assert u0s[0] == 0 and all(u0s[k + 1] == u0s[k] + nus[k] for k in range(3)) and max(nus) <= 5 and max(nus) - min(nus) <= 1
            units_per_wave.append(nus)
            for k in range(sum(nus)):
                byte = (sw[6 + (k >> 2)] >> (8 * (k & 3))) & 255
                pl, sl = byte >> 4, byte & 15
                assert pl < npp and sl < nb
                b = sl + s0 - (pl + p0)
                assert 0 <= b < NBT
                br = (sw[5] >> (6 * pl)) & 63
                assert (br & 7) <= b <= (br >> 3), "a unit outside the b range its task owns"
                written[:, 8 * pl:8 * pl + 8] += PAT[b]
            for k in range(sw[3] >> 16):
                byte = (sw[11 + (k >> 2)] >> (8 * (k & 3))) & 255
                written[:, 8 * (byte >> 4):8 * (byte >> 4) + 8] += PAT[byte & 15]
            for pl in range(npp):
                br = (sw[5] >> (6 * pl)) & 63
                blo, bhi = br & 7, br >> 3
                for xl in range(8):
                    q = (oo_of + (xl >> 1)) >> 2
                    own[:, 8 * pl + xl] = (q >= blo) & (q <= bhi)
            assert (written[own] == 1).all(), "a stored element was written %s times" % set(written[own].tolist())
            xs = [(8 * p0 + xl, xl) for xl in range(8 * npp)]
        else:
            assert dnp >= 1
            own[:, 8 * dp0:8 * (dp0 + dnp)] = True
            xs = [(8 * p0 + xl, xl) for xl in range(8 * dp0, 8 * (dp0 + dnp))]
        for rowid in range(16 * D):
            blk_, oo = divmod(rowid, D)
            rmi, rni = blk_ >> 2, blk_ & 3
            qq, y = 4 * aa + rni - rmi, 2 * (4 * I + rmi) + py
            if not (0 <= qq < D and y < H):
                continue
            for x, xl in xs:
                if x < W and own[rowid, xl]:
                    cnt[n, qq, oo, y, x] += 1
    return cnt, units_per_wave


@pytest.mark.parametrize("shape", [(8, 40, 56), (4, 48, 96), (1, 56, 128), (2, 16, 24), (3, 11, 20), (1, 5, 8), (16, 9, 12), (1, 24, 192)])
@pytest.mark.parametrize("policy", [0, 3, 16, 7])
def test_every_output_element_is_written_exactly_once(shape, policy):
    a = plan(*shape, policy)
    if a is None:
        pytest.skip("no unit plan for this geometry (corr_fwd_pair serves it)")
    cnt, upw = walk(a)
    assert cnt.min() == 1 and cnt.max() == 1
    assert upw, "no live task"


def test_flownetc_shape_is_one_round_of_equal_tasks():
    """[8,256,40,56] (BASELINE config 2): an image row's 36 units are cut into two tasks of 18 (patch 3 is shared: b 0 .. 2 / b 3 .. 5), dealt
    4-4-5-5 to the consumer waves; 96 live tasks per sample = 768 per launch = exactly three per CU; a sample per XCD."""
    a = plan(8, 40, 56, 0)
    assert a is not None and a["nseg"] == 2 and a["G"] == 1 and a["TH"] == 96 and a["LP"] == 96
    _, upw = walk(a)
    assert all(sorted(u) == [4, 4, 5, 5] for u in upw) and len(upw) == 768

