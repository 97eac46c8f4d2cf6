// Geometry of the weight-gradient kernel (csrc/conv_wgrad.hip): every index formula the kernel uses -- LDS images, LDS-DMA slot
// decode, operand addresses, accumulator-tile -> weight-element map, K parts -- as plain constexpr / inline C++, shared with the
// lane-level host model (oracle/wgrad_model.cpp, test infrastructure) that replays the kernel's data movement on the CPU.
#pragma once

namespace fn2 {
namespace wg {

constexpr int cdiv_c(int a, int b) { return (a + b - 1) / b; }
constexpr int up_mod_c(int v, int r, int m) { return v + ((r - v % m) + m) % m; }   // smallest >= v with == r (mod m)

// KS x KS taps, stride S; wave tile = MA groups of 16 `a` channels x NB groups of 16 `b` channels x all KS*KS taps; workgroup =
// WM x WN waves; chunk = R rows x XT pixels of the `a` map (one sample).  Staging is 16-byte LDS-DMA (V = 4 dwords per lane): both maps
// must have a width that is a multiple of 4 (the host pads narrower / odd maps into a workspace first, conv_wgrad.hip: pad_width).
// DB = 1: two staging buffers (the next chunk's DMA flies under this chunk's MFMAs), one workgroup per CU; DB = 0: one buffer and two
// workgroups per CU (8 waves: the other workgroup's MFMAs cover this one's DMA latency, barriers and issue stalls).
template <int KS_, int S_, int MA_, int NB_, int WM_, int WN_, int XT_, int R_, int DB_ = 1>
struct Cfg {
  static constexpr int KS = KS_, S = S_, MA = MA_, NB = NB_, WM = WM_, WN = WN_, XT = XT_, R = R_, V = 4, DB = DB_;
  static constexpr int NW = WM * WN, THREADS = 64 * NW;
  static constexpr int T = KS * KS;                          // taps
  static constexpr int CA = 16 * MA * WM, CB = 16 * NB * WN; // channels of a workgroup's block
  static constexpr int TILES = MA * NB * T;                  // accumulator tiles per wave
  static constexpr int KSTEPS = R * XT / 4;                  // MFMA k-steps (4 pixels) per chunk
  // `a` image: [CA][R][XT], channel stride CSA
  static constexpr int ROWA = R * XT;
  static constexpr int CSA = up_mod_c(ROWA, 4, 32);        // == 4 (mod 32): the A-operand read (16 channels x 2 pixels per half wave) is 2-way
  static constexpr int SLOTS_CA = CSA / V;
  static constexpr int SLOTS_A = CA * SLOTS_CA;
  static constexpr int NRUN_A = cdiv_c(SLOTS_A, 64);
  static constexpr int A_DW = NRUN_A * 64 * V;
  // `b` window: [CB][WR][RSB]; window column wc <-> global column S * x0 - PADL + wc, window row wr <-> global row S * y0 - pad + wr
  static constexpr int PADL = 4;
  static constexpr int WR = S * (R - 1) + KS;
  static constexpr int WCOLS = S * (XT - 1) + KS + PADL;
  static constexpr int RSB = cdiv_c(WCOLS, 4) * 4;
  static constexpr int CSB = up_mod_c(WR * RSB, 4, 32);
  static constexpr int SLOTS_CB = CSB / V;
  static constexpr int SLOTS_B = CB * SLOTS_CB;
  static constexpr int NRUN_B = cdiv_c(SLOTS_B, 64);
  static constexpr int B_DW = NRUN_B * 64 * V;
  static constexpr int BUF = A_DW + B_DW;                    // dwords per staging buffer (two of them)
  static constexpr int RPW_A = cdiv_c(NRUN_A, NW), RPW_B = cdiv_c(NRUN_B, NW);
  static constexpr int LDS_BYTES = (DB ? 2 : 1) * BUF * 4;
  static constexpr int WG_PER_CU = DB ? 1 : 2;
  static_assert(XT % 4 == 0 && XT >= 4, "chunk width: whole k-steps");
  static_assert(NW == 4, "256 threads");
  static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS");
  static_assert(CSA % V == 0 && CSB % V == 0, "whole slots per channel");
};

// ---- LDS-DMA slot decode.  Slot s of an image = V consecutive dwords at LDS dword address image_base + s * V.
struct SlotA { int ch, r, x; bool in_image; };      // channel of the block, row of the chunk, first pixel column of the chunk
template <class K>
constexpr SlotA slot_a(int s) {
  const int ch = s / K::SLOTS_CA, d = (s % K::SLOTS_CA) * K::V;
  return SlotA{ch, d / K::XT, d % K::XT, s < K::SLOTS_A && d < K::ROWA};
}
struct SlotB { int ch, wr, wc; bool in_image; };    // channel of the block, window row, first window column
template <class K>
constexpr SlotB slot_b(int s) {
  const int ch = s / K::SLOTS_CB, d = (s % K::SLOTS_CB) * K::V;
  return SlotB{ch, d / K::RSB, d % K::RSB, s < K::SLOTS_B && d < K::WR * K::RSB};
}

// ---- MFMA operand addresses (dwords inside a staging buffer).  Lane l: m = n = l & 15, k = l >> 4 (pixel 4 * xq + k of the k-step).
template <class K>
constexpr int a_lane_base(int wm, int lane) { return ((wm * K::MA) * 16 + (lane & 15)) * K::CSA + (lane >> 4); }
template <class K>
constexpr int a_step_off(int ma, int r, int xq) { return ma * 16 * K::CSA + r * K::XT + 4 * xq; }
template <class K>
constexpr int b_lane_base(int wn, int lane, int pad) {
  return K::A_DW + ((wn * K::NB) * 16 + (lane & 15)) * K::CSB + K::S * (lane >> 4) + K::PADL - pad;
}
template <class K>
constexpr int b_step_off(int nb, int r, int xq, int ky, int kx) { return nb * 16 * K::CSB + (K::S * r + ky) * K::RSB + K::S * 4 * xq + kx; }

// ---- accumulator tile (ma, nb, tap) of wave (wm, wn): lane l, register j holds D[m = 4 * (l >> 4) + j][n = l & 15]
template <class K> constexpr int tile_index(int ma, int nb, int t) { return (ma * K::NB + nb) * K::T + t; }
template <class K> constexpr int wave_index(int wm, int wn) { return wm + K::WM * wn; }

// Runtime mirror of the constants the finalize pass needs (which partial-sum element holds weight element (ca, cb, tap)).
struct SlabMap {
  int MA, NB, WM, WN, T, CA, CB, TILES;
  // floats from the start of one part's slab to the element, given the channel-block counts
  inline long long offset(int ca, int cb, int t, int nblk_b) const {
    const int blk = (ca / CA) * nblk_b + cb / CB;
    const int ra = ca % CA, rb = cb % CB;
    const int wm = ra / (16 * MA), ma = (ra / 16) % MA, m = ra % 16;
    const int wn = rb / (16 * NB), nb = (rb / 16) % NB, n = rb % 16;
    const int wave = wm + WM * wn, tile = (ma * NB + nb) * T + t, lane = (m / 4) * 16 + n, reg = m % 4;
    return (((long long)blk * 4 + wave) * TILES + tile) * 256 + lane * 4 + reg;
  }
};
template <class K>
constexpr SlabMap slab_map() { return SlabMap{K::MA, K::NB, K::WM, K::WN, K::T, K::CA, K::CB, K::TILES}; }

// ---- K parts: part p of `ksplit` covers the rows (sample n, row y of the `a` map, flattened u = n * Ha + y) [p * U / ksplit,
// (p + 1) * U / ksplit), U = N * Ha.  A function of the layer geometry only (never of the tile variant): it fixes the summation
// order -- per part a k-ordered fma chain over the pixels in (n, y, x) order, the parts added in part order.
constexpr int part_begin(int p, int ksplit, int U) { return (int)((long long)p * U / ksplit); }

// Channel block of a workgroup per tap class -- the wave tiles of csrc/conv_wgrad.hip's variants: (a channels, b channels).
inline void class_block(int KS, int& ca, int& cb) {
  if (KS == 1) { ca = 32; cb = 256; }
  else if (KS == 3) { ca = 64; cb = 64; }
  else { ca = 64; cb = 32; }                                   // 4x4, 5x5, 7x7
}

// Canonical K split of a layer (a function of its geometry only): the k that minimises a two-term cost model over the channel blocks
// of its tap class -- whole rounds of 256 workgroups (one per CU) at 120 TFLOP/s, plus writing and re-reading k slabs of partial sums
// at 3 TB/s -- with at least 2 rows per part and at most 64 parts; ties go to the smaller k.
inline int ksplit_for(int N, int Ca, int Ha, int Wa, int Cb, int KS) {
  int ca, cb;
  class_block(KS, ca, cb);
  const long long blocks = (long long)cdiv_c(Ca, ca) * cdiv_c(Cb, cb);
  const int U = N * Ha;
  int kmax = U / 2 < 64 ? U / 2 : 64;
  if (kmax < 1) kmax = 1;
  const double flops = 2.0 * N * Ha * Wa * (double)(blocks * ca) * cb * KS * KS;       // incl. the padding of the last blocks
  const double t_compute = flops / 120e12, t_slab = 2.0 * 4.0 * (double)(blocks * ca) * cb * KS * KS / 3e12;
  int best = 1;
  double best_t = 0;
  for (int k = 1; k <= kmax; ++k) {
    const long long wgs = blocks * k, rounds = (wgs + 255) / 256;
    const double t = (double)rounds * (t_compute * 256.0 / (double)wgs) + (double)k * t_slab;
    if (k == 1 || t < best_t * (1.0 - 1e-9)) { best = k; best_t = t; }
  }
  return best;
}

}  // namespace wg
}  // namespace fn2
