"""CustomData sample format: Datum wire format, the writer's packing, the decode.

CPU part: the host functions of libflownet2_hip.so and the C oracle against (a) the protobuf runtime installed here
(google.protobuf, the reference's third-party dependency for Datum), (b) an independent numpy statement of the packing.
GPU part (-m gpu): the decode kernels against the oracle, bit for bit, through the C ABI."""
import numpy as np
import pytest
import torch

import flownet2_amd
import oracle
from flownet2_amd import sample_format as SF

ENC = SF.FLOW_SAMPLE_ENCODINGS
SP = SF.FLOW_SAMPLE_SLICE_POINTS


def _datum_class(extra_field=False):
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    T = descriptor_pb2.FieldDescriptorProto
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name = "fn2_datum_%d.proto" % extra_field
    fdp.package = "caffe%d" % extra_field
    fdp.syntax = "proto2"
    m = fdp.message_type.add()
    m.name = "Datum"                                    # src/caffe/proto/caffe.proto:30-41
    for name, num, typ, label in [("channels", 1, T.TYPE_INT32, 1), ("height", 2, T.TYPE_INT32, 1), ("width", 3, T.TYPE_INT32, 1),
                                  ("data", 4, T.TYPE_BYTES, 1), ("label", 5, T.TYPE_INT32, 1), ("float_data", 6, T.TYPE_FLOAT, 3),
                                  ("encoded", 7, T.TYPE_BOOL, 1)]:
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, num, typ, label
    if extra_field:                                     # fields a newer writer might add: must be skipped
        for name, num, typ in [("extra_str", 9, T.TYPE_STRING), ("extra_f64", 10, T.TYPE_DOUBLE), ("extra_f32", 11, T.TYPE_FIXED32),
                               ("extra_i", 300, T.TYPE_INT64)]:
            f = m.field.add()
            f.name, f.number, f.type, f.label = name, num, typ, 1
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    desc = pool.FindMessageTypeByName(fdp.package + ".Datum")
    try:
        return message_factory.GetMessageClass(desc)
    except AttributeError:
        return message_factory.MessageFactory(pool).GetPrototype(desc)


def _sample_inputs(H, W, seed, nan_frac=0.05):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    f = (rng.standard_normal((2, H, W)) * 12).astype(np.float32)
    f[rng.random((2, H, W)) < nan_frac] = np.nan
    o = rng.random((H, W)) < 0.3
    return a, b, f, o


def _numpy_pack(a, b, f, o):
    """Independent statement of tools/convert_imageset_and_flow.cpp:142-206."""
    H, W = a.shape[:2]
    q = np.where(np.isnan(f), 32767, np.trunc(np.nan_to_num(f) * np.float32(32))).astype("<i2")
    bits = np.packbits(np.asarray(o, bool).reshape(-1), bitorder="little")
    return a.transpose(2, 0, 1).tobytes() + b.transpose(2, 0, 1).tobytes() + q.tobytes() + bits.tobytes()


def _numpy_unpack(sample, H, W, mean=None, scale=1.0):
    hw = H * W
    u = np.frombuffer(sample, np.uint8)
    img0 = u[:3 * hw].reshape(3, H, W).astype(np.float32)
    img1 = u[3 * hw:6 * hw].reshape(3, H, W).astype(np.float32)
    q = u[6 * hw:10 * hw].view("<i2").reshape(2, H, W)
    flow = np.where(q == 32767, np.float32(np.nan), q.astype(np.float32) / np.float32(32))
    occ = np.unpackbits(u[10 * hw:10 * hw + (hw - 1) // 8 + 1], bitorder="little")[:hw].reshape(1, H, W).astype(np.float32)
    tops = [img0, img1, flow, occ]
    if mean is not None:
        m = np.asarray(mean, np.float32).reshape(9, H, W)
        tops = [(t - m[c0:c1]) for t, (c0, c1) in zip(tops, [(0, 3), (3, 6), (6, 8), (8, 9)])]
    return [(t * np.float32(scale)).astype(np.float32) for t in tops]


SIZES = [(5, 7), (1, 1), (1, 9), (3, 8), (16, 24), (7, 13)]


# ---------------------------------------------------------------------------------------------------------
# Datum wire format against the protobuf runtime
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(9, 5, 7, 7, 355), (9, 384, 512, 0, 20000), (3, 1, 1, -1, 0), (1, 2 ** 20, 3, 2 ** 31 - 1, 127),
                                  (9, 16, 16, -2 ** 31, 128), (2, 300, 200, 12345, 16384)])
def test_serialize_is_byte_identical_to_protobuf(case):
    channels, height, width, label, nbytes = case
    data = np.random.default_rng(nbytes).integers(0, 256, nbytes, dtype=np.uint8).tobytes()
    d = _datum_class()()
    d.channels, d.height, d.width, d.label, d.data = channels, height, width, label, data
    want = d.SerializeToString()
    assert SF.serialize_datum(channels, height, width, data, label) == want
    assert oracle.datum_serialize(channels, height, width, data, label) == want
    back = _datum_class()()
    back.ParseFromString(SF.serialize_datum(channels, height, width, data, label))
    assert (back.channels, back.height, back.width, back.label, back.data) == (channels, height, width, label, data)


def _check_parsed(rec, want):
    got = SF.parse_datum(rec)
    ora = oracle.datum_parse(rec)
    for k in ("channels", "height", "width", "label", "encoded"):
        assert getattr(got, k) == want[k] == ora[k], k
    assert got.data == want["data"] == ora["data"]
    for fl in (got.float_data, ora["float_data"]):
        if want["float_data"] is None:
            assert fl is None
        else:
            assert np.array_equal(fl.view(np.uint32), np.asarray(want["float_data"], np.float32).view(np.uint32))


def test_parse_reads_what_protobuf_writes_including_float_data_unknown_fields_and_repeats():
    D, DX = _datum_class(), _datum_class(True)
    d = D()
    d.channels, d.height, d.width, d.label, d.encoded = 2, 3, 4, -5, True
    fl = np.random.default_rng(1).standard_normal(24).astype(np.float32)
    d.float_data.extend(fl.tolist())
    _check_parsed(d.SerializeToString(), dict(channels=2, height=3, width=4, label=-5, encoded=True, data=None, float_data=fl))
    # non-packed float_data written by hand (tag 0x35 per element) mixed with a packed run (tag 0x32): both are legal for proto2 readers
    hand = b"\x08\x02" + b"".join(b"\x35" + fl[i:i + 1].tobytes() for i in range(3)) + b"\x32\x08" + fl[3:5].tobytes()
    _check_parsed(hand, dict(channels=2, height=0, width=0, label=0, encoded=False, data=None, float_data=fl[:5]))
    # unknown fields of every wire type are skipped
    x = DX()
    x.channels, x.height, x.width, x.data, x.label = 9, 6, 10, b"\x00\x01\xfe\xff" * 50, 3
    x.extra_str, x.extra_f64, x.extra_f32, x.extra_i = "hello", 2.5, 77, -9
    _check_parsed(x.SerializeToString(), dict(channels=9, height=6, width=10, label=3, encoded=False, data=b"\x00\x01\xfe\xff" * 50, float_data=None))
    # proto2: fields in any order, the last occurrence of a scalar wins
    a, b = D(), D()
    a.channels, a.label = 1, 1
    b.channels, b.data = 4, b"xyz"
    _check_parsed(a.SerializeToString() + b.SerializeToString(), dict(channels=4, height=0, width=0, label=1, encoded=False, data=b"xyz", float_data=None))
    _check_parsed(b"", dict(channels=0, height=0, width=0, label=0, encoded=False, data=None, float_data=None))


def test_truncated_and_malformed_records_fail_exactly_where_protobuf_fails():
    from google.protobuf.message import DecodeError
    D = _datum_class()
    d = D()
    d.channels, d.height, d.width, d.label, d.data = 9, 300, 200, 1000, bytes(range(200))
    d.float_data.extend([1.0, 2.0])
    rec = d.SerializeToString()
    n_bad = 0
    for cut in range(len(rec) + 1):
        part = rec[:cut]
        try:
            D().ParseFromString(part)
            ok = True
        except DecodeError:
            ok = False
        n_bad += not ok
        if ok:
            SF.parse_datum(part)
            oracle.datum_parse(part)
        else:
            with pytest.raises(flownet2_amd.Fn2Error):
                SF.parse_datum(part)
            with pytest.raises(ValueError):
                oracle.datum_parse(part)
    assert n_bad > 200                                    # every cut inside the data bytes or inside a varint / fixed32
    for bad in [b"\x00\x01", b"\x0b", b"\x0c", b"\x0e\x00", b"\x32\x03abc", b"\x08" + b"\xff" * 11]:   # field 0, open / stray group, wire type 6, ragged packed floats, endless varint
        with pytest.raises(flownet2_amd.Fn2Error):
            SF.parse_datum(bad)
        with pytest.raises(ValueError):
            oracle.datum_parse(bad)


def _protobuf_view(D, rec):
    from google.protobuf.message import DecodeError
    try:
        d = D()
        d.ParseFromString(rec)
    except DecodeError:
        return None
    return (d.channels, d.height, d.width, d.label, bool(d.encoded), d.data if d.HasField("data") else None,
            np.asarray(list(d.float_data), np.float32).view(np.uint32).tolist())


def _our_views(rec):
    out = []
    try:
        g = SF.parse_datum(rec)
        out.append((g.channels, g.height, g.width, g.label, g.encoded, g.data, [] if g.float_data is None else g.float_data.view(np.uint32).tolist()))
    except flownet2_amd.Fn2Error:
        out.append(None)
    try:
        o = oracle.datum_parse(rec)
        out.append((o["channels"], o["height"], o["width"], o["label"], o["encoded"], o["data"], [] if o["float_data"] is None else o["float_data"].view(np.uint32).tolist()))
    except ValueError:
        out.append(None)
    return out


def test_edge_cases_of_the_wire_format_follow_protobuf():
    """Groups (skipped when well formed), 10-byte varints, int32 truncation, keys beyond 32 bits, known fields with a foreign wire type
    (treated as unknown), bool from any non-zero varint."""
    D = _datum_class()
    cases = [b"\x0b\x08\x01\x0c", b"\x0b\x08\x01", b"\x0b\x14", b"\x0c", b"\x0b\x13\x2a\x02hi\x14\x0c\x08\x07", b"\x0b\x13\x0c\x14",
             b"\x08" + b"\xff" * 9 + b"\x01", b"\x08" + b"\xff" * 9 + b"\x7f", b"\x08" + b"\xff" * 10 + b"\x01", b"\x08\x85\x80\x80\x80\x10",
             b"\x0a\x01\x05", b"\x20\x05", b"\x31" + b"\x00" * 8, b"\x2d\x01\x00\x00\x00", b"\xf8\xff\xff\xff\x0f\x01", b"\xf8\xff\xff\xff\x1f\x01",
             b"\x38\x02", b"\x22\x00", b"\x22\x03abc\x22\x02xy", b"\x0b" * 101 + b"\x0c" * 101, b"\x0b" * 99 + b"\x0c" * 99]
    for rec in cases:
        want = _protobuf_view(D, rec)
        for got in _our_views(rec):
            assert got == want, (rec, got, want)


def test_fuzzed_records_parse_like_protobuf():
    """20,000 byte strings: random bytes biased towards plausible keys, and mutations (bit flips, cuts, splices) of valid Datums."""
    D = _datum_class()
    rng = np.random.default_rng(2024)
    d = D()
    d.channels, d.height, d.width, d.label, d.data, d.encoded = 9, 24, 32, -3, bytes(range(40)), True
    d.float_data.extend([1.5, -2.25])
    valid = d.SerializeToString()
    keys = np.array([0x08, 0x10, 0x18, 0x22, 0x28, 0x32, 0x35, 0x38, 0x0b, 0x0c, 0x09, 0x0d, 0x00, 0x0e, 0x4a, 0xf8], np.uint8)
    accepted = 0
    for it in range(20000):
        kind = it % 4
        if kind == 0:
            rec = rng.integers(0, 256, int(rng.integers(0, 24)), dtype=np.uint8)
            pos = rng.random(rec.size) < 0.3
            rec[pos] = rng.choice(keys, int(pos.sum()))
        else:
            rec = np.frombuffer(valid, np.uint8).copy()
            if kind == 1:
                for _ in range(int(rng.integers(1, 4))):
                    rec[int(rng.integers(0, rec.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
            elif kind == 2:
                a, b = sorted(rng.integers(0, rec.size + 1, 2))
                rec = np.concatenate([rec[:a], rec[b:]])
            else:
                a = int(rng.integers(0, rec.size + 1))
                rec = np.concatenate([rec[:a], rng.choice(keys, int(rng.integers(1, 4))), rec[a:]])
        rec = rec.tobytes()
        want = _protobuf_view(D, rec)
        accepted += want is not None
        for got in _our_views(rec):
            assert got == want, (rec, got, want)
    assert 2000 < accepted < 19000


# ---------------------------------------------------------------------------------------------------------
# The writer's packing and the decode, on the host
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", SIZES)
def test_encode_sample_matches_the_numpy_statement_and_the_oracle(size):
    H, W = size
    a, b, f, o = _sample_inputs(H, W, H * 100 + W)
    want = _numpy_pack(a, b, f, o)
    assert len(want) == 10 * H * W + (H * W - 1) // 8 + 1 == SF.sample_bytes(9, H, W, SP, ENC) == oracle.custom_data_sample_bytes(9, H, W, SP, ENC)
    assert SF.encode_sample(a, b, f, o) == want
    assert oracle.custom_data_encode_sample(a, b, f, o) == want
    # no flow / no occlusions: zeros (tool :171, :190)
    z = _numpy_pack(a, b, np.zeros((2, H, W), np.float32), np.zeros((H, W), bool))
    assert SF.encode_sample(a, b) == z == oracle.custom_data_encode_sample(a, b)


def test_encode_quantisation_edges():
    """(short)(flow * 32): toward zero; SHRT_MAX is the NaN marker, so flows in [1023.96875, 1024) collide with it; beyond the
    range of short the reference is undefined -- saturated here."""
    H, W = 1, 8
    a = np.zeros((H, W, 3), np.uint8)
    f = np.zeros((2, H, W), np.float32)
    f[0, 0] = [0.03, -0.03, 0.04, -0.04, 1023.9, 1023.97, 5000.0, -5000.0]
    f[1, 0] = [np.nan, 1.0, -1.0, 0.5 / 32, -0.5 / 32, 1.5 / 32, -1024.0, 1e30]
    for enc in (SF.encode_sample, oracle.custom_data_encode_sample):
        q = np.frombuffer(enc(a, a, f), np.uint8)[6 * W:10 * W].view("<i2").reshape(2, W)
        assert q[0].tolist() == [0, 0, 1, -1, 32764, 32767, 32767, -32768]
        assert q[1].tolist() == [32767, 32, -32, 0, 0, 1, -32768, 32767]


@pytest.mark.parametrize("size", SIZES)
def test_oracle_decode_matches_the_numpy_statement_and_round_trips(size):
    H, W = size
    N = 3
    ins = [_sample_inputs(H, W, 7 * i + H) for i in range(N)]
    samples = np.stack([np.frombuffer(oracle.custom_data_encode_sample(*x), np.uint8) for x in ins])
    mean = np.random.default_rng(5).standard_normal(9 * H * W).astype(np.float32) * 50
    for m, scale in [(None, 1.0), (mean, 1.0 / 255), (None, 0.5)]:
        tops = oracle.custom_data_decode(samples, 9, H, W, SP, ENC, mean=m, scale=scale)
        for i in range(N):
            want = _numpy_unpack(samples[i].tobytes(), H, W, m, scale)
            for t, w in zip(tops, want):
                assert np.array_equal(t[i].view(np.uint32) | (np.isnan(t[i]) * 0x7fffffff).astype(np.uint32),
                                      w.view(np.uint32) | (np.isnan(w) * 0x7fffffff).astype(np.uint32))
    tops = oracle.custom_data_decode(samples, 9, H, W, SP, ENC)
    for i, (a, b, f, o) in enumerate(ins):                 # round trip: images and occlusions exact, flow to 1/32 px toward zero
        assert np.array_equal(tops[0][i], a.transpose(2, 0, 1)) and np.array_equal(tops[1][i], b.transpose(2, 0, 1))
        assert np.array_equal(tops[3][i, 0], o)
        assert np.array_equal(np.isnan(tops[2][i]), np.isnan(f))
        ok = ~np.isnan(f)
        assert np.array_equal(tops[2][i][ok], (np.trunc(f[ok] * 32) / 32).astype(np.float32))
        assert np.all(tops[2][i][np.isnan(f)].view(np.uint32) == 0x7fe00000)      # sNaN quieted by the mean subtraction
    # re-encoding what was decoded reproduces the bytes (the format is a fixed point of decode -> encode)
    for i in range(N):
        again = oracle.custom_data_encode_sample(tops[0][i].transpose(1, 2, 0).astype(np.uint8), tops[1][i].transpose(1, 2, 0).astype(np.uint8),
                                                 tops[2][i], tops[3][i, 0] > 0)
        assert again == samples[i].tobytes()


def test_slicing_defaults_float_data_and_invalid_slicings():
    H, W, N = 4, 6, 2
    raw = np.random.default_rng(3).integers(0, 256, (N, 5 * H * W + 3), dtype=np.uint8)
    # encodings shorter than the slices: the rest is UINT8 (:80-83); no slice point: one top with every channel
    t = oracle.custom_data_decode(raw, 5, H, W)
    assert len(t) == 1 and np.array_equal(t[0].reshape(N, -1), raw[:, :5 * H * W].astype(np.float32))
    t = oracle.custom_data_decode(raw, 4, H, W, (1,), (SF.UINT16FLOW,))
    assert [x.shape for x in t] == [(N, 1, H, W), (N, 3, H, W)]
    assert np.array_equal(t[1].reshape(N, -1), raw[:, 2 * H * W:5 * H * W].astype(np.float32))
    fl = np.random.default_rng(4).standard_normal((N, 3 * H * W)).astype(np.float32)
    t = oracle.custom_data_decode(fl, 3, H, W, (2,), (), mean=np.ones(3 * H * W, np.float32), scale=2.0, float_data=True)
    assert np.array_equal(np.concatenate([x.reshape(N, -1) for x in t], 1), (fl - 1) * 2)
    for bad in [dict(channels=9, slice_points=(3, 3, 8), encodings=ENC), dict(channels=9, slice_points=(3, 6, 10), encodings=ENC),
                dict(channels=9, slice_points=(3, 6, 7), encodings=ENC),          # BOOL1 slice with two channels
                dict(channels=9, slice_points=(3, 6, 8), encodings=(1, 1, 2, 4)), dict(channels=0, slice_points=(), encodings=())]:
        assert oracle.custom_data_sample_bytes(bad["channels"], H, W, bad["slice_points"], bad["encodings"]) == 0
        with pytest.raises(ValueError):
            SF.sample_bytes(bad["channels"], H, W, bad["slice_points"], bad["encodings"])
    with pytest.raises(ValueError):                           # "Encoded layers must be stored as uint8 in LMDB." (:55)
        oracle.custom_data_decode(fl, 3, H, W, (2,), (1, 1), float_data=True)
    with pytest.raises(ValueError, match="no CPU path"):
        SF.decode_batch(torch.zeros(1, 10, dtype=torch.uint8), 1, 2, 5)


def test_stage_records_host_function_matches_the_oracle_and_checks_shapes():
    H, W = 12, 20
    recs = [SF.make_record(*_sample_inputs(H, W, 70 + i), label=5 - i) for i in range(5)]
    got, shape, labels = SF.stage_records(recs, "cpu")
    want, wshape, wlabels = oracle.custom_data_stage_records(recs)
    nb = SF.sample_bytes(9, H, W, SP, ENC)
    assert shape == wshape == (9, H, W) and labels == wlabels == [5, 4, 3, 2, 1]
    assert got.shape[1] % 16 == 0 and np.array_equal(got.numpy()[:, :nb], want) and want.shape == (5, nb)
    for i, r in enumerate(recs):
        assert want[i].tobytes() == SF.parse_datum(r).data
    big = [SF.make_record(*_sample_inputs(128, 160, 80 + i)) for i in range(6)]          # > 1 MB in total: the threaded copy
    g2, _, _ = SF.stage_records(big, "cpu")
    w2, _, _ = oracle.custom_data_stage_records(big)
    assert np.array_equal(g2.numpy()[:, :w2.shape[1]], w2)
    other = SF.make_record(*_sample_inputs(H, W + 1, 3))
    fl = _datum_class()()
    fl.channels, fl.height, fl.width = 1, 1, 2
    fl.float_data.extend([1.0, 2.0])
    for bad in ([recs[0], other], [recs[0], recs[1][:-7]], [fl.SerializeToString()]):
        with pytest.raises(flownet2_amd.Fn2Error):
            SF.stage_records(bad, "cpu")
        with pytest.raises(ValueError):
            oracle.custom_data_stage_records(bad)


def test_records_round_trip_through_the_datum_container():
    H, W = 6, 10
    a, b, f, o = _sample_inputs(H, W, 11)
    rec = SF.make_record(a, b, f, o, label=42)
    assert rec == oracle.datum_serialize(9, H, W, oracle.custom_data_encode_sample(a, b, f, o), 42)
    d = _datum_class()()
    d.ParseFromString(rec)
    assert (d.channels, d.height, d.width, d.label) == (9, H, W, 42) and d.data == _numpy_pack(a, b, f, o)
    p = SF.parse_datum(rec)
    assert (p.channels, p.height, p.width, p.label, p.data) == (9, H, W, 42, d.data)


def test_custom_data_layer_mirror_setup_and_checks():
    """Host logic of the Layer mirror (no GPU work in SetUp): top shapes from the first record, the reference's CHECKs, the cursor."""
    from flownet2_amd.layers import Blob, CheckError, LayerParameter, LayerRegistry
    H, W = 6, 10
    recs = [("%08d_p%d" % (i, i), SF.make_record(*_sample_inputs(H, W, i), label=i)) for i in range(5)]

    def make(ntop, **dp):
        base = dict(source=recs, backend="LMDB", batch_size=2, slice_point=list(SP), encoding=["UINT8", "UINT8", "UINT16FLOW", "BOOL1"])
        base.update(dp)
        layer = LayerRegistry.CreateLayer(LayerParameter(name="data", type="CustomData", data_param=base))
        top = [Blob(device="cpu") for _ in range(ntop)]
        layer.SetUp([], top)
        return layer, top
    layer, top = make(4)
    assert [t.shape() for t in top] == [[2, 3, H, W], [2, 3, H, W], [2, 2, H, W], [2, 1, H, W]]
    assert layer.channel_encoding_ == list(ENC)
    order = [SF.parse_datum(layer._next_record()).label for _ in range(12)]
    assert order == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1]                      # wraps around (:177)
    layer, top = make(4, range_start=1, range_end=3, source=dict(reversed(recs)))   # any record order: the environment is sorted by key
    assert [SF.parse_datum(layer._next_record()).label for _ in range(5)] == [1, 2, 3, 1, 2]
    layer, top = make(2, slice_point=[], encoding=[])
    assert [t.shape() for t in top] == [[2, 9, H, W], [2, 1, 1, 1]] and layer.output_labels_
    for ntop, dp, msg in [(3, {}, "has 3 top blobs, but 4 slices"), (4, dict(backend="LEVELDB"), "LevelDB not supported"),
                          (4, dict(crop_size=4), "Cropping currently not supported"), (4, dict(rand_skip=3), "No rand_skip"),
                          (5, {}, "slice_point_.size\\(\\) == top.size\\(\\) - 1"),           # labels + slicing: the reference's own CHECK (:513)
                          (4, dict(range_start=3, range_end=1), "Range end is before start"),
                          (4, dict(slice_point=[3, 3, 8]), "slice_point_\\[i\\] > prev"), (4, dict(rand_permute=True), "rand_permute is not reproduced"),
                          (4, dict(source=[]), "mdb_env_open failed")]:
        with pytest.raises(CheckError, match=msg):
            make(ntop, **dp)
    # data-parallel ranks: rank r owns batches r, r + world, ... of the one cursor the reference shares between its solver threads
    single, _ = make(4, batch_size=2)
    want = [[SF.parse_datum(r).label for r in single._next_batch()] for _ in range(9)]
    for world in (2, 3):
        got = {}
        for rank in range(world):
            layer, _ = make(4, batch_size=2, world=world, rank=rank)
            for it in range(9 // world):
                got[it * world + rank] = [SF.parse_datum(r).label for r in layer._next_batch()]
                layer.iter_ += 1
        assert [got[b] for b in sorted(got)] == want[:len(got)]
    with pytest.raises(CheckError, match="rank must be in"):
        make(4, world=2, rank=2)
    with pytest.raises(CheckError):                                             # bottoms are not allowed (ExactNumBottomBlobs = 0)
        LayerRegistry.CreateLayer(LayerParameter(type="CustomData", data_param=dict(source=recs, backend="LMDB"))).SetUp([Blob(device="cpu")], [Blob(device="cpu")])


# ---------------------------------------------------------------------------------------------------------
# GPU: decode kernels vs the oracle, bit for bit
# ---------------------------------------------------------------------------------------------------------
def _bits(x):
    return np.ascontiguousarray(x).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("size", SIZES + [(48, 64), (33, 47)])
@pytest.mark.parametrize("pad", [0, 1, 6, 16])
def test_hip_decode_is_bit_exact(size, pad):
    H, W = size
    N = 3
    nbytes = SF.sample_bytes(9, H, W, SP, ENC)
    samples = np.zeros((N, nbytes + pad), np.uint8)
    for i in range(N):
        samples[i, :nbytes] = np.frombuffer(oracle.custom_data_encode_sample(*_sample_inputs(H, W, 31 * i + W)), np.uint8)
    mean = (np.random.default_rng(6).standard_normal(9 * H * W) * 40).astype(np.float32)
    dev = torch.from_numpy(samples).cuda()
    for m, scale in [(None, 1.0), (mean, 1.0 / 255), (None, 0.25)]:
        want = oracle.custom_data_decode(samples, 9, H, W, SP, ENC, mean=m, scale=scale)
        got = SF.decode_batch(dev, 9, H, W, SP, ENC, mean=torch.from_numpy(m).cuda() if m is not None else None, scale=scale)
        for g, w in zip(got, want):
            assert np.array_equal(_bits(g.cpu().numpy()), _bits(w))


@pytest.mark.gpu
def test_hip_decode_slicing_variants_float_data_and_errors():
    H, W, N = 5, 9, 4
    rng = np.random.default_rng(8)
    raw = rng.integers(0, 256, (N, 7 * H * W + 5), dtype=np.uint8)
    dev = torch.from_numpy(raw).cuda()
    for channels, sp, enc in [(7, (), ()), (4, (1,), (SF.UINT16FLOW,)), (5, (1, 3), (SF.UINT8, SF.UINT16FLOW)), (3, (1, 2), (SF.BOOL1, SF.UINT16FLOW, SF.BOOL1)),
                              (2, (1,), (SF.BOOL1, SF.BOOL1))]:
        want = oracle.custom_data_decode(raw, channels, H, W, sp, enc, scale=3.0)
        got = SF.decode_batch(dev, channels, H, W, sp, enc, scale=3.0)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert np.array_equal(_bits(g.cpu().numpy()), _bits(w))
    fl = rng.standard_normal((N, 3 * H * W + 2)).astype(np.float32)
    fl[0, 5] = np.nan
    mean = rng.standard_normal(3 * H * W).astype(np.float32)
    want = oracle.custom_data_decode(fl, 3, H, W, (2,), (), mean=mean, scale=2.0, float_data=True)
    got = SF.decode_batch(torch.from_numpy(fl).cuda(), 3, H, W, (2,), (), mean=torch.from_numpy(mean).cuda(), scale=2.0, float_data=True)
    for g, w in zip(got, want):
        assert np.array_equal(np.isnan(g.cpu().numpy()), np.isnan(w)) and np.array_equal(np.nan_to_num(g.cpu().numpy()), np.nan_to_num(w))
    assert [tuple(t.shape) for t in SF.decode_batch(dev[:0], 7, H, W)] == [(0, 7, H, W)]
    with pytest.raises(flownet2_amd.Fn2Error, match="stride"):
        SF.decode_batch(dev[:, :100].contiguous(), 7, H, W)
    with pytest.raises(flownet2_amd.Fn2Error, match="BOOL1 slice"):
        SF.decode_batch(dev, 3, H, W, (1,), (SF.UINT8, SF.BOOL1))
    with pytest.raises(flownet2_amd.Fn2Error, match="Invalid format"):
        SF.decode_batch(dev, 3, H, W, (1,), (SF.UINT8, 9))
    with pytest.raises(flownet2_amd.Fn2Error, match="Encoded layers must be stored as uint8"):
        SF.decode_batch(torch.from_numpy(fl).cuda(), 3, H, W, (2,), (1,), float_data=True)


@pytest.mark.gpu
def test_records_to_training_blobs_at_flying_chairs_size():
    """LMDB values -> host Datum parse -> raw bytes to the GPU -> decode: a batch of 8 FlyingChairs-sized samples (512x384).
    Size-independent properties: exact images / occlusions, flow = trunc(flow * 32) / 32, NaN preserved, labels carried."""
    H, W, N = 384, 512, 8
    ins = [_sample_inputs(H, W, 50 + i, nan_frac=0.01) for i in range(N)]
    records = [SF.make_record(*x, label=i * 3) for i, x in enumerate(ins)]
    samples, (channels, h, w), labels = SF.stage_records(records)
    assert (channels, h, w) == (9, H, W) and labels == [i * 3 for i in range(N)]
    img0, img1, flow, occ = SF.decode_batch(samples, channels, h, w, SP, ENC)
    for i, (a, b, f, o) in enumerate(ins):
        assert torch.equal(img0[i].cpu(), torch.from_numpy(a.transpose(2, 0, 1).astype(np.float32)))
        assert torch.equal(img1[i].cpu(), torch.from_numpy(b.transpose(2, 0, 1).astype(np.float32)))
        assert torch.equal(occ[i, 0].cpu(), torch.from_numpy(o.astype(np.float32)))
        g = flow[i].cpu().numpy()
        assert np.array_equal(np.isnan(g), np.isnan(f))
        ok = ~np.isnan(f)
        assert np.array_equal(g[ok], (np.trunc(f[ok] * 32) / 32).astype(np.float32))
    # one sample against the oracle, bit for bit
    want = oracle.custom_data_decode(samples[:1].cpu().numpy(), 9, H, W, SP, ENC)
    for g, w_ in zip((img0, img1, flow, occ), want):
        assert np.array_equal(_bits(g[:1].cpu().numpy()), _bits(w_))
