"""Generates the HDF5 weight fixtures under tests/golden/h5/ with the REAL libhdf5 (1.10.6, /opt/conda/lib -- present in the build
container only; no h5py anywhere), driven through ctypes with the calls the reference itself makes:

    Net::ToHDF5 (src/caffe/net.cpp:896-950):  H5Fcreate; H5Gcreate2 "data" (and "diff"); per layer H5Gcreate2 <layer name>;
    per parameter blob hdf5_save_nd_dataset -> H5LTmake_dataset_float(<group>, "<index>", num_axes, dims, data)
    (src/caffe/util/hdf5.cpp:81-101).

Run HERE; the outputs (small files + the arrays they hold as .npz) are committed.  The reader under test is
flownet2_amd/csrc/hdf5_reader.cpp, which shares no code with libhdf5.

  tiny.caffemodel.h5      the HDF5 twin of tests/golden/tiny.caffemodel: the same layers and values in the Net::ToHDF5 layout
                          (default libver bounds: superblock v0, v1 object headers, symbol-table groups, contiguous float32)
  tiny_latest.caffemodel.h5   the same written with H5Pset_libver_bounds(LATEST, LATEST): superblock v3, "OHDR" headers, compact links
  many_layers.caffemodel.h5   40 layers: the `data` group's B-tree has several symbol-table nodes; names sort like H5_INDEX_NAME
  odd_types.h5            what hdf5_load_nd_dataset also accepts: chunked + shuffle + gzip big-endian float64 with partial edge chunks,
                          int16, a compact dataset, a never-written dataset (fill value), a dataset inside a nested group
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5")

h5 = C.CDLL("/opt/conda/lib/libhdf5.so.103", mode=C.RTLD_GLOBAL)
hl = C.CDLL("/opt/conda/lib/libhdf5_hl.so")
hid = C.c_int64
h5.H5open()
for f, res, args in (("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]),
                     ("H5Pcreate", hid, [hid]), ("H5Pset_libver_bounds", C.c_int, [hid, C.c_int, C.c_int]),
                     ("H5Pset_chunk", C.c_int, [hid, C.c_int, C.POINTER(C.c_uint64)]), ("H5Pset_deflate", C.c_int, [hid, C.c_uint]),
                     ("H5Pset_shuffle", C.c_int, [hid]), ("H5Pset_layout", C.c_int, [hid, C.c_int]),
                     ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
                     ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                     ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
                     ("H5Dclose", C.c_int, [hid]), ("H5Sclose", C.c_int, [hid]), ("H5Pclose", C.c_int, [hid]),
                     ("H5Gclose", C.c_int, [hid]), ("H5Fclose", C.c_int, [hid])):
    getattr(h5, f).restype, getattr(h5, f).argtypes = res, args
hl.H5LTmake_dataset_float.restype = C.c_int
hl.H5LTmake_dataset_float.argtypes = [hid, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]


def gid(name):
    return hid.in_dll(h5, name).value


def make_float(loc, name, a):
    a = np.ascontiguousarray(a, np.float32)
    dims = (C.c_uint64 * a.ndim)(*a.shape)
    assert hl.H5LTmake_dataset_float(loc, name.encode(), a.ndim, dims, a.ctypes.data) >= 0


def to_hdf5(path, layers, latest=False):
    """Net::ToHDF5 with write_diff = false: layers = [(name, [blob arrays])]."""
    fapl = 0
    if latest:
        fapl = h5.H5Pcreate(gid("H5P_CLS_FILE_ACCESS_ID_g"))
        assert h5.H5Pset_libver_bounds(fapl, 2, 2) >= 0          # H5F_LIBVER_LATEST (= V110 in this library)
    f = h5.H5Fcreate(path.encode(), 2, 0, fapl)                  # H5F_ACC_TRUNC
    assert f >= 0
    data = h5.H5Gcreate2(f, b"data", 0, 0, 0)
    for name, blobs in layers:
        g = h5.H5Gcreate2(data, name.encode(), 0, 0, 0)
        assert g >= 0
        for j, b in enumerate(blobs):
            make_float(g, str(j), b)
        h5.H5Gclose(g)
    h5.H5Gclose(data)
    h5.H5Fclose(f)
    if fapl:
        h5.H5Pclose(fapl)


def tiny_layers():
    z = np.load(os.path.join(HERE, "tiny_caffemodel.npz"))
    return [("img0s_aug", [z["img0s_aug.count"], z["img0s_aug.pixel_mean"], z["img0s_aug.mean"]]),
            ("conv1", [z["conv1.w"], z["conv1.b"]]),
            ("deconv5", [z["deconv5.w"], z["deconv5.b"]]),
            ("net2_conv6", [z["net2_conv6.w"], z["net2_conv6.b"].reshape(2)]),
            ("fuse_conv0", [z["fuse_conv0.w"]])]


def odd_types(path, arrays):
    rng = np.random.default_rng(7)
    f = h5.H5Fcreate(path.encode(), 2, 0, 0)

    def dset(loc, name, a, file_type, mem_type, chunk=None, gzip=None, shuffle=False, compact=False, write=True):
        dims = (C.c_uint64 * a.ndim)(*a.shape)
        sp = h5.H5Screate_simple(a.ndim, dims, None)
        dcpl = h5.H5Pcreate(gid("H5P_CLS_DATASET_CREATE_ID_g"))
        if chunk:
            assert h5.H5Pset_chunk(dcpl, len(chunk), (C.c_uint64 * len(chunk))(*chunk)) >= 0
        if shuffle:
            assert h5.H5Pset_shuffle(dcpl) >= 0
        if gzip is not None:
            assert h5.H5Pset_deflate(dcpl, gzip) >= 0
        if compact:
            assert h5.H5Pset_layout(dcpl, 0) >= 0
        d = h5.H5Dcreate2(loc, name.encode(), gid(file_type), sp, 0, dcpl, 0)
        assert d >= 0
        if write:
            assert h5.H5Dwrite(d, gid(mem_type), 0, 0, 0, a.ctypes.data) >= 0
        h5.H5Dclose(d); h5.H5Sclose(sp); h5.H5Pclose(dcpl)

    a = rng.standard_normal((5, 7, 11))                        # chunks of 2x4x8: partial chunks on every axis, big-endian on disk
    dset(f, "f64be_gzip_shuffle", a, "H5T_IEEE_F64BE_g", "H5T_NATIVE_DOUBLE_g", chunk=(2, 4, 8), gzip=4, shuffle=True)
    arrays["/f64be_gzip_shuffle"] = a.astype(np.float32)
    b = rng.integers(-30000, 30000, (6, 9)).astype(np.int16)
    dset(f, "i16_chunked", b, "H5T_STD_I16LE_g", "H5T_NATIVE_SHORT_g", chunk=(4, 4))
    arrays["/i16_chunked"] = b.astype(np.float32)
    c = rng.standard_normal((3, 4)).astype(np.float32)
    dset(f, "f32_compact", c, "H5T_IEEE_F32LE_g", "H5T_NATIVE_FLOAT_g", compact=True)
    arrays["/f32_compact"] = c
    dset(f, "never_written", np.zeros((4, 3), np.float32), "H5T_IEEE_F32LE_g", "H5T_NATIVE_FLOAT_g", write=False)
    arrays["/never_written"] = np.zeros((4, 3), np.float32)
    e = rng.integers(0, 2 ** 32, (10,), dtype=np.uint64).astype(np.uint32)
    dset(f, "u32be", e, "H5T_STD_U32BE_g", "H5T_NATIVE_UINT_g")
    arrays["/u32be"] = e.astype(np.float32)
    g = h5.H5Gcreate2(f, b"outer", 0, 0, 0)
    g2 = h5.H5Gcreate2(g, b"inner", 0, 0, 0)
    s = rng.standard_normal((2, 2)).astype(np.float32)
    make_float(g2, "leaf", s)
    arrays["/outer/inner/leaf"] = s
    h5.H5Gclose(g2); h5.H5Gclose(g); h5.H5Fclose(f)


def main():
    os.makedirs(OUT, exist_ok=True)
    tiny = tiny_layers()
    to_hdf5(os.path.join(OUT, "tiny.caffemodel.h5"), tiny)
    to_hdf5(os.path.join(OUT, "tiny_latest.caffemodel.h5"), tiny, latest=True)
    rng = np.random.default_rng(11)
    many = [("layer_%02d" % i if i % 3 else "Conv%d" % i, [rng.standard_normal((2, 3, 1, 1)).astype(np.float32), rng.standard_normal(2).astype(np.float32)])
            for i in range(40)]
    to_hdf5(os.path.join(OUT, "many_layers.caffemodel.h5"), many)
    np.savez(os.path.join(OUT, "many_layers.npz"), **{"%s/%d" % (n, j): b for n, bl in many for j, b in enumerate(bl)})
    arrays = {}
    odd_types(os.path.join(OUT, "odd_types.h5"), arrays)
    np.savez(os.path.join(OUT, "odd_types.npz"), **{k.replace("/", "|"): v for k, v in arrays.items()})
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
