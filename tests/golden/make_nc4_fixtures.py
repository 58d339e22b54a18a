#!/usr/bin/env python3
"""Writes the NetCDF-4 / HDF5 fixtures of tests/test_hdf5_cpu.py with libhdf5 ITSELF (1.10.6, /opt/conda/lib — present in the
build container only; the GPU box and the tests need just the committed files):

    python tests/golden/make_nc4_fixtures.py

  nc4_era5_like.nc   what netCDF-C writes for an ERA5 day file: creation-order-tracked root group and variables (version-2
                     object headers), dimension scales with DIMENSION_LIST / REFERENCE_LIST, an unlimited time axis (chunked,
                     B-tree v1 index), int16-packed u10 / v10 (scale_factor, add_offset, _FillValue, missing_value) with
                     shuffle + deflate, a big-endian float64 variable with fletcher32, a variable-length string attribute
  nc4_latest.nc      the same data written with libver = latest (superblock 3, version-4 layouts: fixed-array, single-chunk and
                     implicit chunk indexes; extensible arrays for the unlimited axes), 17 variables (dense link storage: fractal heap + B-tree v2) and 11 attributes on
                     one of them (dense attribute storage)
  h5_plain.h5        superblock 0, symbol-table groups (nested, 23 links), version-1 object headers, compact / big-endian /
                     deflated datasets: what h5py and PyTables write by default
  nc4_expected.npz   the arrays as handed to H5Dwrite / H5Awrite (what a reader must return before CF decoding)
"""
import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
lib = C.CDLL("/opt/conda/lib/libhdf5.so.103")
hl = C.CDLL("/opt/conda/lib/libhdf5_hl.so.100")
hid = C.c_int64
lib.H5open()
G = lambda name: hid.in_dll(lib, name).value
for fn in ("H5Fcreate", "H5Pcreate", "H5Tcopy", "H5Screate_simple", "H5Screate", "H5Dcreate2", "H5Acreate2"):
    getattr(lib, fn).restype = hid
UNLIMITED = C.c_uint64(-1).value


def check(rc, what):
    if rc < 0:
        raise RuntimeError(what)
    return rc


def dims(v):
    return (C.c_uint64 * len(v))(*v)


def space(shape, maxshape=None):
    if shape == ():
        return check(lib.H5Screate(C.c_int(0)), "H5Screate")
    return check(lib.H5Screate_simple(C.c_int(len(shape)), dims(shape), dims(maxshape) if maxshape else None), "H5Screate_simple")


def attr(loc, name, value, ftype=None, vlen=False):
    if isinstance(value, str):
        t = check(lib.H5Tcopy(hid(G("H5T_C_S1_g"))), "H5Tcopy")
        raw = value.encode()
        if vlen:
            lib.H5Tset_size(hid(t), C.c_size_t(-1))
            buf = (C.c_char_p * 1)(raw)
        else:
            lib.H5Tset_size(hid(t), C.c_size_t(len(raw) + 1))
            buf = C.create_string_buffer(raw, len(raw) + 1)
        s = space(())
        a = check(lib.H5Acreate2(hid(loc), name.encode(), hid(t), hid(s), hid(0), hid(0)), "H5Acreate2 " + name)
        check(lib.H5Awrite(hid(a), hid(t), buf), "H5Awrite " + name)
    else:
        arr = np.atleast_1d(np.asarray(value))
        mem = {"float64": "H5T_NATIVE_DOUBLE_g", "float32": "H5T_NATIVE_FLOAT_g", "int16": "H5T_NATIVE_SHORT_g",
               "int32": "H5T_NATIVE_INT_g"}[arr.dtype.name]
        s = space(arr.shape)
        a = check(lib.H5Acreate2(hid(loc), name.encode(), hid(G(ftype or mem)), hid(s), hid(0), hid(0)), "H5Acreate2 " + name)
        check(lib.H5Awrite(hid(a), hid(G(mem)), arr.ctypes.data_as(C.c_void_p)), "H5Awrite " + name)
    lib.H5Aclose(hid(a))
    lib.H5Sclose(hid(s))


def dataset(f, name, arr, ftype, maxshape=None, chunks=None, deflate=0, shuffle=False, fletcher=False, early=False, track=True):
    mem = {"float64": "H5T_NATIVE_DOUBLE_g", "float32": "H5T_NATIVE_FLOAT_g", "int16": "H5T_NATIVE_SHORT_g",
           "int32": "H5T_NATIVE_INT_g"}[arr.dtype.name]
    dcpl = check(lib.H5Pcreate(hid(G("H5P_CLS_DATASET_CREATE_ID_g"))), "H5Pcreate")
    if track:
        check(lib.H5Pset_attr_creation_order(hid(dcpl), C.c_uint(3)), "attr creation order")      # tracked | indexed, as netCDF-C
    if chunks:
        check(lib.H5Pset_chunk(hid(dcpl), C.c_int(len(chunks)), dims(chunks)), "H5Pset_chunk")
    if shuffle:
        check(lib.H5Pset_shuffle(hid(dcpl)), "shuffle")
    if deflate:
        check(lib.H5Pset_deflate(hid(dcpl), C.c_uint(deflate)), "deflate")
    if fletcher:
        check(lib.H5Pset_fletcher32(hid(dcpl)), "fletcher32")
    if early:
        check(lib.H5Pset_alloc_time(hid(dcpl), C.c_int(1)), "alloc time early")
    s = space(arr.shape, maxshape)
    d = check(lib.H5Dcreate2(hid(f), name.encode(), hid(G(ftype)), hid(s), hid(0), hid(dcpl), hid(0)), "H5Dcreate2 " + name)
    check(lib.H5Dwrite(hid(d), hid(G(mem)), hid(0), hid(0), hid(0), np.ascontiguousarray(arr).ctypes.data_as(C.c_void_p)), "H5Dwrite " + name)
    lib.H5Sclose(hid(s))
    lib.H5Pclose(hid(dcpl))
    return d


def write(path, latest, expected):
    rng = np.random.default_rng(7)
    T, NY, NX = 6, 5, 7
    lon = np.linspace(5.0, 6.5, NX).astype(np.float32)
    lat = np.linspace(47.5, 46.5, NY).astype(np.float32)
    time = (1058448 + np.arange(T)).astype(np.int32)                    # hours since 1900-01-01: 2020-09-30 00:00 ...
    u10 = rng.integers(-30000, 30000, (T, NY, NX)).astype(np.int16)
    u10[1, 2, 3] = -32767                                               # _FillValue
    v10 = rng.integers(-30000, 30000, (T, NY, NX)).astype(np.int16)
    v10[0, 0, 0] = -32767
    z = (rng.standard_normal((T, NY, NX)) * 500 + 9000).astype(np.float64)
    fcpl = check(lib.H5Pcreate(hid(G("H5P_CLS_FILE_CREATE_ID_g"))), "fcpl")
    check(lib.H5Pset_link_creation_order(hid(fcpl), C.c_uint(3)), "link creation order")
    check(lib.H5Pset_attr_creation_order(hid(fcpl), C.c_uint(3)), "root attr creation order")
    fapl = check(lib.H5Pcreate(hid(G("H5P_CLS_FILE_ACCESS_ID_g"))), "fapl")
    if latest:
        check(lib.H5Pset_libver_bounds(hid(fapl), C.c_int(2), C.c_int(2)), "libver bounds")     # H5F_LIBVER_V110 = latest of 1.10.6
    f = check(lib.H5Fcreate(str(path).encode(), C.c_uint(2), hid(fcpl), hid(fapl)), "H5Fcreate")  # H5F_ACC_TRUNC
    attr(f, "Conventions", "CF-1.6")
    attr(f, "history", "2020-10-01 12:00:00 GMT by grib_to_netcdf-2.16.0: a variable-length string attribute", vlen=True)
    d_lon = dataset(f, "longitude", lon, "H5T_IEEE_F32LE_g")
    d_lat = dataset(f, "latitude", lat, "H5T_IEEE_F32LE_g")
    d_time = dataset(f, "time", time, "H5T_STD_I32LE_g", maxshape=(UNLIMITED,), chunks=(4,))
    for d, name, units in ((d_lon, "longitude", "degrees_east"), (d_lat, "latitude", "degrees_north"),
                           (d_time, "time", "hours since 1900-01-01 00:00:00.0")):
        check(hl.H5DSset_scale(hid(d), name.encode()), "H5DSset_scale")
        attr(d, "units", units)
        attr(d, "long_name", name)
    attr(d_time, "calendar", "gregorian")
    for i, d in enumerate((d_time, d_lat, d_lon)):
        attr(d, "_Netcdf4Dimid", np.int32(i))
    kw = dict(maxshape=(UNLIMITED, NY, NX), chunks=(2 if latest else 1, NY, NX))
    d_u = dataset(f, "u10", u10, "H5T_STD_I16LE_g", deflate=4, shuffle=True, **kw)
    d_v = dataset(f, "v10", v10, "H5T_STD_I16LE_g", deflate=1, shuffle=True, **kw)
    d_z = dataset(f, "z", z, "H5T_IEEE_F64BE_g", chunks=(T, NY, NX) if latest else (3, 2, 4), fletcher=True)
    packed = {"u10": (d_u, 3.1e-4, 1.25), "v10": (d_v, 2.9e-4, -0.75)}
    for name, (d, sf, ao) in packed.items():
        attr(d, "scale_factor", np.float64(sf))
        attr(d, "add_offset", np.float64(ao))
        attr(d, "_FillValue", np.int16(-32767))
        attr(d, "missing_value", np.int16(-32767))
        attr(d, "units", "m s**-1")
        attr(d, "long_name", f"10 metre {name[0].upper()} wind component")
    attr(d_z, "units", "m**2 s**-2")
    extra = {}
    if latest:
        for k in range(5):                                              # 11 attributes on u10: dense attribute storage
            attr(d_u, f"extra_{k}", np.float32(k + 0.5))
        # implicit chunk index: no filters, early allocation, fixed shape; nine more variables: dense link storage
        for k in range(9):
            arr = (rng.standard_normal((NY, NX)) * (k + 1)).astype(np.float32)
            extra[f"field_{k}"] = arr
            d = dataset(f, f"field_{k}", arr, "H5T_IEEE_F32LE_g", chunks=(2, 3) if k == 0 else None, early=(k == 0))
            for i, s in enumerate((d_lat, d_lon)):
                check(hl.H5DSattach_scale(hid(d), hid(s), C.c_uint(i)), "attach scale")
            lib.H5Dclose(hid(d))
        # 300 one-row chunks along an unlimited axis: extensible-array index through its index block, data blocks and a super block
        series = rng.integers(-9, 9, (300, 2)).astype(np.int16)
        extra["series"] = series
        lib.H5Dclose(hid(dataset(f, "series", series, "H5T_STD_I16LE_g", maxshape=(UNLIMITED, 2), chunks=(1, 2))))
        # ... and with the unlimited axis NOT the slowest one (chunk indices count along it first)
        wide = rng.integers(-9, 9, (3, 40)).astype(np.int16)
        extra["wide"] = wide
        lib.H5Dclose(hid(dataset(f, "wide", wide, "H5T_STD_I16LE_g", maxshape=(3, UNLIMITED), chunks=(1, 4), deflate=2)))
    for d in (d_u, d_v, d_z):
        for i, s in enumerate((d_time, d_lat, d_lon)):
            check(hl.H5DSattach_scale(hid(d), hid(s), C.c_uint(i)), "H5DSattach_scale")
    for d in (d_lon, d_lat, d_time, d_u, d_v, d_z):
        lib.H5Dclose(hid(d))
    check(lib.H5Fclose(hid(f)), "H5Fclose")
    tag = "latest_" if latest else ""
    expected.update({tag + "longitude": lon, tag + "latitude": lat, tag + "time": time, tag + "u10": u10, tag + "v10": v10, tag + "z": z})
    expected.update({tag + k: v for k, v in extra.items()})
    expected[tag + "scale_u10"] = np.array([3.1e-4, 1.25])
    expected[tag + "scale_v10"] = np.array([2.9e-4, -0.75])


def write_plain(path, expected):
    """A plain HDF5 file as h5py / PyTables write by default: superblock 0, symbol-table groups (B-tree v1 + local heap),
    version-1 object headers; a nested group, a compact dataset, big-endian integers, a deflated chunked dataset."""
    rng = np.random.default_rng(11)
    f = check(lib.H5Fcreate(str(path).encode(), C.c_uint(2), hid(0), hid(0)), "H5Fcreate")
    lib.H5Gcreate2.restype = hid
    g = check(lib.H5Gcreate2(hid(f), b"fields", hid(0), hid(0), hid(0)), "H5Gcreate2")
    a = rng.standard_normal((9, 13)).astype(np.float32)
    bint = rng.integers(-1000, 1000, (4, 6)).astype(np.int32)
    small = np.arange(5, dtype=np.int16)
    d = dataset(g, "gz", a, "H5T_IEEE_F32LE_g", chunks=(4, 5), deflate=6, track=False)
    attr(d, "units", "K")
    attr(d, "levels", np.array([1.5, 2.5, 3.5], dtype=np.float64))
    lib.H5Dclose(hid(d))
    lib.H5Dclose(hid(dataset(g, "be", bint, "H5T_STD_I32BE_g", track=False)))
    dcpl = check(lib.H5Pcreate(hid(G("H5P_CLS_DATASET_CREATE_ID_g"))), "dcpl")
    check(lib.H5Pset_layout(hid(dcpl), C.c_int(0)), "compact layout")
    sp = space(small.shape)
    d = check(lib.H5Dcreate2(hid(f), b"compact", hid(G("H5T_STD_I16LE_g")), hid(sp), hid(0), hid(dcpl), hid(0)), "H5Dcreate2")
    check(lib.H5Dwrite(hid(d), hid(G("H5T_NATIVE_SHORT_g")), hid(0), hid(0), hid(0), small.ctypes.data_as(C.c_void_p)), "H5Dwrite")
    lib.H5Dclose(hid(d))
    for k in range(20):                                  # enough links for a second symbol-table node level
        lib.H5Dclose(hid(dataset(g, f"v{k:02d}", np.full((2,), k, np.int32), "H5T_STD_I32LE_g", track=False)))
    lib.H5Gclose(hid(g))
    attr(f, "title", "plain HDF5")
    check(lib.H5Fclose(hid(f)), "H5Fclose")
    expected.update({"plain_gz": a, "plain_be": bint, "plain_compact": small})


def main():
    expected = {}
    write_plain(HERE / "h5_plain.h5", expected)
    write(HERE / "nc4_era5_like.nc", False, expected)
    write(HERE / "nc4_latest.nc", True, expected)
    np.savez_compressed(HERE / "nc4_expected.npz", **expected)
    for p in ("nc4_era5_like.nc", "nc4_latest.nc", "h5_plain.h5", "nc4_expected.npz"):
        print(p, (HERE / p).stat().st_size, "bytes")


if __name__ == "__main__":
    main()
