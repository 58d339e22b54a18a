"""The package's own read-only HDF5 / NetCDF-4 reader (downscaling/io/hdf5.py: what `xr.open_mfdataset` of the reference's CLI,
/root/reference/src/downscaling/cli.py:22, reads through netCDF4 / h5netcdf) against files written by libhdf5 1.10.6 itself
(tests/golden/make_nc4_fixtures.py, run in the build container) and the arrays that were handed to the library."""
from pathlib import Path

import numpy as np
import pytest

from downscaling.io import hdf5
from downscaling.io.netcdf import open_dataset, open_mfdataset

GOLD = Path(__file__).resolve().parent / "golden"
EXPECTED = np.load(GOLD / "nc4_expected.npz")


def test_netcdf_c_style_file_raw_values():
    """Version-2 object headers, dimension scales, unlimited time axis (chunk B-tree v1), shuffle + deflate int16, fletcher32
    big-endian float64, variable-length string attribute: every array bit for bit, every attribute."""
    f = hdf5.File(GOLD / "nc4_era5_like.nc")
    assert sorted(f.root.members) == ["latitude", "longitude", "time", "u10", "v10", "z"]
    assert f.root.attrs["Conventions"] == "CF-1.6" and f.root.attrs["history"].startswith("2020-10-01 12:00:00 GMT by grib_to_netcdf")
    for name, d in f.root.datasets().items():
        got = d.read()
        assert got.dtype == EXPECTED[name].dtype and np.array_equal(got, EXPECTED[name]), name
    u = f.root["u10"]
    assert u.maxshape[0] == 2 ** 64 - 1 and u.shape == (6, 5, 7)
    assert float(u.attrs["scale_factor"][0]) == EXPECTED["scale_u10"][0] and float(u.attrs["add_offset"][0]) == EXPECTED["scale_u10"][1]
    assert int(u.attrs["_FillValue"][0]) == -32767 and u.attrs["units"] == "m s**-1"
    assert f.root["time"].attrs["units"] == "hours since 1900-01-01 00:00:00.0" and f.root["time"].attrs["CLASS"] == "DIMENSION_SCALE"
    coords, variables, _ = hdf5.read_netcdf4(GOLD / "nc4_era5_like.nc")
    assert set(coords) == {"time", "latitude", "longitude"}
    assert all(variables[v][0] == ("time", "latitude", "longitude") for v in ("u10", "v10", "z"))      # from DIMENSION_LIST


def test_libver_latest_file():
    """Superblock 3; dense link storage (17 links: fractal heap + B-tree v2), dense attribute storage (12 attributes), version-4
    layouts with single-chunk, implicit, fixed-array and extensible-array chunk indexes (300 chunks: index block, data blocks and a
    super block; an unlimited axis that is not the slowest)."""
    f = hdf5.File(GOLD / "nc4_latest.nc")
    names = sorted(f.root.members)
    assert len(names) == 17 and "field_8" in names and "series" in names
    for name, d in f.root.datasets().items():
        assert np.array_equal(d.read(), EXPECTED["latest_" + name]), name
    assert sorted(k for k in f.root["u10"].attrs if k.startswith("extra_")) == [f"extra_{k}" for k in range(5)]
    assert float(f.root["u10"].attrs["extra_3"][0]) == 3.5
    _, variables, _ = hdf5.read_netcdf4(GOLD / "nc4_latest.nc")
    assert variables["field_4"][0] == ("latitude", "longitude") and variables["v10"][0] == ("time", "latitude", "longitude")


def test_plain_hdf5_file():
    """Superblock 0, symbol-table groups (nested; 22 links under one group), version-1 object headers with continuation blocks,
    compact / big-endian / deflated datasets — the h5py / PyTables default."""
    f = hdf5.File(GOLD / "h5_plain.h5")
    assert f.root.attrs["title"] == "plain HDF5" and sorted(f.root.members) == ["compact", "fields"]
    g = f.root["fields"]
    assert isinstance(g, hdf5.Group) and len(g.members) == 22
    assert np.array_equal(g["gz"].read(), EXPECTED["plain_gz"]) and g["gz"].attrs["units"] == "K"
    assert np.array_equal(g["gz"].attrs["levels"], [1.5, 2.5, 3.5])
    assert g["be"].dtype == np.dtype(">i4") and np.array_equal(g["be"].read(), EXPECTED["plain_be"])
    assert np.array_equal(f.root["compact"].read(), EXPECTED["plain_compact"])
    assert np.array_equal(f.root["fields/v17"][...], [17, 17])


@pytest.mark.parametrize("name,tag", [("nc4_era5_like.nc", ""), ("nc4_latest.nc", "latest_")])
def test_open_dataset_decodes_netcdf4_like_xarray(name, tag):
    """open_dataset on a NetCDF-4 day file: packed int16 -> float32 with _FillValue -> NaN, hours since 1900 -> datetime64, as
    xarray's decode_cf (and as the NetCDF-3 path of the same function)."""
    ds = open_dataset(GOLD / name)
    assert ds.coords["time"].dtype.kind == "M" and ds.coords["time"][0] == np.datetime64("2020-09-30T00:00:00")
    assert np.array_equal(ds.coords["latitude"], EXPECTED[tag + "latitude"])
    for v in ("u10", "v10"):
        raw, (sf, ao) = EXPECTED[tag + v], EXPECTED[tag + "scale_" + v]
        got = ds[v]
        assert got.dtype == np.float32 and got.shape == raw.shape
        want = np.where(raw == -32767, np.nan, raw.astype(np.float32) * np.float32(sf) + np.float32(ao))
        np.testing.assert_array_equal(got, want)
        assert np.isnan(got).sum() == 1
    np.testing.assert_array_equal(ds["z"], EXPECTED[tag + "z"])
    assert open_mfdataset([GOLD / name])["u10"].shape == (6, 5, 7)


def test_unsupported_features_are_named(tmp_path):
    p = tmp_path / "truncated.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + bytes([9]) + b"\0" * 64)
    with pytest.raises(NotImplementedError, match="superblock version 9"):
        hdf5.File(p)
    assert hdf5.is_hdf5(GOLD / "h5_plain.h5") and not hdf5.is_hdf5(GOLD / "nc4_expected.npz")


def test_fill_value_of_unwritten_storage():
    """Storage that was never written (a contiguous dataset without an address, a missing chunk) reads as the dataset's FILL VALUE,
    as with libhdf5 — zeros would decode to add_offset, a plausible wind, where xarray gives NaN.  The Fill Value message (0x05,
    all three versions) is parsed; netCDF-4's `_FillValue` attribute is the fallback."""
    import struct
    from downscaling.io import hdf5
    val = struct.pack("<h", -32767)
    assert hdf5.File._fill_value(bytes([1, 2, 2, 1]) + struct.pack("<I", 2) + val) == val          # version 1
    assert hdf5.File._fill_value(bytes([2, 2, 2, 1]) + struct.pack("<I", 2) + val) == val          # version 2, defined
    assert hdf5.File._fill_value(bytes([2, 2, 2, 0])) is None                                      # version 2, undefined
    assert hdf5.File._fill_value(bytes([3, 0x20 | 0x09]) + struct.pack("<I", 2) + val) == val      # version 3, value follows
    assert hdf5.File._fill_value(bytes([3, 0x10 | 0x09])) is None                                  # version 3, explicitly undefined
    f = hdf5.File(GOLD / "nc4_era5_like.nc")
    packed = [d for d in f.root.datasets().values() if "_FillValue" in d.attrs and d.type.dtype.kind in "iu"]
    assert packed, "the fixture has packed variables"
    for d in packed:
        want = np.asarray(d.attrs["_FillValue"]).reshape(-1)[0]
        # (this fixture's Fill Value message is "defined, size 0" = the library default: the attribute decides; netCDF-C writes
        # the value into the message as well, and then the message decides)
        assert d._fill is None
        filled = f._filled(d, (2, 3), d.type.dtype.newbyteorder("="))
        assert filled.shape == (2, 3) and (filled == want).all()
        d._fill = struct.pack("<h", -5)
        assert (f._filled(d, (4,), d.type.dtype.newbyteorder("=")) == -5).all()
    plain = [d for d in f.root.datasets().values() if "_FillValue" not in d.attrs and d.type.dtype.kind in "iuf"]
    assert plain and (f._filled(plain[0], (3,), plain[0].type.dtype.newbyteorder("=")) == 0).all()
