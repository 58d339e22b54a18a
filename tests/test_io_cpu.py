"""File / grid I/O of the `downscale` driver (reference: cli.py:22-26 `open_mfdataset` / `open_rasterio` /
`to_netcdf`, api.py:31-62 nearest-neighbour regridding) — NetCDF-3 through scipy, the built-in GeoTIFF reader checked
against files written by an independent TIFF implementation (Pillow + libtiff), the regridding against brute force,
and the CLI end to end on the CPU oracle backend with small patched constants."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from downscaling.io import (GridDataset, nearest_index, open_dataset, open_mfdataset, open_raster, read_geotiff,
                            write_geotiff, write_netcdf)


def test_nearest_index_matches_brute_force_and_breaks_ties_upwards():
    rng = np.random.default_rng(0)
    for coord in (np.sort(rng.uniform(0, 10, 17)), np.sort(rng.uniform(0, 10, 17))[::-1].copy(), np.arange(5.0)):
        target = rng.uniform(-1, 11, 200)
        got = nearest_index(coord, target)
        want = np.abs(coord[None, :] - target[:, None]).argmin(1)
        assert np.array_equal(coord[got], coord[want])
    # a tie: pandas (and so xarray's method="nearest") prefers the larger index VALUE
    assert nearest_index(np.array([0.0, 1.0, 2.0]), np.array([0.5, 1.5])).tolist() == [1, 2]
    assert nearest_index(np.array([2.0, 1.0, 0.0]), np.array([0.5, 1.5])).tolist() == [1, 0]


def test_netcdf3_packed_era5_day_roundtrip(tmp_path):
    """An ERA5-style file as the CDS served it: int16-packed u10 / v10 with scale_factor / add_offset / _FillValue,
    `hours since 1900-01-01` time, descending latitudes."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(1)
    lon, lat = np.arange(5.0, 8.0, 0.25), np.arange(47.5, 45.0, -0.25)
    hours = 1_060_000 + np.arange(24)
    truth = {v: rng.uniform(-20, 20, (24, len(lat), len(lon))) for v in ("u10", "v10")}
    path = tmp_path / "20201203_era5_surface_hourly.nc"
    with netcdf_file(str(path), "w", version=2) as nc:
        for name, vals, typ in (("longitude", lon, "f4"), ("latitude", lat, "f4"), ("time", hours, "i4")):
            nc.createDimension(name, len(vals))
            var = nc.createVariable(name, typ, (name,))
            var[:] = vals
        nc.variables["time"].units = "hours since 1900-01-01 00:00:00.0"
        for v, arr in truth.items():
            scale, offset = 42.0 / 65000, 0.5
            packed = np.round((arr - offset) / scale).astype(np.int16)
            packed[3, 2, 1] = -32767
            var = nc.createVariable(v, "h", ("time", "latitude", "longitude"))
            var[:] = packed
            var.scale_factor, var.add_offset = scale, offset
            var._FillValue = np.int16(-32767)
            var.missing_value = np.int16(-32767)
    ds = open_dataset(path)
    assert ds.dims == {"longitude": len(lon), "latitude": len(lat), "time": 24}
    assert ds.coords["time"].dtype.kind == "M"
    assert ds.coords["time"][0] == np.datetime64("1900-01-01T00:00:00") + np.timedelta64(1_060_000, "h")
    for v, arr in truth.items():
        got = ds[v]
        assert got.dtype == np.float32 and np.isnan(got[3, 2, 1])
        mask = ~np.isnan(got)
        assert np.abs(got[mask] - arr[mask]).max() <= 42.0 / 65000 * 0.51 + 1e-5
    # write -> read of our own writer, and two half-day files combined along time like open_mfdataset
    halves = []
    for i, sl in enumerate((slice(12, 24), slice(0, 12))):
        part = ds.isel(time=np.arange(24)[sl])
        write_netcdf(part, tmp_path / f"20201204_part{i}_surface.nc")
        halves.append(tmp_path / f"20201204_part{i}_surface.nc")
    both = open_mfdataset(halves)
    assert np.array_equal(both.coords["time"], ds.coords["time"])
    np.testing.assert_array_equal(both["u10"], ds["u10"])
    with open(tmp_path / "not_netcdf.nc", "wb") as f:
        f.write(b"GRIB" + b"\0" * 64)
    with pytest.raises(OSError, match="neither"):
        open_dataset(tmp_path / "not_netcdf.nc")


@pytest.mark.parametrize("compression,dtype,predictor", [("raw", np.uint16, 1), ("tiff_lzw", np.uint16, 1),
                                                         ("tiff_lzw", np.uint16, 2), ("tiff_adobe_deflate", np.float32, 1),
                                                         ("packbits", np.uint8, 1), ("tiff_adobe_deflate", np.int32, 2)])
def test_geotiff_reader_against_pillow_written_files(tmp_path, compression, dtype, predictor):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(2)
    H, W = 70, 93
    yy, xx = np.mgrid[0:H, 0:W]
    dem = (500 + 300 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 3, (H, W)))
    dem = dem.astype(dtype) if np.dtype(dtype).kind == "f" else np.round(np.clip(dem, 0, 250 if dtype == np.uint8 else 4000)).astype(dtype)
    x0, y0, sx, sy = 5.75, 47.25, 1 / 120, 1 / 120                # upper-left CORNER, pixel size
    info = {33550: (sx, sy, 0.0), 33922: (0.0, 0.0, 0.0, x0, y0, 0.0)}
    if predictor != 1:
        info[317] = predictor
    path = tmp_path / "dem.tif"
    Image.fromarray(dem).save(str(path), compression=compression, tiffinfo=info)
    ds = read_geotiff(path)
    band = ds["band_data"]
    assert band.shape == (1, H, W) and band.dtype == np.dtype(dtype)
    np.testing.assert_array_equal(band[0], dem)
    np.testing.assert_allclose(ds.coords["x"], x0 + (np.arange(W) + 0.5) * sx, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ds.coords["y"], y0 - (np.arange(H) + 0.5) * sy, rtol=0, atol=1e-12)


def test_geotiff_writer_is_readable_by_pillow_and_by_us(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    dem = rng.integers(0, 4000, (130, 77)).astype(np.int16)
    x, y = 6.0 + np.arange(77) * 0.01, 47.0 - np.arange(130) * 0.01
    for comp in ("none", "deflate"):
        path = tmp_path / f"w_{comp}.tif"
        write_geotiff(path, dem, x, y, compression=comp, nodata=-9999)
        np.testing.assert_array_equal(np.asarray(Image.open(str(path))), dem)
        ds = open_raster(path)
        np.testing.assert_array_equal(ds["band_data"][0], dem)
        np.testing.assert_allclose(ds.coords["x"], x, atol=1e-12)
        np.testing.assert_allclose(ds.coords["y"], y, atol=1e-12)
        assert ds.attrs["nodatavals"] == -9999.0


def _synthetic_inputs(rng, n_lon, n_lat, hours):
    lon = 6.0 + 0.25 * np.arange(n_lon)
    lat = 47.0 - 0.25 * np.arange(n_lat)                          # descending, as ERA5
    time = np.datetime64("2020-12-03T00:00:00") + np.arange(hours).astype("timedelta64[h]")
    era5 = GridDataset({"time": time, "latitude": lat, "longitude": lon},
                       {v: (("time", "latitude", "longitude"), rng.standard_normal((hours, n_lat, n_lon)).astype(np.float32) * 5)
                        for v in ("u10", "v10", "t2m")})
    x = np.arange(lon.min() - 0.1, lon.max() + 0.1, 0.004)
    y = np.arange(lat.max() + 0.1, lat.min() - 0.1, -0.004)
    dem = (1500 + 800 * np.sin(x[None, :] * 9) * np.cos(y[:, None] * 7)).astype(np.float32)
    return era5, x, y, dem


def _brute_nearest(coord, target):
    # ties -> larger coordinate value (pandas); generic random grids have none
    return np.abs(np.asarray(coord)[None, :] - np.asarray(target)[:, None]).argmin(1)


def test_template_and_regridding_follow_the_reference(tmp_path):
    from downscaling.api import build_high_res_template_from_era5, process_era5, process_topo
    rng = np.random.default_rng(4)
    era5, x, y, dem = _synthetic_inputs(rng, 7, 5, 3)
    raster = GridDataset({"band": [1], "y": y, "x": x}, {"band_data": (("band", "y", "x"), dem[None])})
    tpl = build_high_res_template_from_era5(era5)
    assert set(tpl.coords) == {"time", "lon_1", "lat_1"}
    np.testing.assert_allclose(tpl.coords["lon_1"], np.linspace(6.0, 7.5, 18 * 7))
    np.testing.assert_allclose(tpl.coords["lat_1"], np.linspace(46.0, 47.0, 26 * 5))        # ascending: min -> max
    # a sub-range: the count is the number of ERA5 points INSIDE it (api.py:51-59)
    sub = build_high_res_template_from_era5(era5, range_lon=(6.2, 7.1), range_lat=(46.2, 46.8))
    np.testing.assert_allclose(sub.coords["lon_1"], np.linspace(6.2, 7.1, 18 * 4))           # 6.25, 6.5, 6.75, 7.0
    np.testing.assert_allclose(sub.coords["lat_1"], np.linspace(46.2, 46.8, 26 * 3))         # 46.75, 46.5, 46.25
    winds = process_era5(era5, sub)
    assert set(winds.variables) == {"u10", "v10"} and winds.dims_of("u10") == ("time", "lat_1", "lon_1")
    li, lj = _brute_nearest(era5.coords["latitude"], sub.coords["lat_1"]), _brute_nearest(era5.coords["longitude"], sub.coords["lon_1"])
    np.testing.assert_array_equal(winds["u10"], era5["u10"][:, li][:, :, lj])
    topo = process_topo(raster, sub)
    assert topo.dims_of("elevation") == ("lat_1", "lon_1")
    np.testing.assert_array_equal(topo["elevation"], dem[_brute_nearest(y, sub.coords["lat_1"])][:, _brute_nearest(x, sub.coords["lon_1"])])


def test_cli_end_to_end_on_the_oracle_backend(tmp_path, monkeypatch):
    """`downscale --era DIR --dem FILE --date YYYYMMDD [--lon a:b] [--lat a:b] -o out` (cli.py:10-17) from files to
    file, with small patched constants on the CPU backend; the written field equals predict_array on the regridded
    inputs built independently here."""
    from downscaling.engine import runtime
    from oracle.torch_backend import TorchOps
    import downscaling.api as api
    from downscaling import cli
    runtime.set_ops(TorchOps(torch.float64))
    try:
        for k, v in dict(IMG_SIZE=20, SEQUENCE_LENGTH=2, NOISE_CHANNELS=5, BATCH_SIZE=2).items():
            monkeypatch.setattr(api, k, v)
        monkeypatch.setenv("DOWNSCALING_ALLOW_RANDOM_INIT", "1")
        monkeypatch.setenv("DOWNSCALING_RANDOM_SEED", "31")
        rng = np.random.default_rng(5)
        era5, x, y, dem = _synthetic_inputs(rng, 3, 2, 4)
        (tmp_path / "era").mkdir()
        write_netcdf(era5, tmp_path / "era" / "20201203_era5_surface_hourly.nc")
        write_geotiff(tmp_path / "dem.tif", dem, x, y)
        out_path = tmp_path / "downscaled.nc"
        assert cli.main(["--era", str(tmp_path / "era"), "--dem", str(tmp_path / "dem.tif"), "--date", "20201203",
                         "-o", str(out_path)]) == 0
        got = open_dataset(out_path)
        assert got.dims_of("u10") == ("time", "lat_1", "lon_1") and set(got.variables) == {"u10", "v10"}
        # independent restatement of the wrappers: linspace template, brute-force nearest regridding, predict_array
        lon1, lat1 = np.linspace(6.0, 6.5, 18 * 3), np.linspace(46.75, 47.0, 26 * 2)
        li, lj = _brute_nearest(era5.coords["latitude"], lat1), _brute_nearest(era5.coords["longitude"], lon1)
        elev = dem[_brute_nearest(y, lat1)][:, _brute_nearest(x, lon1)]
        fields = np.stack([era5["u10"][:, li][:, :, lj], era5["v10"][:, li][:, :, lj],
                           np.broadcast_to(elev[None], (4, len(lat1), len(lon1)))], -1)
        network = api.get_network(random_seed=31)
        want, cnt = api.predict_array(fields, overlap_factor=cli.OVERLAP_FACTOR, network=network, return_count=True)
        keep_lat, keep_lon = cnt[0].any(1), cnt[0].any(0)
        want = want[:, keep_lat][:, :, keep_lon]
        np.testing.assert_allclose(got.coords["lat_1"], lat1[keep_lat])
        np.testing.assert_allclose(got.coords["lon_1"], lon1[keep_lon])
        assert got["u10"].shape == want[..., 0].shape
        np.testing.assert_allclose(got["u10"], want[..., 0], rtol=1e-6, atol=1e-6, equal_nan=True)
        np.testing.assert_allclose(got["v10"], want[..., 1], rtol=1e-6, atol=1e-6, equal_nan=True)
        assert got.coords["time"][0] == np.datetime64("2020-12-03T00:00:00")
        # reference flags, reference defaults
        assert [n for n, _ in cli.FLAGS] == [("--era",), ("--dem",), ("--date",), ("--lon",), ("--lat",), ("-o", "--output")]
        assert dict(cli.FLAGS)[("-o", "--output")]["default"] == "downscaled.nc" and cli.OVERLAP_FACTOR == 0.01
        with pytest.raises(FileNotFoundError):
            cli.main(["--era", str(tmp_path / "era"), "--dem", str(tmp_path / "dem.tif"), "--date", "19990101"])
    finally:
        runtime.set_ops(None)
