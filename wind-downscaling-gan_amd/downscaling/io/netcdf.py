"""Day files of the `downscale` CLI (/root/reference/src/downscaling/cli.py:22-26: `xr.open_mfdataset` of
`<date>*surface*.nc`, `Dataset.to_netcdf`).  netCDF4 / h5py are not part of the GPU image; what is there is scipy's
NetCDF-3 reader / writer — NetCDF-3 (classic / 64-bit offset, int16-packed with scale_factor / add_offset) is what the
Climate Data Store served for ERA5 single-level requests when the reference was written — and this package's own read-only
HDF5 parser for the NetCDF-4 files it serves now (io/hdf5.py).  So:

  * `.nc` files are read as NetCDF-3 through `scipy.io.netcdf_file` or as NetCDF-4 through `io.hdf5` (by magic number; CF
    packing, fill values and "<unit> since <epoch>" time axes decoded as xarray's `decode_cf` does) and written as NetCDF-3;
  * `.npz` files hold the same content as named arrays (coordinates + variables + a `__dims__` table).
"""
import json
import re
from pathlib import Path

import numpy as np

from .grid import GridDataset

_UNITS = {"seconds": "s", "second": "s", "minutes": "m", "minute": "m", "hours": "h", "hour": "h", "days": "D", "day": "D"}


def _decode_time(values, units):
    m = re.match(r"\s*(\w+)\s+since\s+(\d{4}-\d{1,2}-\d{1,2})(?:[T ](\d{1,2}):(\d{2})(?::(\d{2})(?:\.\d*)?)?)?", str(units))
    if not m or m.group(1).lower() not in _UNITS:
        return np.asarray(values)
    y, mo, d = (int(x) for x in m.group(2).split("-"))
    epoch = np.datetime64(f"{y:04d}-{mo:02d}-{d:02d}T{int(m.group(3) or 0):02d}:{int(m.group(4) or 0):02d}:{int(m.group(5) or 0):02d}", "s")
    unit = _UNITS[m.group(1).lower()]
    vals = np.asarray(values)
    if np.issubdtype(vals.dtype, np.integer):
        return (epoch + vals.astype("int64").astype(f"timedelta64[{unit}]")).astype("datetime64[s]")
    secs = {"s": 1, "m": 60, "h": 3600, "D": 86400}[unit]
    return (epoch + np.round(vals.astype(np.float64) * secs).astype("int64").astype("timedelta64[s]")).astype("datetime64[s]")


def _attr(var, name):
    v = getattr(var, name, None)
    if v is None:
        return None
    return v.decode() if isinstance(v, bytes) else v


def _cf_decode(raw, attrs):
    """xarray's `decode_cf` for one variable: _FillValue / missing_value -> NaN, scale_factor / add_offset (float32 for packed
    data of <= 2 bytes, else float64)."""
    fill = [attrs[a] for a in ("_FillValue", "missing_value") if attrs.get(a) is not None]
    scale, offset = attrs.get("scale_factor"), attrs.get("add_offset")
    packed = scale is not None or offset is not None
    data = raw.astype(np.float32 if (packed and raw.dtype.itemsize <= 2) or raw.dtype == np.float32 else np.float64) \
        if (packed or fill) and raw.dtype.kind in "iuf" else raw
    if fill and data.dtype.kind == "f":
        mask = np.zeros(raw.shape, dtype=bool)
        for fv in fill:
            fv = np.asarray(fv).reshape(-1)[0]
            mask |= np.isnan(raw) if (raw.dtype.kind == "f" and np.isnan(fv)) else (raw == fv)
        data = np.where(mask, np.nan, data).astype(data.dtype)
    if packed:
        data = data * data.dtype.type(1.0 if scale is None else np.asarray(scale).reshape(-1)[0]) \
            + data.dtype.type(0.0 if offset is None else np.asarray(offset).reshape(-1)[0])
    return data


def _read_netcdf4(path):
    """HDF5-based NetCDF-4 (what the Climate Data Store serves today) through the package's own HDF5 reader (io/hdf5.py)."""
    from .hdf5 import read_netcdf4
    raw_coords, raw_vars, gattrs = read_netcdf4(path)
    coords = {}
    for name, (values, attrs) in raw_coords.items():
        units = attrs.get("units")
        coords[name] = _decode_time(values, units) if units and " since " in str(units) else values
    variables = {name: (dims, _cf_decode(data, attrs)) for name, (dims, data, attrs) in raw_vars.items()}
    return GridDataset(coords, variables, {k: v for k, v in gattrs.items() if isinstance(v, (str, int, float))})


def read_netcdf(path):
    """One NetCDF file -> GridDataset with CF-decoded variables (float32 for packed ones, as xarray): NetCDF-3 (classic /
    64-bit offset) through scipy, NetCDF-4 through io/hdf5.py."""
    from scipy.io import netcdf_file
    path = str(path)
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic == b"\x89HDF":
        return _read_netcdf4(path)
    if magic[:3] != b"CDF":
        raise OSError(f"{path}: neither a NetCDF-3 (CDF) nor a NetCDF-4 (HDF5) file")
    coords, variables = {}, {}
    with netcdf_file(path, "r", mmap=False, maskandscale=False) as nc:
        for name, var in nc.variables.items():
            raw = np.array(var[...])
            dims = tuple(var.dimensions)
            if dims == (name,):
                units = _attr(var, "units")
                coords[name] = _decode_time(raw, units) if units and " since " in str(units) else raw
                continue
            attrs = {a: getattr(var, a) for a in ("_FillValue", "missing_value", "scale_factor", "add_offset") if hasattr(var, a)}
            variables[name] = (dims, _cf_decode(raw, attrs))
        attrs = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in nc._attributes.items()}
    return GridDataset(coords, variables, attrs)


def write_netcdf(ds, path):
    """GridDataset -> NetCDF-3 (64-bit offset).  datetime64 coordinates are written as `hours since 1900-01-01`
    (the ERA5 convention), float variables as float32."""
    from scipy.io import netcdf_file
    with netcdf_file(str(path), "w", version=2) as nc:
        for k, v in ds.attrs.items():
            if isinstance(v, (str, int, float)):
                setattr(nc, k, v)
        for name, c in ds.coords.items():
            c = np.asarray(c)
            nc.createDimension(name, len(c))
            if np.issubdtype(c.dtype, np.datetime64):
                hours = (c.astype("datetime64[s]") - np.datetime64("1900-01-01T00:00:00", "s")).astype("int64") / 3600.0
                exact = np.all(hours == np.round(hours))
                var = nc.createVariable(name, "i4" if exact else "f8", (name,))
                var[:] = hours.astype("int32" if exact else "float64")
                var.units = "hours since 1900-01-01 00:00:00.0"
                var.calendar = "gregorian"
            else:
                c = c.astype(np.float64 if c.dtype.kind == "f" else np.int32)
                var = nc.createVariable(name, "f8" if c.dtype.kind == "f" else "i4", (name,))
                var[:] = c
        for name, (dims, arr) in ds.variables.items():
            for d, n in zip(dims, arr.shape):
                if d not in nc.dimensions:
                    nc.createDimension(d, n)
            arr = np.asarray(arr)
            arr = arr.astype(np.float32) if arr.dtype.kind == "f" else arr.astype(np.int32)
            var = nc.createVariable(name, "f4" if arr.dtype.kind == "f" else "i4", dims)
            var[...] = arr


def read_npz(path):
    with np.load(str(path), allow_pickle=False) as z:
        names = list(z.files)
        dims_table = json.loads(str(z["__dims__"])) if "__dims__" in names else {}
        arrays = {k: z[k] for k in names if k != "__dims__"}
    coords = {k: v for k, v in arrays.items() if v.ndim == 1 and k not in dims_table}
    variables = {}
    for k, v in arrays.items():
        if k in coords:
            continue
        dims = dims_table.get(k)
        if dims is None:        # plain savez without a table: the ERA5 axis order
            dims = {3: ("time", "latitude", "longitude"), 2: ("y", "x")}.get(v.ndim)
            if dims is None:
                raise ValueError(f"{path}: no dimension names for {k!r} (add a __dims__ JSON table)")
        variables[k] = (tuple(dims), v)
    return GridDataset(coords, variables)


def write_npz(ds, path):
    table = {k: list(d) for k, (d, _) in ds.variables.items()}
    np.savez(str(path), __dims__=np.array(json.dumps(table)), **ds.coords, **{k: a for k, (_, a) in ds.variables.items()})


def open_dataset(path):
    path = Path(path)
    if path.suffix == ".npz":
        return read_npz(path)
    return read_netcdf(path)


def save_dataset(ds, path):
    path = Path(path)
    (write_npz if path.suffix == ".npz" else write_netcdf)(ds, path)


def open_mfdataset(paths, concat_dim="time"):
    """Several day / hour files of one request combined by coordinates along `concat_dim` (what
    `xr.open_mfdataset(paths)` does for files that differ in their time axis only)."""
    parts = [open_dataset(p) for p in sorted(map(str, paths))]
    if not parts:
        raise FileNotFoundError("no input files")
    if len(parts) == 1:
        return parts[0]
    parts.sort(key=lambda d: d.coords[concat_dim][0])
    first = parts[0]
    for p in parts[1:]:
        for k, c in first.coords.items():
            if k != concat_dim and not np.array_equal(c, p.coords[k]):
                raise ValueError(f"files disagree on coordinate {k!r}")
    coords = dict(first.coords)
    coords[concat_dim] = np.concatenate([p.coords[concat_dim] for p in parts])
    variables = {}
    for name, (dims, arr) in first.variables.items():
        if concat_dim in dims:
            variables[name] = (dims, np.concatenate([p.variables[name][1] for p in parts], axis=dims.index(concat_dim)))
        else:
            variables[name] = (dims, arr)
    order = np.argsort(coords[concat_dim], kind="stable")
    return GridDataset(coords, variables, first.attrs).isel(**{concat_dim: order})
