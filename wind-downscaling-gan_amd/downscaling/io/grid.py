"""A labelled-array container for the inference driver: what the reference keeps in `xarray.Dataset` objects
(/root/reference/src/downscaling/api.py:31-62,89-160) — named 1-D coordinates plus variables on named dimensions —
on plain numpy, because xarray / netCDF4 / rasterio are not part of the GPU image.  Nearest-neighbour selection goes
through `pandas.Index.get_indexer(method="nearest")`, the routine xarray's `.sel(..., method="nearest")` calls, so
tie-breaking is the reference's."""
import numpy as np
import pandas as pd


class GridDataset:
    """coords: {name: 1-D array}; variables: {name: (dims, array)} with `dims` a tuple of coordinate names."""

    def __init__(self, coords=None, variables=None, attrs=None):
        self.coords = {k: np.asarray(v) for k, v in (coords or {}).items()}
        self.variables = {}
        self.attrs = dict(attrs or {})
        for name, (dims, arr) in (variables or {}).items():
            self[name] = (dims, arr)

    # ---- mapping surface ---------------------------------------------------------------------------------------------
    def __setitem__(self, name, value):
        dims, arr = value
        dims, arr = tuple(dims), np.asarray(arr)
        if arr.ndim != len(dims):
            raise ValueError(f"{name}: {arr.ndim}-D array for dims {dims}")
        for d, n in zip(dims, arr.shape):
            if d in self.coords and len(self.coords[d]) != n:
                raise ValueError(f"{name}: axis {d!r} has {n} entries, the coordinate {len(self.coords[d])}")
        self.variables[name] = (dims, arr)

    def __getitem__(self, name):
        return self.variables[name][1] if name in self.variables else self.coords[name]

    def __contains__(self, name):
        return name in self.variables or name in self.coords

    def dims_of(self, name):
        return self.variables[name][0]

    @property
    def dims(self):
        out = {k: len(v) for k, v in self.coords.items()}
        for dims, arr in self.variables.values():
            out.update(zip(dims, arr.shape))
        return out

    def get(self, name, default=None):
        return self[name] if name in self else default

    def copy(self):
        return GridDataset(self.coords, {k: (d, a.copy()) for k, (d, a) in self.variables.items()}, self.attrs)

    def __repr__(self):
        vs = ", ".join(f"{k}{list(d)}" for k, (d, _) in self.variables.items())
        return f"GridDataset(dims={self.dims}, variables=[{vs}])"

    # ---- selection ---------------------------------------------------------------------------------------------------
    def transposed(self, name, *dims):
        have, arr = self.variables[name]
        return np.transpose(arr, [have.index(d) for d in dims])

    def isel(self, **indexers):
        """Positional selection along named dimensions (slices or integer arrays)."""
        coords = {k: (v[indexers[k]] if k in indexers else v) for k, v in self.coords.items()}
        variables = {}
        for name, (dims, arr) in self.variables.items():
            key = tuple(indexers.get(d, slice(None)) for d in dims)
            for ax, k in enumerate(key):          # one axis at a time: integer arrays must not broadcast together
                if not isinstance(k, slice) or k != slice(None):
                    arr = arr[(slice(None),) * ax + (k,)]
            variables[name] = (dims, arr)
        return GridDataset(coords, variables, self.attrs)

    def sel_range(self, dim, lo, hi):
        """`.sel(dim=slice(lo, hi))` on a monotonic coordinate: label-based, both ends inclusive, in the
        coordinate's own direction (ERA5 latitudes descend: the reference passes slice(max, min), api.py:57)."""
        c = self.coords[dim]
        keep = (c >= min(lo, hi)) & (c <= max(lo, hi))
        ascending = len(c) < 2 or c[-1] >= c[0]
        if (lo <= hi) != ascending and lo != hi:
            keep[:] = False                       # a slice against the coordinate's direction selects nothing
        return self.isel(**{dim: np.nonzero(keep)[0]})

    def sel_nearest(self, rename=None, **targets):
        """`.sel(a=new_a, b=new_b, method="nearest")` followed by dropping the old coordinates: every listed
        dimension is re-sampled at the positions nearest to the target values and takes the target's name
        (`rename[dim]`) and values."""
        rename = rename or {}
        out = self
        for dim, target in targets.items():
            target = np.asarray(target, dtype=np.float64)
            idx = nearest_index(out.coords[dim], target)
            out = out.isel(**{dim: idx})
            new = rename.get(dim, dim)
            coords = {(new if k == dim else k): (target if k == dim else v) for k, v in out.coords.items()}
            variables = {k: (tuple(new if d == dim else d for d in dims), a) for k, (dims, a) in out.variables.items()}
            out = GridDataset(coords, variables, out.attrs)
        return out

    # ---- interchange -------------------------------------------------------------------------------------------------
    @classmethod
    def from_xarray(cls, ds):
        """Accept the reference's own container when the caller has xarray (duck-typed, not imported here)."""
        if isinstance(ds, cls):
            return ds
        if hasattr(ds, "data_vars"):
            coords = {str(k): np.asarray(ds.coords[k].values) for k in ds.coords if ds.coords[k].ndim == 1}
            variables = {str(k): (tuple(map(str, ds[k].dims)), np.asarray(ds[k].values)) for k in ds.data_vars}
            return cls(coords, variables, dict(getattr(ds, "attrs", {})))
        if hasattr(ds, "dims") and hasattr(ds, "values"):                       # a DataArray (the DEM raster)
            coords = {str(k): np.asarray(ds.coords[k].values) for k in ds.coords if ds.coords[k].ndim == 1}
            return cls(coords, {str(ds.name or "band_data"): (tuple(map(str, ds.dims)), np.asarray(ds.values))})
        raise TypeError(f"cannot read a gridded dataset from {type(ds).__name__}")

    def to_netcdf(self, path):
        """`Dataset.to_netcdf(path)` of the reference's CLI (cli.py:26): NetCDF-3, or .npz by extension."""
        from .netcdf import save_dataset
        save_dataset(self, path)

    def to_xarray(self):
        import xarray as xr
        return xr.Dataset({k: (d, a) for k, (d, a) in self.variables.items()}, coords=self.coords, attrs=self.attrs)


def nearest_index(coord, target):
    """Positions in the monotonic 1-D `coord` nearest to each `target` value — pandas' indexer, which is what
    xarray's method="nearest" resolves to (ties go to the larger index value)."""
    coord = np.asarray(coord, dtype=np.float64)
    index = pd.Index(coord)
    if not (index.is_monotonic_increasing or index.is_monotonic_decreasing):
        raise ValueError("nearest-neighbour selection needs a monotonic coordinate")
    return np.asarray(index.get_indexer(np.asarray(target, dtype=np.float64), method="nearest"), dtype=np.int64)
