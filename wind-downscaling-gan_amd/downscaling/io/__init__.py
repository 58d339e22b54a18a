"""File and grid I/O of the `downscale` driver without xarray / netCDF4 / rasterio (absent from the GPU image):
GridDataset (labelled numpy arrays), NetCDF-3 and .npz day files, GeoTIFF DEM rasters."""
from .geotiff import read_geotiff, write_geotiff
from .grid import GridDataset, nearest_index
from .netcdf import open_dataset, open_mfdataset, read_netcdf, read_npz, save_dataset, write_netcdf, write_npz


def open_raster(path):
    """The DEM file of `downscale --dem`: a GeoTIFF (the reference's `xr.open_rasterio`), or a .npz / NetCDF-3 file
    holding `x`, `y` and one 2-D (y, x) or 3-D (band, y, x) array."""
    from pathlib import Path
    import numpy as np
    path = Path(path)
    if path.suffix.lower() in (".tif", ".tiff", ".gtiff"):
        return read_geotiff(path)
    ds = open_dataset(path)
    name = next(k for k, (d, a) in ds.variables.items() if a.ndim in (2, 3))
    dims, arr = ds.variables[name]
    if arr.ndim == 2:
        dims, arr = ("band",) + tuple(dims), arr[None]
    coords = dict(ds.coords)
    coords.setdefault("band", np.arange(1, arr.shape[0] + 1))
    return GridDataset(coords, {"band_data": (dims, arr)}, ds.attrs)


__all__ = ["GridDataset", "nearest_index", "open_dataset", "open_mfdataset", "open_raster", "read_geotiff",
           "read_netcdf", "read_npz", "save_dataset", "write_geotiff", "write_netcdf", "write_npz"]
