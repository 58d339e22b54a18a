"""`downscale` console entry point: the reference's flags (/root/reference/src/downscaling/cli.py:10-17) on the
file readers of `downscaling.io` — the day's ERA5 surface files `<era>/<date>*surface*.nc` (NetCDF-3) or `.npz`, a
GeoTIFF / .npz DEM, output as NetCDF-3 or `.npz` by extension.  overlap_factor is the reference's 0.01 (cli.py:24)."""
import argparse
import sys
from pathlib import Path

FLAGS = (
    (('--era',), dict(help='path to folder with ERA5 data', required=True)),
    (('--dem',), dict(help='path to DEM data file', required=True)),
    (('--date',), dict(help='date to downscale in YYYYMMDD format', required=True)),
    (('--lon',), dict(default=None, help='longitude range (ex: 45.6:46.2)')),
    (('--lat',), dict(default=None, help='latitude range (ex: 45.6:46.2)')),
    (('-o', '--output'), dict(help='output path for the downscaled map (*.nc)', default='downscaled.nc')),
)
OVERLAP_FACTOR = 0.01


def _range(text):
    if not text:
        return None
    lo, hi = (float(part) for part in text.split(':'))
    return lo, hi


def era5_files(folder, date):
    """The day's surface files, NetCDF first; `.npz` files of the same stem pattern stand in where the data was
    converted with numpy."""
    folder = Path(folder)
    for ext in ('nc', 'npz'):
        found = sorted(folder.glob(f'{date}*surface*.{ext}'))
        if found:
            return found
    raise FileNotFoundError(f'no {date}*surface*.nc (or .npz) file in {folder}')


def main(argv=None):
    parser = argparse.ArgumentParser(description='Downscale ER5 wind fields')
    for names, options in FLAGS:
        parser.add_argument(*names, **options)
    args = parser.parse_args(argv)

    from downscaling import downscale
    from downscaling.io import open_mfdataset, open_raster
    era5 = open_mfdataset(era5_files(args.era, args.date))
    raster_topo = open_raster(args.dem)
    downscaled_maps = downscale(era5, raster_topo, range_lon=_range(args.lon), range_lat=_range(args.lat),
                                overlap_factor=OVERLAP_FACTOR)
    downscaled_maps.to_netcdf(args.output)
    return 0


if __name__ == '__main__':
    sys.exit(main())
