"""DEM raster input of the `downscale` CLI (/root/reference/src/downscaling/cli.py:23: `xr.open_rasterio(args.dem)`;
api.py:31-37 reads band 0 with its `x` / `y` pixel-centre coordinates).  rasterio / GDAL are not part of the GPU image,
so this is a self-contained baseline-TIFF + GeoTIFF-tag reader: classic TIFF and BigTIFF, either byte order, strips or
tiles, chunky or planar samples, compression none / LZW / Deflate / PackBits, predictors 1-3, 8/16/32/64-bit integer and
float samples; georeferencing from ModelPixelScale + ModelTiepoint or ModelTransformation (north-up), the
PixelIsPoint raster type, and GDAL's nodata tag.  `write_geotiff` produces files of the same kind (tests, conversion).
"""
import struct
import zlib

import numpy as np

from .grid import GridDataset

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2),
          9: ("i", 4), 10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 13: ("I", 4), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}


def _read_ifd(buf, bo, big):
    if big:
        (off,) = struct.unpack_from(bo + "Q", buf, 8)
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        pos, esz, cfmt, inline = off + 8, 20, "Q", 8
    else:
        (off,) = struct.unpack_from(bo + "I", buf, 4)
        (n,) = struct.unpack_from(bo + "H", buf, off)
        pos, esz, cfmt, inline = off + 2, 12, "I", 4
    tags = {}
    for i in range(n):
        e = pos + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        (count,) = struct.unpack_from(bo + cfmt, buf, e + 4)
        if typ not in _TYPES:
            continue
        fmt, size = _TYPES[typ]
        voff = e + 4 + struct.calcsize(cfmt)
        if size * count > inline:
            (voff,) = struct.unpack_from(bo + cfmt, buf, voff)
        if typ == 2:
            tags[tag] = bytes(buf[voff:voff + count]).split(b"\0")[0].decode("latin1")
        else:
            vals = struct.unpack_from(bo + fmt * count, buf, voff)
            if typ in (5, 10):
                vals = tuple(vals[j] / vals[j + 1] if vals[j + 1] else 0.0 for j in range(0, len(vals), 2))
            tags[tag] = vals
    return tags


def _lzw_decode(data):
    """TIFF LZW (MSB-first codes, 9..12 bits, ClearCode 256, EndOfInformation 257, 'early change')."""
    out = bytearray()
    table = [bytes((i,)) for i in range(256)] + [b"", b""]
    bits, nbits, width = 0, 0, 9
    prev = None
    for byte in data:
        bits = (bits << 8) | byte
        nbits += 8
        while nbits >= width:
            code = (bits >> (nbits - width)) & ((1 << width) - 1)
            nbits -= width
            bits &= (1 << nbits) - 1
            if code == 256:
                table = table[:258]
                width, prev = 9, None
                continue
            if code == 257:
                return bytes(out)
            if prev is None:
                entry = table[code]
            elif code < len(table):
                entry = table[code]
                table.append(prev + entry[:1])
            else:
                entry = prev + prev[:1]
                table.append(entry)
            out += entry
            prev = entry
            if len(table) >= (1 << width) - 1 and width < 12:
                width += 1
    return bytes(out)


def _packbits_decode(data):
    out, i = bytearray(), 0
    while i < len(data):
        n = data[i]
        i += 1
        if n < 128:
            out += data[i:i + n + 1]
            i += n + 1
        elif n > 128:
            out += data[i:i + 1] * (257 - n)
            i += 1
    return bytes(out)


def _decompress(chunk, compression):
    if compression == 1:
        return bytes(chunk)
    if compression in (8, 32946):
        return zlib.decompress(bytes(chunk))
    if compression == 5:
        return _lzw_decode(bytes(chunk))
    if compression == 32773:
        return _packbits_decode(bytes(chunk))
    raise NotImplementedError(f"TIFF compression {compression} (supported: none, LZW, Deflate, PackBits)")


def _sample_dtype(bits, fmt, bo):
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bits not in (8, 16, 32, 64) or (kind == "f" and bits < 32):
        raise NotImplementedError(f"TIFF sample format {fmt} with {bits} bits")
    return np.dtype(f"{'<' if bo == '<' else '>'}{kind}{bits // 8}")


def _unpredict(block, predictor, dtype, spp):
    """block: [rows, cols, spp] raw samples of one strip / tile."""
    if predictor == 1:
        return block
    if predictor == 2:
        return np.cumsum(block.astype(dtype.newbyteorder("=")), axis=1, dtype=dtype.newbyteorder("="))
    raise NotImplementedError(f"TIFF predictor {predictor}")


def _unpredict_float(raw, rows, cols, spp, dtype):
    """Predictor 3: bytes of each row are differenced and stored plane-wise, most significant byte first."""
    bps = dtype.itemsize
    rowb = np.frombuffer(raw, dtype=np.uint8, count=rows * cols * spp * bps).reshape(rows, cols * spp * bps)
    rowb = _cumsum_u8(rowb, spp)
    planes = rowb.reshape(rows, bps, cols * spp)                     # byte plane b holds byte (bps-1-b) little-endian
    le = np.ascontiguousarray(planes[:, ::-1, :].transpose(0, 2, 1))  # [rows, cols*spp, bps] little-endian bytes
    return le.view(np.dtype(f"<f{bps}")).reshape(rows, cols, spp)


def _cumsum_u8(rowb, spp):
    out = rowb.copy()
    for s in range(spp, out.shape[1]):
        out[:, s] += out[:, s - spp]
    return out


def read_geotiff(path):
    """-> GridDataset with coords `band`, `y` (pixel-centre latitudes, north to south), `x` (pixel-centre
    longitudes) and one variable `band_data` on (band, y, x) — the layout of `xr.open_rasterio`.  nodata -> attrs."""
    buf = memoryview(open(str(path), "rb").read())
    bo = {b"II": "<", b"MM": ">"}.get(bytes(buf[:2]))
    if bo is None:
        raise OSError(f"{path}: not a TIFF file")
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic not in (42, 43):
        raise OSError(f"{path}: not a TIFF file")
    t = _read_ifd(buf, bo, magic == 43)
    W, H = int(t[256][0]), int(t[257][0])
    spp = int(t.get(277, (1,))[0])
    bits = int(t.get(258, (1,))[0])
    fmt = int(t.get(339, (1,))[0])
    comp = int(t.get(259, (1,))[0])
    pred = int(t.get(317, (1,))[0])
    planar = int(t.get(284, (1,))[0])
    dtype = _sample_dtype(bits, fmt, bo)
    tiled = 322 in t
    if tiled:
        cw, chh = int(t[322][0]), int(t[323][0])
        offsets, counts = t[324], t[325]
    else:
        cw, chh = W, int(t.get(278, (H,))[0])
        chh = min(chh, H)
        offsets, counts = t[273], t[279]
    across, down = -(-W // cw), -(-H // chh)
    planes = spp if planar == 2 else 1
    cspp = 1 if planar == 2 else spp
    out = np.zeros((spp, H, W), dtype=dtype.newbyteorder("="))
    for p in range(planes):
        for j in range(down):
            for i in range(across):
                k = (p * down + j) * across + i
                rows = chh if tiled else min(chh, H - j * chh)
                raw = _decompress(buf[offsets[k]:offsets[k] + counts[k]], comp)
                if pred == 3:
                    block = _unpredict_float(raw, rows, cw, cspp, dtype)
                else:
                    block = np.frombuffer(raw, dtype=dtype, count=rows * cw * cspp).reshape(rows, cw, cspp)
                    block = _unpredict(block, pred, dtype, cspp)
                r0, c0 = j * chh, i * cw
                r1, c1 = min(r0 + rows, H), min(c0 + cw, W)
                blk = block[:r1 - r0, :c1 - c0]
                if planar == 2:
                    out[p, r0:r1, c0:c1] = blk[..., 0]
                else:
                    out[:, r0:r1, c0:c1] = np.moveaxis(blk, 2, 0)
    # ---- georeferencing
    point = False
    gk = t.get(34735)
    if gk:
        for i in range(4, len(gk) - 3, 4):
            if gk[i] == 1025 and gk[i + 1] == 0:
                point = gk[i + 3] == 2
    if 33550 in t and 33922 in t:
        sx, sy = t[33550][0], t[33550][1]
        i0, j0, _, x0, y0 = t[33922][:5]
        ox, oy = x0 - i0 * sx, y0 + j0 * sy
    elif 34264 in t:
        m = t[34264]
        if m[1] != 0 or m[4] != 0:
            raise NotImplementedError("rotated GeoTIFF (ModelTransformation with shear terms)")
        sx, sy, ox, oy = m[0], -m[5], m[3], m[7]
    else:
        sx = sy = 1.0
        ox, oy, point = 0.0, float(H), False
    half = 0.0 if point else 0.5
    x = ox + (np.arange(W) + half) * sx
    y = oy - (np.arange(H) + half) * sy
    attrs = {}
    if 42113 in t:
        try:
            attrs["nodatavals"] = float(t[42113])
        except ValueError:
            pass
    return GridDataset({"band": np.arange(1, spp + 1), "y": y, "x": x}, {"band_data": (("band", "y", "x"), out)}, attrs)


def write_geotiff(path, data, x, y, compression="deflate", nodata=None, rows_per_strip=64):
    """data [H, W] (or [bands, H, W]) with pixel-centre coordinates x (ascending) / y (descending): a north-up,
    PixelIsArea, chunky, stripped little-endian GeoTIFF."""
    data = np.asarray(data)
    if data.ndim == 2:
        data = data[None]
    spp, H, W = data.shape
    dt = data.dtype.newbyteorder("<")
    fmt = {"u": 1, "i": 2, "f": 3}[dt.kind]
    comp = {"none": 1, "deflate": 8}[compression]
    chunky = np.ascontiguousarray(np.moveaxis(data, 0, 2)).astype(dt)
    strips = []
    for r in range(0, H, rows_per_strip):
        raw = chunky[r:r + rows_per_strip].tobytes()
        strips.append(zlib.compress(raw, 6) if comp == 8 else raw)
    sx = float(x[1] - x[0]) if W > 1 else 1.0
    sy = float(y[0] - y[1]) if H > 1 else 1.0
    entries = []                                    # (tag, type, values)

    def add(tag, typ, vals):
        entries.append((tag, typ, list(vals) if not isinstance(vals, (str, bytes)) else vals))
    add(256, 4, [W]); add(257, 4, [H]); add(258, 3, [dt.itemsize * 8] * spp); add(259, 3, [comp]); add(262, 3, [1])
    add(273, 4, [0] * len(strips)); add(277, 3, [spp]); add(278, 4, [rows_per_strip]); add(279, 4, [len(s) for s in strips])
    add(284, 3, [1]); add(339, 3, [fmt] * spp)
    add(33550, 12, [sx, sy, 0.0]); add(33922, 12, [0.0, 0.0, 0.0, float(x[0]) - 0.5 * sx, float(y[0]) + 0.5 * sy, 0.0])
    add(34735, 3, [1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 1, 2048, 0, 1, 4326])
    if nodata is not None:
        add(42113, 2, repr(float(nodata)))
    entries.sort(key=lambda e: e[0])
    head = 8 + 2 + 12 * len(entries) + 4
    blobs, pos = [], head

    def place(b):
        nonlocal pos
        off = pos
        blobs.append(b + (b"\0" if len(b) & 1 else b""))
        pos += len(blobs[-1])
        return off
    packed = {}
    for tag, typ, vals in entries:
        if tag == 273:
            continue
        if typ == 2:
            b = vals.encode() + b"\0"
            packed[tag] = (len(b), b)
        else:
            f = _TYPES[typ][0]
            packed[tag] = (len(vals), struct.pack("<" + f * len(vals), *vals))
    offs = {}
    for tag, (cnt, b) in packed.items():
        if len(b) > 4:
            offs[tag] = place(b)
    strip_table_off = place(b"\0" * 4 * len(strips)) if len(strips) > 1 else None
    strip_offs = [place(s) for s in strips]
    packed[273] = (len(strips), struct.pack("<" + "I" * len(strips), *strip_offs))
    ifd = struct.pack("<H", len(entries))
    for tag, typ, vals in entries:
        cnt, b = packed[tag]
        if len(b) <= 4:
            field = b + b"\0" * (4 - len(b))
        elif tag == 273:
            field = struct.pack("<I", strip_table_off)
        else:
            field = struct.pack("<I", offs[tag])
        ifd += struct.pack("<HHI", tag, typ, cnt) + field
    ifd += struct.pack("<I", 0)
    body = b"".join(blobs)
    if strip_table_off is not None:
        rel = strip_table_off - head
        body = body[:rel] + packed[273][1] + body[rel + 4 * len(strips):]
    with open(str(path), "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, 8) + ifd + body)
