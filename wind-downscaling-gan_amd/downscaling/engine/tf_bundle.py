"""Reader for TensorFlow tensor-bundle checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-00001`),
the format of the reference's weights-55.ckpt (/root/reference/src/downscaling/gan/ganbase.py:132-140,
api.py:21,85).  The index is a LevelDB-style SSTable: 48-byte footer (metaindex + index block handles,
magic), prefix-compressed key blocks with a restart array, values = BundleEntryProto
{1: dtype, 2: shape, 3: shard_id, 4: offset, 5: size, 6: crc32c}.  Pure host code (numpy)."""
import struct

import numpy as np

_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


def _varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _block(data, off, size):
    blk = data[off:off + size]
    nrestart = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * nrestart
    i, key, out = 0, b"", []
    while i < end:
        shared, i = _varint(blk, i)
        nonshared, i = _varint(blk, i)
        vlen, i = _varint(blk, i)
        key = key[:shared] + blk[i:i + nonshared]
        i += nonshared
        out.append((key, blk[i:i + vlen]))
        i += vlen
    return out


def _proto(b):
    i, f = 0, {}
    while i < len(b):
        tag, i = _varint(b, i)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 2:
            n, i = _varint(b, i)
            v = b[i:i + n]
            i += n
        elif wt == 5:
            v = struct.unpack("<I", b[i:i + 4])[0]
            i += 4
        elif wt == 1:
            v = struct.unpack("<Q", b[i:i + 8])[0]
            i += 8
        else:
            raise ValueError(f"unsupported wire type {wt}")
        f.setdefault(fn, []).append(v)
    return f


def read_index(index_path):
    """-> [(key, dtype enum, shape, shard, offset, size)] for every tensor of the bundle."""
    data = open(index_path, "rb").read()
    footer = data[-48:]
    i = 0
    _, i = _varint(footer, i)
    _, i = _varint(footer, i)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)
    entries = []
    for _, handle in _block(data, ioff, isize):
        j = 0
        boff, j = _varint(handle, j)
        bsize, j = _varint(handle, j)
        entries += _block(data, boff, bsize)
    out = []
    for key, val in entries:
        if key == b"":
            continue  # BundleHeaderProto
        p = _proto(val)
        shape = []
        if 2 in p:
            for d in _proto(p[2][0]).get(2, []):
                shape.append(_proto(d).get(1, [0])[0])
        out.append((key.decode(), p.get(1, [0])[0], tuple(shape), p.get(3, [0])[0], p.get(4, [0])[0], p.get(5, [0])[0]))
    return out


def read_bundle(prefix, include_optimizer_slots=False):
    """{variable name: numpy array} (names stripped of '/.ATTRIBUTES/VARIABLE_VALUE')."""
    prefix = str(prefix)
    entries = read_index(prefix + ".index")
    out = {}
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        for key, dt, shape, shard, off, size in entries:
            if not key.endswith(_SUFFIX) or dt not in _DTYPES:
                continue
            if ".OPTIMIZER_SLOT" in key and not include_optimizer_slots:
                continue
            f.seek(off)
            arr = np.frombuffer(f.read(size), dtype=_DTYPES[dt]).reshape(shape)
            out[key[:-len(_SUFFIX)]] = arr
    return out
