"""Reader and writer for TensorFlow tensor-bundle checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-00001`),
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


def read_num_shards(index_path):
    """BundleHeaderProto.num_shards (field 1) of the entry with the empty key; 1 when absent."""
    data = open(index_path, "rb").read()
    footer = data[-48:]
    i = 0
    _, i = _varint(footer, i)
    _, i = _varint(footer, i)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)
    for _, handle in _block(data, ioff, isize):
        j = 0
        boff, j = _varint(handle, j)
        bsize, j = _varint(handle, j)
        for key, val in _block(data, boff, bsize):
            if key == b"":
                return int(_proto(val).get(1, [1])[0]) or 1
    return 1


def read_bundle(prefix, include_optimizer_slots=False):
    """{variable name: numpy array} (names stripped of '/.ATTRIBUTES/VARIABLE_VALUE').  Every entry is read from the
    data shard its BundleEntryProto names (`<prefix>.data-<shard_id>-of-<num_shards>`); a missing shard file raises
    FileNotFoundError naming it."""
    prefix = str(prefix)
    entries = read_index(prefix + ".index")
    num_shards = read_num_shards(prefix + ".index")
    out, files = {}, {}
    try:
        for key, dt, shape, shard, off, size in entries:
            if not key.endswith(_SUFFIX) or dt not in _DTYPES:
                continue
            if ".OPTIMIZER_SLOT" in key and not include_optimizer_slots:
                continue
            if not 0 <= shard < num_shards:
                raise ValueError(f"{prefix}.index: entry {key!r} names shard {shard} of {num_shards}")
            f = files.get(shard)
            if f is None:
                f = files[shard] = open(f"{prefix}.data-{shard:05d}-of-{num_shards:05d}", "rb")
            f.seek(off)
            arr = np.frombuffer(f.read(size), dtype=_DTYPES[dt]).reshape(shape).copy()   # writable, detached from the buffer
            out[key[:-len(_SUFFIX)]] = arr
    finally:
        for f in files.values():
            f.close()
    return out


# ---- writer -------------------------------------------------------------------------------------------------------------
_MAGIC = 0xdb4775248b80fb57
_RESTART_INTERVAL = 16
_NP2DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like object through the library's host routine (wdg_crc32c)."""
    import ctypes as C
    from . import native
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    return int(native.load().wdg_crc32c(C.c_char_p(bytes(buf)), len(buf), crc))


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _pv(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def encode_entry(dtype, shape, shard=0, offset=0, size=0, crc=0):
    """BundleEntryProto bytes: zero-valued scalar fields are omitted (proto3), the shape message always written."""
    dims = b"".join(b"\x12" + _pv(len(d)) + d for d in (b"\x08" + _pv(int(n)) for n in shape))
    out = b"\x08" + _pv(dtype) + b"\x12" + _pv(len(dims)) + dims
    if shard:
        out += b"\x18" + _pv(shard)
    if offset:
        out += b"\x20" + _pv(offset)
    if size:
        out += b"\x28" + _pv(size)
    return out + b"\x35" + struct.pack("<I", crc)


def _build_block(items):
    """LevelDB block: prefix-compressed entries, a restart point every 16 entries, restart array, count."""
    out, restarts, last = bytearray(), [], b""
    for n, (key, val) in enumerate(items):
        shared = 0
        if n % _RESTART_INTERVAL == 0:
            restarts.append(len(out))
        else:
            m = min(len(key), len(last))
            while shared < m and key[shared] == last[shared]:
                shared += 1
        out += _pv(shared) + _pv(len(key) - shared) + _pv(len(val)) + key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _short_successor(key):
    """LevelDB BytewiseComparator::FindShortSuccessor: first byte that can be incremented, truncated after it."""
    for i, c in enumerate(key):
        if c != 0xFF:
            return key[:i] + bytes([c + 1])
    return key


def encode_index(items):
    """items: [(key bytes, value bytes)] sorted by key, first key b"" = BundleHeaderProto -> bytes of `<prefix>.index`
    (one data block, as TensorFlow writes for bundles below its 256 KB table block size)."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", _mask(crc32c(block + b"\x00"))))
        return _pv(off) + _pv(len(block))

    data_handle = emit(_build_block(items))
    meta_handle = emit(_build_block([]))
    index_handle = emit(_build_block([(_short_successor(items[-1][0]), data_handle)]))
    footer = meta_handle + index_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))
    return bytes(out)


HEADER = b"\x08\x01\x1a\x02\x08\x01"    # BundleHeaderProto{num_shards: 1, version{producer: 1}}


def write_bundle(prefix, tensors):
    """{variable name: array} -> `<prefix>.index` + `<prefix>.data-00000-of-00001`, keys
    `<name>/.ATTRIBUTES/VARIABLE_VALUE` as Keras' TF-format `save_weights` names them (ganbase.py:132-135).
    Readable by `read_bundle` and by TensorFlow's checkpoint reader (`tf.train.load_checkpoint`); Keras'
    `load_weights` additionally wants the `_CHECKPOINTABLE_OBJECT_GRAPH` entry, which is not written."""
    prefix = str(prefix)
    items, offset = [(b"", HEADER)], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors, key=lambda k: (k + _SUFFIX).encode()):
            arr = np.ascontiguousarray(tensors[name])
            if arr.dtype not in _NP2DT:
                arr = arr.astype(np.float32)
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            items.append(((name + _SUFFIX).encode(), encode_entry(_NP2DT[arr.dtype], arr.shape, 0, offset, len(raw),
                                                                  _mask(crc32c(raw)))))
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        f.write(encode_index(items))


def read_raw_items(index_path):
    """[(key bytes, value bytes)] of every entry of an index file, in file order (tests re-encode these)."""
    data = open(index_path, "rb").read()
    footer, i = data[-48:], 0
    _, i = _varint(footer, i)
    _, i = _varint(footer, i)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)
    out = []
    for _, handle in _block(data, ioff, isize):
        j = 0
        boff, j = _varint(handle, j)
        bsize, j = _varint(handle, j)
        out += [(bytes(k), bytes(v)) for k, v in _block(data, boff, bsize)]
    return out
