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


OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
_DT_STRING = 7


def _pstr(field, b):
    return bytes([(field << 3) | 2]) + _pv(len(b)) + b


def build_object_graph(names):
    """Serialized `TrackableObjectGraph` (tensorflow/core/protobuf/trackable_object_graph.proto) for the variables `names`
    (checkpoint names without the attribute suffix, e.g. `layer_with_weights-0/layer/w`): the entry Keras' TF-format
    `load_weights` (ganbase.py:137-140) walks.  Its restore binds node 0 to the model and follows `children.local_name`
    through the live object's tracked dependencies, so what must be right is the TREE of local names and every leaf's
    `attributes{name: "VARIABLE_VALUE", checkpoint_key}`; both follow from the key paths themselves (a key IS the path of
    local names from the root).  Nodes are numbered breadth-first from the root, children in first-seen key order, as
    TensorFlow's graph view does.  A spectral-normalisation wrapper (children `layer`, `w`, `sn_u`: tf_utils.py:19-31)
    additionally exposes its kernel as `layer/kernel` — the same variable node as `w`, which is how the wrapped Conv2D
    reaches it.  Not reproduced (unknowable here: the blob holding the shipped graph is absent, only its size and CRC are in
    the index): `full_name` strings with Keras' per-process layer-name counters, the `layer-<i>` aliases, and the
    optimizer's slot-variable table — none of them take part in restoring model weights."""
    children, order = {(): []}, [()]
    for name in names:
        parts = tuple(name.split("/"))
        for i in range(len(parts)):
            parent, node = parts[:i], parts[:i + 1]
            if node not in children:
                children[node] = []
                children[parent].append((parts[i], node))
    alias = {}
    for node, ch in list(children.items()):
        local = dict(ch)
        if {"layer", "w", "sn_u"} <= set(local):
            conv = local["layer"]
            if not any(n == "kernel" for n, _ in children[conv]):
                children[conv].insert(0, ("kernel", local["w"]))
                alias[(conv, "kernel")] = local["w"]
    ids, queue = {(): 0}, [()]
    while queue:                                      # breadth-first numbering (aliases keep the id of their first visit)
        node = queue.pop(0)
        for _, child in children[node]:
            if child not in ids:
                ids[child] = len(ids)
                order.append(child)
                queue.append(child)
    leaves = {tuple(n.split("/")) for n in names}
    out = bytearray()
    for node in sorted(ids, key=ids.get):
        body = bytearray()
        for local_name, child in children[node]:
            ref = (b"\x08" + _pv(ids[child]) if ids[child] else b"") + _pstr(2, local_name.encode())
            body += _pstr(1, ref)
        if node in leaves:
            key = "/".join(node)
            attr = _pstr(1, b"VARIABLE_VALUE") + _pstr(2, "/".join(node[-2:]).encode()) + _pstr(3, (key + _SUFFIX).encode())
            body += _pstr(2, attr)
        out += _pstr(1, bytes(body))
    return bytes(out)


def parse_object_graph(blob):
    """-> [(children [(local_name, node_id)], attributes [(name, full_name, checkpoint_key)])] per node."""
    nodes = []
    for nb in _proto(blob).get(1, []):
        f = _proto(nb)
        ch = []
        for c in f.get(1, []):
            cf = _proto(c)
            ch.append((bytes(cf.get(2, [b""])[0]).decode(), cf.get(1, [0])[0]))
        at = []
        for a in f.get(2, []):
            af = _proto(a)
            at.append(tuple(bytes(af.get(i, [b""])[0]).decode() for i in (1, 2, 3)))
        nodes.append((ch, at))
    return nodes


def encode_string_tensor(strings):
    """Bytes of a DT_STRING tensor inside a bundle data file (tensor_bundle.cc, WriteStringTensor): the varint64 lengths,
    the masked CRC-32C of the length bytes (4 bytes, little endian), then the strings; returns (bytes, unmasked CRC-32C the
    BundleEntryProto records masked) — the entry checksum runs over lengths, length checksum and string bytes."""
    lens = b"".join(_pv(len(x)) for x in strings)
    cks = struct.pack("<I", _mask(crc32c(lens)))
    body = b"".join(strings)
    crc = crc32c(body, crc32c(cks, crc32c(lens)))
    return lens + cks + body, crc


def decode_string_tensor(raw, count=1):
    i, lens = 0, []
    for _ in range(count):
        n, i = _varint(raw, i)
        lens.append(n)
    if struct.unpack("<I", raw[i:i + 4])[0] != _mask(crc32c(raw[:i])):
        raise ValueError("string tensor: length checksum mismatch")
    i += 4
    out = []
    for n in lens:
        out.append(bytes(raw[i:i + n]))
        i += n
    return out


def read_object_graph(prefix):
    """The parsed `_CHECKPOINTABLE_OBJECT_GRAPH` of a bundle (None when the index has no such entry)."""
    prefix = str(prefix)
    for key, dt, shape, shard, off, size in read_index(prefix + ".index"):
        if key == OBJECT_GRAPH_KEY and dt == _DT_STRING:
            num_shards = read_num_shards(prefix + ".index")
            with open(f"{prefix}.data-{shard:05d}-of-{num_shards:05d}", "rb") as f:
                f.seek(off)
                return parse_object_graph(decode_string_tensor(f.read(size))[0])
    return None


def write_bundle(prefix, tensors, object_graph=True):
    """{variable name: array} -> `<prefix>.index` + `<prefix>.data-00000-of-00001`, keys
    `<name>/.ATTRIBUTES/VARIABLE_VALUE` as Keras' TF-format `save_weights` names them (ganbase.py:132-135), plus the
    `_CHECKPOINTABLE_OBJECT_GRAPH` string tensor (last in the data file, first in key order, as TensorFlow lays it out) that
    Keras' object-based `load_weights` walks (build_object_graph).  Readable by `read_bundle` and, by construction of the
    SSTable / CRC layout (tensor_bundle.cc), by TensorFlow's checkpoint reader (`tf.train.load_checkpoint`).  The object
    graph is SYNTHESISED from the key paths — structurally valid, but untested against Keras (no TensorFlow in this image):
    `keras.Model.load_weights` may still want the name-based route (`by_name` / `tf.train.load_checkpoint` + assign)."""
    prefix = str(prefix)
    items, offset = [(b"", HEADER)], 0
    order = sorted(tensors, key=lambda k: (k + _SUFFIX).encode())
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in order:
            arr = np.ascontiguousarray(tensors[name])
            if arr.dtype not in _NP2DT:
                arr = arr.astype(np.float32)
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            items.append(((name + _SUFFIX).encode(), encode_entry(_NP2DT[arr.dtype], arr.shape, 0, offset, len(raw),
                                                                  _mask(crc32c(raw)))))
            offset += len(raw)
        if object_graph:
            raw, crc = encode_string_tensor([build_object_graph(order)])
            f.write(raw)
            items.append((OBJECT_GRAPH_KEY.encode(), encode_entry(_DT_STRING, (), 0, offset, len(raw), _mask(crc))))
            items.sort(key=lambda kv: kv[0])
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
