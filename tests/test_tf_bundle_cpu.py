"""TF tensor-bundle checkpoint format (ganbase.py:132-140, api.py:21,85): the writer re-encodes the
reference's SHIPPED index files byte for byte (tests/golden/weights-55_*.index, copied by make_golden.py), CRC-32C
matches its published check value, and write -> read round-trips."""
from pathlib import Path

import numpy as np

from downscaling.engine import tf_bundle as tb

GOLD = Path(__file__).parent / "golden"


def test_crc32c_check_value():
    assert tb.crc32c(b"123456789") == 0xE3069283          # the CRC-32C (Castagnoli) check value
    assert tb.crc32c(b"6789", tb.crc32c(b"12345")) == 0xE3069283
    assert tb.crc32c(b"") == 0


def test_writer_reencodes_shipped_index_byte_for_byte():
    for name in ("generator", "discriminator"):
        path = GOLD / f"weights-55_{name}.index"
        items = tb.read_raw_items(path)
        assert items[0] == (b"", tb.HEADER)
        assert tb.encode_index(items) == path.read_bytes()
        for key, val in items[1:]:                           # every BundleEntryProto re-encodes identically
            p = tb._proto(val)
            shape = [tb._proto(d).get(1, [0])[0] for d in tb._proto(p[2][0]).get(2, [])] if 2 in p else []
            assert tb.encode_entry(p[1][0], shape, p.get(3, [0])[0], p.get(4, [0])[0], p.get(5, [0])[0], p.get(6, [0])[0]) == val, key


def test_write_read_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"layer_with_weights-0/layer/w": rng.standard_normal((8, 8, 23, 128)).astype(np.float32),
               "layer_with_weights-0/layer/layer/bias": rng.standard_normal(128).astype(np.float32),
               "layer_with_weights-10/moving_mean": np.zeros(16, dtype=np.float32),
               "layer_with_weights-2/layer/sn_u": rng.standard_normal((1, 128)).astype(np.float32),
               "step": np.asarray(7, dtype=np.int64)}
    tb.write_bundle(tmp_path / "generator", tensors)
    back = tb.read_bundle(tmp_path / "generator")
    assert set(back) == set(tensors)
    for k in tensors:
        np.testing.assert_array_equal(back[k], tensors[k])
    ents = {e[0]: e for e in tb.read_index(str(tmp_path / "generator") + ".index")}
    # entries are laid out in key order, offsets contiguous, masked CRC-32C of each tensor's bytes
    # (the object-graph string tensor: first key, LAST in the data file — where the shipped checkpoints have it too)
    off = 0
    for key in sorted((k for k in ents if k != tb.OBJECT_GRAPH_KEY), key=lambda s: s.encode()) + [tb.OBJECT_GRAPH_KEY]:
        _, dt, shape, shard, o, size = ents[key]
        assert (shard, o) == (0, off)
        off += size
    assert ents[tb.OBJECT_GRAPH_KEY][1:3] == (7, ()) and list(ents)[0] == tb.OBJECT_GRAPH_KEY
    raw = (tmp_path / "generator.data-00000-of-00001").read_bytes()
    assert len(raw) == off
    for key, val in tb.read_raw_items(str(tmp_path / "generator") + ".index")[1:]:
        p = tb._proto(val)
        o, size = p.get(4, [0])[0], p[5][0]
        assert p[6][0] == tb._mask(tb.crc32c(raw[o:o + size]))


def test_object_graph_entry_matches_shipped_layout():
    """The shipped checkpoints keep `_CHECKPOINTABLE_OBJECT_GRAPH` as a DT_STRING scalar, first key of the index and last
    tensor of the data file; the blob itself is not in the reference tree (.MISSING_LARGE_BLOBS), only its size and CRC."""
    for name in ("generator", "discriminator"):
        ents = tb.read_index(GOLD / f"weights-55_{name}.index")
        assert ents[0][0] == tb.OBJECT_GRAPH_KEY and ents[0][1:3] == (7, ())
        assert ents[0][4] == max(e[4] for e in ents)


def _walk(nodes, path):
    node = 0
    for part in path:
        node = dict(nodes[node][0])[part]
    return node


def test_object_graph_reaches_every_variable_of_the_shipped_checkpoints(tmp_path):
    """Keras' object-based restore follows children.local_name from node 0 and reads the tensor its leaf's checkpoint_key
    names: for the variable sets of the SHIPPED generator / discriminator checkpoints (model variables; optimizer slots
    are not part of this package's checkpoints) every key must be reachable along its own path, the spectral-norm wrapper's
    kernel additionally as `layer/layer/kernel`, breadth-first numbering, one attribute per variable node."""
    for name in ("generator", "discriminator"):
        keys = [e[0][:-len(tb._SUFFIX)] for e in tb.read_index(GOLD / f"weights-55_{name}.index")
                if e[0].endswith(tb._SUFFIX) and ".OPTIMIZER_SLOT" not in e[0] and not e[0].startswith("optimizer/")]
        assert len(keys) == {"generator": 39, "discriminator": 38}[name], len(keys)
        tensors = {k: np.zeros(3, dtype=np.float32) for k in keys}
        tb.write_bundle(tmp_path / name, tensors)
        nodes = tb.read_object_graph(tmp_path / name)
        seen = set()
        for k in keys:
            leaf = _walk(nodes, k.split("/"))
            (attr,) = nodes[leaf][1]
            assert attr[0] == "VARIABLE_VALUE" and attr[2] == k + tb._SUFFIX
            seen.add(leaf)
            if k.endswith("/layer/w"):                 # SpectralNormalization: w IS the wrapped layer's kernel
                assert _walk(nodes, k.split("/")[:-1] + ["layer", "kernel"]) == leaf
        assert len(seen) == len(keys)
        # breadth-first: a child's id is larger than its parent's unless it was reached earlier through another parent
        for i, (ch, _) in enumerate(nodes):
            assert all(c > i or any(c in [x for _, x in nodes[j][0]] for j in range(i)) for _, c in ch)
        # variables are leaves, inner nodes carry no tensors
        assert all((not ch) == bool(at) for ch, at in nodes)
        assert set(tb.read_bundle(tmp_path / name)) == set(keys)


def test_string_tensor_encoding_round_trip():
    blob = bytes(range(256)) * 41                                    # 10,496 bytes: two-byte varint length like the shipped graphs
    raw, crc = tb.encode_string_tensor([blob])
    assert len(raw) == 2 + 4 + len(blob) and tb.decode_string_tensor(raw) == [blob]
    assert crc == tb.crc32c(raw)                                     # the entry checksum runs over everything written
    bad = bytearray(raw)
    bad[0] ^= 1
    try:
        tb.decode_string_tensor(bytes(bad))
        assert False
    except ValueError:
        pass
