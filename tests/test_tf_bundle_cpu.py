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
    off = 0
    for key in sorted(ents, key=lambda s: s.encode()):
        _, dt, shape, shard, o, size = ents[key]
        assert (shard, o) == (0, off)
        off += size
    raw = (tmp_path / "generator.data-00000-of-00001").read_bytes()
    assert len(raw) == off
    for key, val in tb.read_raw_items(str(tmp_path / "generator") + ".index")[1:]:
        p = tb._proto(val)
        o, size = p.get(4, [0])[0], p[5][0]
        assert p[6][0] == tb._mask(tb.crc32c(raw[o:o + size]))
