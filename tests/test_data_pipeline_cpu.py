"""Host-side data feeding: the numpy restatement in downscaling/data/data_generator.py against golden
vectors produced by the REFERENCE's own numpy code (tests/golden/make_golden.py::reference_data_pipeline runs
/root/reference/src/downscaling/data/data_generator.py with TensorFlow / xarray stubbed), plus the crop / provider
logic that has no TensorFlow-free counterpart to record."""
import datetime
import json
from pathlib import Path

import numpy as np

from downscaling.data import data_generator as dg

GOLD = json.loads((Path(__file__).parent / "golden" / "data_pipeline.json").read_text())


def _eq(a, b):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-6, atol=1e-6,
                               equal_nan=True)


def test_transform_sequence_matches_reference():
    bg = dg._BatchGenerator.__new__(dg._BatchGenerator)
    for case in GOLD["transform_sequence"]:
        bg.prng = np.random.RandomState(case["seed"])
        X = np.arange(2 * 4 * 4 * 2, dtype=np.float64).reshape(2, 4, 4, 2)
        Y = -np.arange(2 * 4 * 4 * 1, dtype=np.float64).reshape(2, 4, 4, 1)
        x2, y2 = bg.transform_sequence(X, Y)
        _eq(x2, case["X"])
        _eq(y2, case["Y"])
        _eq(bg.transform_sequence(X), case["X_only"])     # same RandomState keeps drawing


def test_survey_known_answer_for_transform_sequence():
    """SURVEY §8 c: RandomState(123) on arange(32).reshape(2,4,4,1) -> first frame rows [12..15],[8..11],[4..7],[0..3]."""
    bg = dg._BatchGenerator.__new__(dg._BatchGenerator)
    bg.prng = np.random.RandomState(123)
    out = bg.transform_sequence(np.arange(32).reshape(2, 4, 4, 1))
    assert out[0, :, :, 0].tolist() == [[12, 13, 14, 15], [8, 9, 10, 11], [4, 5, 6, 7], [0, 1, 2, 3]]


def test_naive_decoder_matches_reference():
    g = GOLD["naive"]
    img = np.asarray(g["input"], dtype=np.float64)
    nd = dg.NaiveDecoder()
    _eq(nd.normalize(img), g["normalize"])
    _eq(nd.normalize_positive(img), g["normalize_positive"])
    _eq(nd.denormalize(img), g["denormalize"])
    _eq(nd.denormalize_positive(img), g["denormalize_positive"])
    _eq(nd(img), g["call"])
    _eq(dg.NaiveDecoder(False)(img), g["call_off"])


def test_wind_decoders_match_reference():
    g = GOLD["wind_speed"]
    w = np.asarray(g["input"], dtype=np.float64)
    ws, wsn = dg.WindSpeedDecoder(below_val=-2.0), dg.WindSpeedDecoder(below_val=-2.0, normalize=True)
    _eq(ws(w), g["call"])
    _eq(wsn(w), g["call_normalized"])
    _eq(ws.denormalize(wsn(w).copy()), g["denormalize"])
    _eq(dg.WindSpeedDecoder()(w), g["default_nan"])
    g = GOLD["wind_component"]
    c = np.asarray(g["input"], dtype=np.float64)
    wc, wcr = dg.WindComponentDecoder(below_val=-10.0), dg.WindComponentDecoder(below_val=-10.0, normalize=False)
    _eq(wc(c), g["call"])
    _eq(wcr(c), g["call_raw"])
    _eq(wc.denormalize(wcr(c).copy(), set_nan=False), g["denormalize"])


def _days(n=3, nt=24, nx=40, ny=50):
    rng = np.random.default_rng(0)
    base = datetime.datetime(2020, 3, 1)
    return {base + datetime.timedelta(days=2 * i): {"u10": rng.standard_normal((nt, nx, ny)), "v10": rng.standard_normal((nt, nx, ny)),
                                                    "elevation": np.full((nt, nx, ny), 1500.0),
                                                    "U_10M": rng.standard_normal((nt, nx, ny)), "V_10M": rng.standard_normal((nt, nx, ny))}
            for i in range(n)}


def test_batch_generator_crops_and_pairs():
    days = _days()
    bg = dg._BatchGenerator(dg.ArrayProvider(days), dg.NaiveDecoder(False), dg.ArrayProvider(days), sequence_length=6,
                            patch_length_pixel=16, batch_size=5, transform=False,
                            input_variables=("u10", "v10", "elevation"), output_variables=("U_10M", "V_10M"))
    assert bg.dates == sorted(days)
    np.random.seed(7)
    X, Y = next(bg)                                       # first date
    assert X.shape == (5, 6, 16, 16, 3) and Y.shape == (5, 6, 16, 16, 2)
    assert np.all(X[..., 2] == 1.5)                       # elevation served in km (data_generator.py:222-223)
    # replay the reference's draw order (x, y, time per sample) on the same global RNG
    np.random.seed(7)
    day = days[bg.dates[0]]
    for b in range(5):
        rx, ry, rt = np.random.randint(0, 40 + 1 - 16), np.random.randint(0, 50 + 1 - 16), np.random.randint(0, 24 + 1 - 6)
        np.testing.assert_array_equal(X[b, ..., 0], day["u10"][rt:rt + 6, rx:rx + 16, ry:ry + 16])
        np.testing.assert_array_equal(Y[b, ..., 1], day["V_10M"][rt:rt + 6, rx:rx + 16, ry:ry + 16])
    # the same transform hits input and output
    bg2 = dg._BatchGenerator(dg.ArrayProvider(days), dg.NaiveDecoder(False), dg.ArrayProvider(days), sequence_length=6,
                             patch_length_pixel=16, batch_size=4, transform=True, input_variables=("U_10M",),
                             output_variables=("U_10M",))
    bg2.reset(random_seed=3)
    X, Y = bg2()
    np.testing.assert_array_equal(X, Y)
    # date filters, cycling, Sequence face
    bg3 = dg._BatchGenerator(dg.ArrayProvider(days), dg.NaiveDecoder(), start_date="20200302", end_date="2020-03-05",
                             patch_length_pixel=16, batch_size=2, input_variables=("u10",))
    assert len(bg3) == 2 and bg3.next_date() == bg3.dates[0] and bg3.next_date() == bg3.dates[1] and bg3.next_date() == bg3.dates[0]
    seq = dg.BatchGenerator(dg.ArrayProvider(days), dg.NaiveDecoder(), patch_length_pixel=16, batch_size=2, input_variables=("u10",))
    assert len(seq) == 5 and seq[1].shape == (2, 6, 16, 16, 1)    # 1 March .. 5 March inclusive
    with seq as inner:
        assert next(inner).shape == (2, 6, 16, 16, 1)


def test_local_file_provider(tmp_path):
    days = _days(2)
    for d, arrs in days.items():
        np.savez(tmp_path / f"x_{d:%Y%m%d}.npz", **arrs)
    (tmp_path / "notes.txt").write_text("ignored")
    prov = dg.LocalFileProvider(tmp_path)
    assert prov.available_dates == sorted(days)
    got = prov.provide(sorted(days)[1])
    np.testing.assert_array_equal(got["v10"], days[sorted(days)[1]]["v10"])


def test_structured_noise_layout():
    """NoiseGenerator (data_generator.py:296-316): reshape(repeat(flat draw, n), (bs,t,x,y)) — only the time-varying
    channel is constant along the axes its name suggests."""
    bs, t, x, y = 2, 3, 4, 5
    d = np.arange(bs * t, dtype=np.float64).reshape(bs, t)
    tv = dg.NoiseGenerator.layout(d, x * y, (bs, t, x, y))
    assert np.all(tv == d[:, :, None, None])
    d = np.arange(bs * x, dtype=np.float64).reshape(bs, x)
    lon = dg.NoiseGenerator.layout(d, t * y, (bs, t, x, y))
    flat = lon.reshape(bs, -1)
    for b in range(bs):
        assert flat[b].tolist() == np.repeat(d[b], t * y).tolist()
    assert not np.all(lon == d[:, None, :, None])          # the documented scrambling
