"""Generates the golden fixtures of tests/golden/ from the reference tree (run in the build container
only: /root/reference does not exist on the GPU box).

  tile_plan.json        the integer tile planner of predict(), AST-extracted from
                        /root/reference/src/downscaling/api.py:99-116 and executed as written
                        (np.math shimmed for numpy 2) — TensorFlow/xarray are not needed for it.
  generator_index.json  variable names / shapes decoded from
                        /root/reference/src/downscaling/weights-55.ckpt/generator.index
  discriminator_index.json  same for the discriminator (shortcut variant, SURVEY §8 a2 note 2)
  data_pipeline.json    the numpy-only members of /root/reference/src/downscaling/data/data_generator.py
                        (`_BatchGenerator.transform_sequence` :271-290, `NaiveDecoder` :338-360,
                        `WindSpeedDecoder` / `WindComponentDecoder` :363-417) imported with TensorFlow, xarray,
                        parse stubbed by MagicMock and run on seeded inputs
  metrics.json          the numpy variants of /root/reference/src/downscaling/gan/metrics.py
                        (`tanh_wind_speed_weighted_rmse_from_xarray` :48-60, `cosine_similarity_from_xarray` :108-113,
                        `log_spectral_distance_from_xarray` :143-152, `rmse_from_xarray` :181-187) imported with
                        TensorFlow / tfa / tfp stubbed and run on seeded wind fields; they share their formulas with
                        the TF metrics (:32-45, 94-101, 121-137) the package implements
"""
import ast
import json
import math
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/src/downscaling")
sys.path.insert(0, str(HERE.parent.parent / "wind-downscaling-gan_amd"))


def reference_planner():
    src = (REF / "api.py").read_text()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "predict")
    # statements from `ntimeseq = ...` to `slices_start_y = ...` (api.py:99-116)
    names = [getattr(s.targets[0], "id", None) if isinstance(s, ast.Assign) else None for s in fn.body]
    first = next(i for i, s in enumerate(fn.body) if isinstance(s, ast.Assign) and getattr(s.targets[0], "id", "") == "ntimeseq")
    last = max(i for i, s in enumerate(fn.body) if isinstance(s, ast.Assign) and
               any(getattr(t, "id", "") == "slices_start_y" for t in s.targets))
    body = fn.body[first:last + 1]
    mod = ast.Module(body=body, type_ignores=[])
    code = compile(mod, "api.py[99:116]", "exec")

    class _NP:
        math = math

        def __getattr__(self, k):
            return getattr(np, k)

    def plan(pixels_lat, pixels_lon, time_window, overlap_factor):
        env = dict(np=_NP(), SEQUENCE_LENGTH=24, IMG_SIZE=96, pixels_lat=pixels_lat, pixels_lon=pixels_lon,
                   time_window=time_window, overlap_factor=overlap_factor)
        exec(code, env)
        return {k: (int(env[k]) if not isinstance(env[k], list) else [int(v) for v in env[k]])
                for k in ("ntimeseq", "ncols", "nrows", "xdist", "ydist", "leftovers_x", "leftovers_y",
                          "slices_start_x", "slices_start_y")}
    return plan


def main():
    plan = reference_planner()
    cases = []
    for (la, lo, tw, of) in [(1200, 1200, 24, 0.05), (1200, 1200, 24, 0.01), (128, 128, 24, 0.05), (260, 300, 48, 0.05),
                             (300, 500, 72, 0.3), (2000, 1500, 25, 0.1), (400, 400, 24, 1.0), (192, 192, 24, 0.0)]:
        cases.append(dict(pixels_lat=la, pixels_lon=lo, time_window=tw, overlap_factor=of, expected=plan(la, lo, tw, of)))
    (HERE / "tile_plan.json").write_text(json.dumps(cases, indent=1))
    from downscaling.engine.tf_bundle import read_index
    for name in ("generator", "discriminator"):
        ents = read_index(REF / "weights-55.ckpt" / f"{name}.index")
        suffix = "/.ATTRIBUTES/VARIABLE_VALUE"
        var = {k[:-len(suffix)]: list(shape) for (k, dt, shape, shard, off, size) in ents
               if k.endswith(suffix) and k.startswith("layer_with_weights") and ".OPTIMIZER_SLOT" not in k}
        nbytes = sum(size for (k, dt, shape, shard, off, size) in ents
                     if k.endswith(suffix) and k.startswith("layer_with_weights") and ".OPTIMIZER_SLOT" not in k)
        (HERE / f"{name}_index.json").write_text(json.dumps(dict(variables=var, variable_bytes=nbytes), indent=1))
        # the shipped index file itself (5 KB of data): the bundle WRITER is checked by re-encoding it byte for byte
        (HERE / f"weights-55_{name}.index").write_bytes((REF / "weights-55.ckpt" / f"{name}.index").read_bytes())


def reference_data_pipeline():
    """Import the reference's data_generator with its heavy imports stubbed and record what its numpy code does."""
    import importlib.util
    from unittest.mock import MagicMock
    for m in ("tensorflow", "xarray", "parse", "silence_tensorflow", "tensorflow.keras", "tensorflow.keras.utils"):
        sys.modules.setdefault(m, MagicMock())
    sys.modules["tensorflow"].keras.utils.Sequence = object
    spec = importlib.util.spec_from_file_location("ref_data_generator", REF / "data" / "data_generator.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {"transform_sequence": [], "naive": {}, "wind_speed": {}, "wind_component": {}}
    bg = mod._BatchGenerator.__new__(mod._BatchGenerator)
    for seed in (0, 1, 2, 3, 5, 7, 11, 123):
        bg.prng = np.random.RandomState(seed)
        X = np.arange(2 * 4 * 4 * 2, dtype=np.float64).reshape(2, 4, 4, 2)
        Y = -np.arange(2 * 4 * 4 * 1, dtype=np.float64).reshape(2, 4, 4, 1)
        x2, y2 = bg.transform_sequence(X, Y)
        x3 = bg.transform_sequence(X)       # the same RandomState keeps drawing
        out["transform_sequence"].append(dict(seed=seed, X=np.asarray(x2).tolist(), Y=np.asarray(y2).tolist(),
                                              X_only=np.asarray(x3).tolist()))
    rng = np.random.default_rng(42)
    img = rng.standard_normal((3, 4, 4, 2)) * 3 + 1
    img[1, 2, 3, 0] = np.nan
    nd = mod.NaiveDecoder()
    out["naive"] = dict(input=img.tolist(), normalize=nd.normalize(img).tolist(),
                        normalize_positive=nd.normalize_positive(img).tolist(), denormalize=nd.denormalize(img).tolist(),
                        denormalize_positive=nd.denormalize_positive(img).tolist(), call=nd(img).tolist(),
                        call_off=mod.NaiveDecoder(False)(img).tolist())
    w = rng.standard_normal((2, 3, 3, 1)) * 2
    w[0, 0, 0, 0] = 0.0
    w[1, 1, 1, 0] = 5.0
    w[1, 2, 2, 0] = -3.0
    ws, wsn = mod.WindSpeedDecoder(below_val=-2.0), mod.WindSpeedDecoder(below_val=-2.0, normalize=True)
    dec = ws(w)
    out["wind_speed"] = dict(input=w.tolist(), call=dec.tolist(), call_normalized=wsn(w).tolist(),
                             denormalize=ws.denormalize(wsn(w).copy()).tolist(), default_nan=mod.WindSpeedDecoder()(w).tolist())
    c = rng.standard_normal((2, 3, 3, 2)) * 6
    c[0, 1, 1, 1] = 0.0
    wc, wcr = mod.WindComponentDecoder(below_val=-10.0), mod.WindComponentDecoder(below_val=-10.0, normalize=False)
    out["wind_component"] = dict(input=c.tolist(), call=wc(c).tolist(), call_raw=wcr(c).tolist(),
                                 denormalize=wc.denormalize(wcr(c).copy(), set_nan=False).tolist())
    (HERE / "data_pipeline.json").write_text(json.dumps(out))


def reference_metrics():
    """Import the reference's gan/metrics.py with TensorFlow & co. stubbed and run its numpy-only members."""
    import importlib.util
    import types
    from unittest.mock import MagicMock
    for m in ("tensorflow", "tensorflow_addons", "tensorflow_probability", "tensorflow.keras", "tensorflow.keras.losses"):
        sys.modules[m] = MagicMock()
    sys.modules["tensorflow"].keras.backend.epsilon = lambda: 1e-7
    sys.modules["xarray"] = types.SimpleNamespace(where=np.where)
    spec = importlib.util.spec_from_file_location("ref_metrics", REF / "gan" / "metrics.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class Fields:                      # the slice of an xarray Dataset those functions touch
        def __init__(self, **arrays):
            self.__dict__.update(arrays)

        def __getitem__(self, names):
            arrays = [self.__dict__[n] for n in names]
            return types.SimpleNamespace(to_array=lambda: np.stack(arrays))
    rng = np.random.default_rng(2024)
    T, H, W = 3, 12, 10
    real = rng.standard_normal((T, H, W, 2)) * 4
    fake = real + rng.standard_normal((T, H, W, 2)) * 1.5
    r = Fields(U_10M=real[..., 0], V_10M=real[..., 1])
    f = Fields(u10=fake[..., 0], v10=fake[..., 1])
    out = dict(real=real.tolist(), fake=fake.tolist(),
               tanh_ws_weighted_rmse=np.asarray(mod.tanh_wind_speed_weighted_rmse_from_xarray(r, f)).tolist(),
               cosine_similarity=np.asarray(mod.cosine_similarity_from_xarray(r, f)).tolist(),
               rmse=np.asarray(mod.rmse_from_xarray(real[None], fake[None])).tolist())
    out["log_spectral_distance"] = np.asarray(mod.log_spectral_distance_from_xarray(r, f)).tolist()
    (HERE / "metrics.json").write_text(json.dumps(out))


if __name__ == "__main__":
    main()
    reference_data_pipeline()
    reference_metrics()
