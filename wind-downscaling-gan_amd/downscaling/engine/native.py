"""ctypes binding of libwdgan.so (the C ABI declared in include/wdgan.h).

The library is the product path: if it is missing or cannot be loaded this module raises — there
is no CPU fallback anywhere in the package.
"""
import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent.parent / "libwdgan.so"

c_fp = C.c_void_p  # device pointers travel as integers
i32, i64, u64, f32, f64, szt = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t


class ConvGeom(C.Structure):
    _fields_ = [
        ("n_img", i32), ("H", i32), ("W", i32), ("Cin", i32), ("ldx", i32), ("img_stride_x", i64),
        ("Ho", i32), ("Wo", i32), ("Cout", i32), ("ldy", i32), ("img_stride_y", i64),
        ("kh", i32), ("kw", i32), ("stride", i32), ("pad_h", i32), ("pad_w", i32),
    ]


class PrepLayer(C.Structure):
    _fields_ = [
        ("w", c_fp), ("u", c_fp), ("wF", c_fp), ("wD", c_fp),
        ("rows", i32), ("cols", i32), ("taps", i32), ("cin", i32), ("cout", i32), ("sn", i32),
    ]


PREP_SN, PREP_PACK_ALL = 1, 2

# name -> (restype, argtypes); every symbol of include/wdgan.h
SIGNATURES = {
    "wdg_last_error": (C.c_char_p, []),
    "wdg_version": (C.c_char_p, []),
    "wdg_device_cus": (i32, []),
    "wdg_crc32c": (C.c_uint32, [C.c_void_p, szt, C.c_uint32]),
    "wdg_set_tuning": (i32, [C.c_char_p, i32]),
    "wdg_tuning_epoch": (i32, []),
    "wdg_convlstm_step_supported": (i32, [c_fp, i32]),
    "wdg_convlstm_step": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i32, c_fp]),
    "wdg_convlstm_step_gemm_supported": (i32, [C.c_void_p, i32]),
    "wdg_convlstm_step_gemm": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i32, c_fp]),
    "wdg_convlstm16_supported": (i32, [c_fp]),
    "wdg_convlstm16_pack": (i32, [c_fp, c_fp, c_fp, c_fp]),
    "wdg_convlstm16_step": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, c_fp]),
    "wdg_convlstm16_bwd_step": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp]),
    "wdg_convlstm_bwd_step_supported": (i32, [c_fp, i32]),
    "wdg_convlstm_bwd_step": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, c_fp]),
    "wdg_convlstm_h16_supported": (i32, [c_fp, i32]),
    "wdg_conv_fwd_h16_gates": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, c_fp]),
    "wdg_convlstm_step_h16": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i32, i32, c_fp]),
    "wdg_conv_fwd_h16_gates_x16": (i32, [c_fp, c_fp, i32, c_fp, c_fp, c_fp, i32, i32, c_fp]),
    "wdg_convlstm_step_h16x": (i32, [c_fp, c_fp, i32, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, c_fp]),
    "wdg_tiles_gather_normalise": (i32, [c_fp, i32, i32, i32, c_fp, i32, i32, i32, c_fp, c_fp, i32, c_fp, c_fp]),
    "wdg_tiles_blend": (i32, [c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    "wdg_conv_plan_create": (i32, [C.POINTER(C.c_void_p), C.POINTER(ConvGeom)]),
    "wdg_conv_plan_create_sliced": (i32, [C.POINTER(C.c_void_p), C.POINTER(ConvGeom), i32]),
    "wdg_conv_plan_destroy": (i32, [C.c_void_p]),
    "wdg_conv_ws_bytes": (szt, [C.c_void_p]),
    "wdg_conv_plan_info": (i32, [C.c_void_p, C.POINTER(i32)]),
    "wdg_conv_fwd": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, i32, f32, i32, c_fp, szt, c_fp]),
    "wdg_conv_dgrad": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, i32, f32, i32, c_fp, szt, c_fp]),
    "wdg_conv_fwd_bn": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp, i32, c_fp, c_fp, szt, c_fp]),
    "wdg_conv_dgrad_bn": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp, i32, c_fp, c_fp, szt, c_fp]),
    "wdg_conv_dgrad_lnbwd": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, i32, i64, c_fp, c_fp, i32, i32, f32, c_fp, c_fp, c_fp, c_fp, c_fp, szt, c_fp]),
    "wdg_conv_dgrad_lnbwd_par_floats": (i64, [i32]),
    "wdg_conv_dgrad_lnbwd_route": (i32, [C.c_void_p, i32, i32, i32, i32, szt]),
    "wdg_conv_fwd_ln": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, f32, c_fp, i32, f32, c_fp, szt, c_fp]),
    "wdg_conv_fwd_ln_strided": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i64, c_fp, c_fp, f32, c_fp, i32, f32, c_fp, szt, c_fp]),
    "wdg_upconv_fwd": (i32, [C.c_void_p, c_fp, i32, i64, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    "wdg_upconv4_weight_floats": (szt, [i32, i32]),
    "wdg_upconv4_supported": (i32, [i32, i32, i32, i32]),
    "wdg_upconv4_pack": (i32, [c_fp, i32, i32, c_fp, c_fp]),
    "wdg_upconv4_fwd": (i32, [c_fp, i32, i64, i32, i32, i32, i32, c_fp, c_fp, c_fp, i32, i64, i32, i32, f32, c_fp]),
    "wdg_convert_bf16": (i32, [c_fp, c_fp, i64, c_fp]),
    "wdg_conv_fwd_bf16": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, i32, c_fp]),
    "wdg_conv_dgrad_bf16": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, i32, c_fp]),
    "wdg_conv_halo_fwd_bf16": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    "wdg_upconv_fwd_bf16": (i32, [C.c_void_p, c_fp, i32, i64, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    "wdg_convert_f16": (i32, [c_fp, c_fp, i64, c_fp]),
    "wdg_conv_fwd_f16": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, i32, c_fp]),
    "wdg_conv_dgrad_f16": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, i32, c_fp]),
    "wdg_conv_halo_fwd_f16": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    "wdg_upconv_fwd_f16": (i32, [C.c_void_p, c_fp, i32, i64, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    "wdg_conv_wgrad": (i32, [C.c_void_p, c_fp, c_fp, c_fp, i32, c_fp, szt, c_fp]),
    "wdg_conv_wgrad_bias": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, i32, c_fp, szt, c_fp]),
    "wdg_weight_pack": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "wdg_sn_scratch_floats": (szt, [i32, i32]),
    "wdg_sn_power_iter": (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp]),
    "wdg_prep_batch_create": (i32, [C.POINTER(C.c_void_p), C.POINTER(PrepLayer), i32]),
    "wdg_prep_batch_scratch_floats": (szt, [C.c_void_p]),
    "wdg_prep_batch_run": (i32, [C.c_void_p, c_fp, i32, c_fp]),
    "wdg_prep_batch_destroy": (i32, [C.c_void_p]),
    "wdg_bn_stats": (i32, [c_fp, i64, i32, i32, c_fp, c_fp]),
    "wdg_bn_finalize_train": (i32, [c_fp, i32, f64, c_fp, c_fp, c_fp, c_fp, f32, f32, c_fp, c_fp, i32, c_fp]),
    "wdg_bn_collapse": (i32, [c_fp, i32, i32, c_fp]),
    "wdg_bn_finalize_infer": (i32, [c_fp, c_fp, c_fp, c_fp, f32, c_fp, i32, c_fp]),
    "wdg_bn_apply": (i32, [c_fp, i32, c_fp, c_fp, i32, i64, i32, c_fp]),
    "wdg_bn_bwd_reduce": (i32, [c_fp, i32, c_fp, i32, c_fp, i64, i32, c_fp, c_fp]),
    "wdg_bn_bwd_apply": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, c_fp, f64, f32, c_fp, i32,
                                c_fp, c_fp, c_fp, i64, i32, c_fp]),
    "wdg_ln_fwd": (i32, [c_fp, i32, c_fp, c_fp, f32, c_fp, i32, c_fp, i64, i32, c_fp]),
    "wdg_ln_bwd": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, f32, c_fp, i32, c_fp, c_fp, c_fp, i64, i32, c_fp]),
    "wdg_lstm_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, i32, i64, i32, c_fp]),
    "wdg_lstm_bwd": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, i32, c_fp, i32, c_fp, i32, c_fp, i32,
                            i64, i32, c_fp]),
    "wdg_convlstm1_supported": (i32, [i32, i32]),
    "wdg_convlstm_gates_x_supported": (i32, [i32, i32]),
    "wdg_convlstm_gates_x": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "wdg_convlstm1_fwd": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, i32, i64, i32, i32, i32, i32, i32, c_fp]),
    "wdg_convlstm1_bwd": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, i32, i64, c_fp, c_fp, i32, i64, i32,
                                 i32, i32, i32, i32, i32, c_fp]),
    "wdg_convlstm1_bwd_dx_from": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, i32, i64, c_fp, i32, i64, i32,
                                         i32, i32, i32, i32, i32, i32, c_fp]),
    "wdg_convlstm_gates_dx_supported": (i32, [i32, i32]),
    "wdg_convlstm_gates_dx": (i32, [c_fp, c_fp, c_fp, i32, i64, i32, i32, i32, i32, i32, i32, c_fp]),
    "wdg_convlstm16_pair_supported": (i32, [C.c_void_p, C.c_void_p]),
    "wdg_convlstm16_pair_step": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32,
                                        C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, c_fp]),
    "wdg_convlstm16_pair_bwd_step": (i32, [C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32,
                                            C.c_void_p, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp]),
    "wdg_convlstm1_x2_supported": (i32, [i32, i32, i32]),
    "wdg_convlstm1_fwd_x2": (i32, [c_fp, i32, i64, c_fp, i32, i64, i32, c_fp, c_fp, c_fp, i32, i64, i32, i32, i32, i32, i32, c_fp]),
    "wdg_convlstm1_bwd_x2": (i32, [c_fp, i32, i64, c_fp, i32, i64, i32, c_fp, c_fp, c_fp, i32, i64, c_fp, i32, i64, i32,
                                    i32, i32, i32, i32, i32, c_fp, c_fp, c_fp, szt, c_fp]),
    "wdg_convlstm1_wgrad_ws_bytes": (szt, [i32, i32, i32, i32, i32]),
    "wdg_convlstm1_bwd_wgrad": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, i32, i64, c_fp, i32, i64, i32,
                                      i32, i32, i32, i32, i32, c_fp, c_fp, c_fp, szt, c_fp]),
    "wdg_convln_supported": (i32, [i32, i32]),
    "wdg_convln_fwd": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, c_fp, f32, f32, c_fp, i32, i64, c_fp, i32, i64,
                              c_fp, i32, i32, i32, i32, i32, c_fp]),
    "wdg_convln_bwd": (i32, [c_fp, i32, i64, c_fp, i32, i64, c_fp, c_fp, c_fp, f32, c_fp, c_fp, i32, i64,
                              c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "wdg_convln_wgrad_ws_bytes": (szt, [i32, i32, i32, i32]),
    "wdg_convln_bwd_x": (i32, [c_fp, i32, i64, c_fp, i32, i64, c_fp, c_fp, c_fp, f32, f32, c_fp, i32, i64, c_fp, c_fp, c_fp,
                               c_fp, c_fp, szt, i32, i32, i32, i32, i32, c_fp]),
    "wdg_upsample2x_fwd": (i32, [c_fp, i32, i64, c_fp, i32, i64, i32, i32, i32, i32, c_fp]),
    "wdg_upsample2x_bwd": (i32, [c_fp, i32, i64, c_fp, i32, i64, i32, i32, i32, i32, i32, c_fp]),
    "wdg_upconv_col_supported": (i32, [i32]),
    "wdg_upconv_col": (i32, [c_fp, i32, i64, c_fp, i32, i32, i32, i32, c_fp]),
    "wdg_upconv_gather": (i32, [c_fp, c_fp, c_fp, c_fp, i32, i64, i32, i32, i32, i32, i32, f32, c_fp, i32, c_fp]),
    "wdg_upconv_colgemm_h16_supported": (i32, [C.c_void_p]),
    "wdg_upconv_colgemm_h16": (i32, [C.c_void_p, c_fp, C.c_void_p, C.c_void_p, i32, c_fp]),
    "wdg_upconv_gather_h16": (i32, [C.c_void_p, i32, c_fp, c_fp, c_fp, i32, i64, i32, i32, i32, i32, i32, f32, c_fp]),
    "wdg_upconv_fused_h16_supported": (i32, [i32, i32]),
    "wdg_upconv_fused_h16": (i32, [c_fp, i32, i64, C.c_void_p, i32, c_fp, c_fp, c_fp, i32, i64, i32, i32, i32, i32, i32, i32, f32, i32, i32, c_fp]),
    "wdg_conv_h16_act16_supported": (i32, [C.c_void_p, i32, i32, i32]),
    "wdg_conv_fwd_h16_act16": (i32, [C.c_void_p, c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, i32, i32, f32, c_fp]),
    "wdg_conv_dgrad_h16_act16": (i32, [C.c_void_p, c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, i32, i32, f32, c_fp]),
    "wdg_conv_thin16_fwd_h16": (i32, [C.c_void_p, c_fp, i32, i64, c_fp, i32, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    "wdg_patch_gather": (i32, [c_fp, i32, i64, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "wdg_patch_scatter": (i32, [c_fp, c_fp, i32, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "wdg_dense_gap_fwd": (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "wdg_dense_gap_bwd": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp]),
    "wdg_dense_gap_bwd_ln": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, c_fp, i32, f32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "wdg_copy_channels": (i32, [c_fp, i32, i64, c_fp, i32, i64, i32, i64, i32, i32, c_fp]),
    "wdg_copy_channels_2level": (i32, [c_fp, i32, i64, i64, c_fp, i32, i64, i64, i32, i32, i64, i32, i32, c_fp]),
    "wdg_colsum": (i32, [c_fp, i32, i64, i32, c_fp, i32, c_fp]),
    "wdg_lerp_batch": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, i64, i64, i32, i32, c_fp]),
    "wdg_sumsq_batch_ch": (i32, [c_fp, i32, i64, i32, i32, i32, c_fp, c_fp]),
    "wdg_segment_meansq": (i32, [c_fp, c_fp, i32, c_fp, c_fp]),
    "wdg_metrics_pointwise": (i32, [c_fp, c_fp, i64, i32, c_fp, c_fp]),
    "wdg_lsd_reduce": (i32, [c_fp, c_fp, i64, i32, f32, c_fp, c_fp]),
    "wdg_spatial_ks_scratch_bytes": (szt, [i32, i32, i32, i32, i32]),
    "wdg_spatial_ks": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    "wdg_philox_normal": (i32, [c_fp, i32, c_fp, i32, i64, i32, u64, u64, f32, c_fp]),
    "wdg_input_assemble_supported": (i32, [i32, i32, i32]),
    "wdg_input_assemble": (i32, [c_fp, i64, i64, i32, c_fp, i32, i64, i32, i32, i32, u64, u64, f32, c_fp]),
    "wdg_input_assemble_h16": (i32, [c_fp, i64, i64, i32, c_fp, i32, i64, i32, i32, i32, u64, u64, f32, i32, i32, i32, c_fp]),
    "wdg_dp_proxy": (i32, [c_fp, c_fp, i64, i64, i32, f32, c_fp]),
    "wdg_input_assemble_slots": (i32, [c_fp, i64, i64, i32, c_fp, i32, i64, i32, i32, i32, u64, u64, f32, i32, i32, c_fp]),
    "wdg_philox_uniform": (i32, [c_fp, i64, u64, u64, c_fp]),
    "wdg_adam_tf": (i32, [c_fp, c_fp, c_fp, c_fp, i64, f32, f32, f32, f32, f32, c_fp]),
    "wdg_zero_ranges": (i32, [c_fp, C.POINTER(i64), i32, c_fp]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load libwdgan.so once; raise loudly when it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ["WDG_LIB"]) if os.environ.get("WDG_LIB") else _LIB_PATH     # (WDG_LIB: measurement builds of the same ABI)
    if not path.exists():
        raise NativeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C wind-downscaling-gan_amd/csrc`. The HIP library is required; there is no CPU path.")
    lib = C.CDLL(os.fspath(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing = broken build
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # measurement runs: WDG_TUNE="key=int,key=int" -> wdg_set_tuning before anything is planned (unknown keys raise)
    for kv in filter(None, (os.environ.get("WDG_TUNE", "") + "," + os.environ.get("WDG_TUNING", "")).split(",")):      # (WDG_TUNING: alias)
        key, _, val = kv.partition("=")
        if lib.wdg_set_tuning(key.strip().encode(), int(val)) != 0:
            raise NativeError(f"WDG_TUNE: {kv!r} rejected by wdg_set_tuning")
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().wdg_last_error().decode(errors="replace")
        raise NativeError(f"libwdgan {what} failed (status {rc}): {msg}")
