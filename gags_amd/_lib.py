"""ctypes binding of libgags_hip.so (C ABI: include/gags_raster.h).

The library is the product: there is NO fallback.  If it is missing or fails to load,
every op raises -- nothing silently routes to PyTorch or to the CPU oracle.
"""
import ctypes
import os
import subprocess

# torch must be imported BEFORE libgags_hip.so is dlopen'ed: the torch wheel bundles its own
# libamdhip64.so (soname libamdhip64.so.7) and resolves it by path.  If our library pulled in
# /opt/rocm's copy first, the process would hold two HIP runtimes and every launch on a torch
# pointer/stream would fail.  Loaded in this order, the dynamic linker binds our DT_NEEDED
# libamdhip64.so.7 to the runtime torch already loaded.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libgags_hip.so")

_vp, _i32, _i64, _f32, _f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/gags_raster.h declares
SIGNATURES = {
    "gags_abi_version": (_i32, []),
    "gags_strerror": (ctypes.c_char_p, [_i32]),
    "gags_device_count": (_i32, []),
    "gags_project_fwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _f32,
                                _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_project_fwd_raw": (_i32, [_i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _f32,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_project_bwd_raw": (_i32, [_i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp]),
    "gags_scan_scratch_bytes": (_i64, [_i32]),
    "gags_cumsum_i32": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gags_cumsum_gather_i32": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gags_read_i32": (_i32, [_vp, _vp, _vp]),
    "gags_depth_order_scratch_bytes": (_i64, [_i32]),
    "gags_depth_order": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gags_tile_emit": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "gags_tile_emit_cap": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "gags_sort_scratch_bytes": (_i64, [_i64]),
    "gags_sort_pairs": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gags_tile_offsets": (_i32, [_i64, _vp, _i32, _vp, _vp]),
    "gags_pack_isects": (_i32, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_raster_fwd_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "gags_raster_fwd": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                               _vp, _i64, _vp, _i32, _vp]),
    "gags_raster_bwd": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _i32, _vp]),
    "gags_bwd_rowmap_scratch_bytes": (_i64, [_i64]),
    "gags_bwd_rowmap_elems": (_i64, [_i64, _i32, _i32]),
    "gags_bwd_rowmap": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "gags_raster_bwd_geom_scratch_bytes": (_i64, [_i64, _i32, _i32, _i32, _i32, _i64]),
    "gags_raster_bwd_geom": (_i32, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64,
                                    _vp, _vp, _vp, _i64, _i32, _vp]),
    "gags_blended_mask": (_i32, [_i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "gags_bwd_staged_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "gags_raster_bwd_colors_staged": (_i32, [_i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                             _i64, _vp, _i32, _vp]),
    "gags_raster_bwd_colors_staged_range": (_i32, [_i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                                   _i64, _vp, _i32, _i32, _i32, _vp]),
    "gags_raster_bwd_colors_staged_cap": (_i32, [_i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                                 _i64, _vp, _i32, _i32, _i32, _vp, _vp]),
    "gags_raster_bwd_colors_staged_wire": (_i32, [_i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                                  _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gags_raster_bwd_colors_staged_keep": (_i32, [_i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp,
                                                  _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "gags_raster_list_need": (_i32, [_i32, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _vp, _vp]),
    "gags_trim_lists": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_trim_last_ids": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gags_raster_stats": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gags_project_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp]),
    "gags_sh_fwd": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_sh_bwd": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_sh_bwd_dirs": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_ed_normalize": (_i32, [_i64, _i32, _vp, _vp, _vp]),
    "gags_adam_step": (_i32, [_i64, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _i32, _vp]),
    "gags_pack_rows": (_i32, [_i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "gags_unpack_rows": (_i32, [_i64, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gags_compact_mask_scratch_bytes": (_i64, [_i32]),
    "gags_compact_mask": (_i32, [_i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "gags_compact_mask_pos": (_i32, [_i32, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gags_dot_scratch_bytes": (_i64, []),
    "gags_dot_f32": (_i32, [_i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    # include/gags_next.h
    "gags_trained_seg": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp]),
    "gags_entropy_fwd": (_i32, [_i64, _vp, _vp, _vp]),
    "gags_entropy_bwd": (_i32, [_i64, _vp, _f32, _vp, _vp]),
    "gags_entropy_bwd_dev": (_i32, [_i64, _vp, _vp, _vp, _vp]),
    "gags_segment_stats": (_i32, [_i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gags_segment_stats_multi": (_i32, [_i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "gags_segment_stats_runs_copies": (_i32, [_i64, _i32, _i32, _i32]),
    "gags_segment_stats_runs": (_i32, [_i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "gags_segment_loss": (_i32, [_i32, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_region_var_bwd_layout": (_i32, [_i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "gags_region_var_bwd_add": (_i32, [_i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gags_region_var_bwd": (_i32, [_i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gags_gather_seg_coef": (_i32, [_i64, _vp, _i32, _vp, _vp, _vp]),
    "gags_sam_clip_feature": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_sam_clip_feature_bwd_scale": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gags_distill_l1_map_fwd": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "gags_distill_l1_map_bwd": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "gags_decoder_head_distill_fwd": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_head_distill_bwd": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_head_distill_bwd_f32": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_pack_input": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp]),
    "gags_decoder_pack_layer": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_pack_layers": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_layer": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_wgrad_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "gags_decoder_wgrad": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gags_decoder_wgrad_out": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp]),
    "gags_decoder_head_bwd": (_i32, [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "gags_decoder_unpack_grad": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp]),
    "gags_decoder_head": (_i32, [_i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "gags_decoder_fwd_fused": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_bwd_fused": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_bwd_fused_scaled": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_scale_decoder_fwd_fused": (_i32, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_scale_decoder_fwd_fused_head": (_i32, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_softmax_head_bwd_y": (_i32, [_i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "gags_scale_decoder_bwd_fused": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "gags_decoder_layer_exact": (_i32, [_i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "gags_decoder_layer_split": (_i32, [_i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "gags_decoder_wgrad_split": (_i32, [_i64, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp]),
    "gags_decoder_wgrad_exact_scratch_bytes": (_i64, [_i64, _i32, _i32]),
    "gags_decoder_wgrad_exact": (_i32, [_i64, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp]),
    "gags_decoder_head_bwd_exact": (_i32, [_i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp]),
    # the "f16" decoder tier: the same kernels with IEEE half operands (include/gags_next.h), same signatures
    **{name + "_h16": None for name in ("gags_decoder_pack_layer", "gags_decoder_pack_layers", "gags_decoder_pack_input", "gags_decoder_layer", "gags_decoder_head",
                                        "gags_decoder_wgrad_scratch_bytes", "gags_decoder_wgrad", "gags_decoder_wgrad_out",
                                        "gags_decoder_head_bwd",
                                        "gags_decoder_unpack_grad", "gags_decoder_fwd_fused", "gags_decoder_bwd_fused", "gags_decoder_bwd_fused_scaled",
                                        "gags_scale_decoder_fwd_fused", "gags_scale_decoder_bwd_fused", "gags_scale_decoder_fwd_fused_head",
                                        "gags_softmax_head_bwd_y")},
    "gags_pow2_scale": (_i32, [_vp, _f32, _f32, _vp, _vp]),
    "gags_decoder_head_distill_bwd_h16": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gags_relevancy": (_i32, [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gags_relevancy_activate_scratch_bytes": (_i64, [_i32, _i32, _i32]),
    "gags_relevancy_activate": (_i32, [_i32, _i32, _i32, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
}

for _name in [k for k, v in SIGNATURES.items() if v is None]:  # (a twin's signature is its bf16 counterpart's)
    SIGNATURES[_name] = SIGNATURES[_name[:-4]]

GAGS_BWD_COLORS_ONLY = 1
GAGS_FWD_NO_MFMA = 2
GAGS_FWD_ONLY_WEIGHTS, GAGS_FWD_ONLY_FEATURES = 512, 1024  # C flags: one kernel of the split forward per call (per-kernel timing)
GAGS_RECS_BY_GAUSSIAN = 256  # C flag: `packed` is the per-Gaussian record table (gags_pack_isects with packed = NULL)
GAGS_BWD_ATOMIC = 4  # python-side: use the atomic colours-only backward instead of the staged one
GAGS_FEAT_F16 = 32  # forward: colors is an fp16 table (include/gags_raster.h)
GAGS_BWD_F32MFMA = 64  # python-side: staged backward contracts with v_mfma_f32_32x32x2_f32 (round 1-2's kernel) instead of the
#                        default fp32-equivalent split operands on the 16-bit matrix cores (csrc/raster_bwd_mfma.hip)
GAGS_BWD_BLOCKWAVES = 4096  # python-side: the staged backward's rows kernel in round 4's shape (a wave per 8x8 pixel block, rows
#                             merged in LDS: stage bit 512) instead of the default (a wave per 32 channels, rows merged in the accumulators)
GAGS_BWD_EXACT_WEIGHTS = 8192  # python-side: the default rows kernel with the weights as THREE fp16 terms (exact) and five product terms
#                                (stage bit 1024) instead of two terms / three product terms: 1.60e-7 instead of 1.68e-7 of float64, 1.33x the time
GAGS_BWD_F16SPLIT = 0   # (round 2's opt-in flag: that kernel, made exact, is the default now)
GAGS_FWD_F16MFMA = 128  # python-side: fp16 feature table + D % 128 == 0: feature pass on the 16-bit matrix cores (opt-in; C flag 64)
GAGS_FWD_EXACT = 2048  # fp32 table, D >= 128: feature pass on v_mfma_f32_32x32x2_f32, bit-identical to the sequential fmaf chain (the
#                        oracle); default: 16-bit matrix cores on operands split into three bf16 terms (include/gags_raster.h)
GAGS_FWD_FUSED = 8  # python-side: single-kernel matrix-core forward (no scratch) instead of weights + features

_lib = None


class GagsLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile gags_amd/csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j8", "libgags_hip.so"], stdout=out)
    return LIB_PATH


def load():
    """dlopen the library and type every entry point.  Raises GagsLibraryError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GagsLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gags_amd/csrc`.  gags_amd has no CPU / PyTorch fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise GagsLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GagsLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().gags_strerror(code)
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else code} ({code})")


def ptr(t):
    """Raw device pointer of a tensor (or NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
