"""ctypes front-end of the CPU oracle (oracle/gags_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (gags_amd/) must never import this module.

Parity status: **parity unpinned** (see the header of gags_oracle.c): the reference's
rasterizer is the absent, unpinned pip package gsplat, so this restates the published
gsplat-1.4-style algorithm declared in SURVEY.md Appendix A and anchors on the reference
call site /root/reference/gaussian_renderer/__init__.py:27-85.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgags_oracle.so")
_lib = None

TILE = 16


def build(force=False):
    """Compile libgags_oracle.so with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("gags_oracle.c", "gags_cpu.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgags_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_exp_neg_scalar.restype = ctypes.c_float
        _lib.orc_exp_neg_scalar.argtypes = [ctypes.c_float]
        _lib.orc_cumsum.restype = ctypes.c_int64
        _lib.orc_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"], "oracle wants contiguous arrays"
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def max_threads():
    return int(lib().orc_max_threads())


def exp_neg(sigma):
    l = lib()
    return np.array([l.orc_exp_neg_scalar(ctypes.c_float(float(s))) for s in np.ravel(sigma)], dtype=np.float32)


def project_fwd(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near=0.01, far=1e10,
                radius_clip=0.0):
    means, quats, scales = _f32(means), _f32(quats), _f32(scales)
    viewmat, K = _f32(viewmat).reshape(4, 4), _f32(K).reshape(3, 3)
    N = means.shape[0]
    radii = np.zeros(N, np.int32)
    means2d = np.zeros((N, 2), np.float32)
    depths = np.zeros(N, np.float32)
    conics = np.zeros((N, 3), np.float32)
    lib().orc_project_fwd(ctypes.c_int(N), _p(means), _p(quats), _p(scales), _p(viewmat), _p(K),
                          ctypes.c_int(width), ctypes.c_int(height), ctypes.c_float(eps2d),
                          ctypes.c_float(near), ctypes.c_float(far), ctypes.c_float(radius_clip),
                          _p(radii), _p(means2d), _p(depths), _p(conics))
    return radii, means2d, depths, conics


def tile_bin(means2d, radii, depths, width, height):
    """K4-K8: returns tiles_per_gauss, isect_ids (sorted), flatten_ids (sorted), isect_offsets."""
    l = lib()
    N = radii.shape[0]
    tile_w, tile_h = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    tpg = np.zeros(N, np.int32)
    l.orc_tile_count(ctypes.c_int(N), _p(means2d), _p(radii), ctypes.c_int(tile_w), ctypes.c_int(tile_h), _p(tpg))
    cum = np.zeros(N, np.int64)
    n_isects = int(l.orc_cumsum(ctypes.c_int(N), _p(tpg), _p(cum)))
    ids = np.zeros(max(n_isects, 1), np.int64)
    flat = np.zeros(max(n_isects, 1), np.int32)
    l.orc_tile_emit(ctypes.c_int(N), _p(means2d), _p(radii), _p(depths), _p(cum), ctypes.c_int(tile_w),
                    ctypes.c_int(tile_h), _p(ids), _p(flat))
    n_tiles = tile_w * tile_h
    tile_bits = max(1, int(n_tiles - 1).bit_length())
    ids_s = np.zeros_like(ids)
    flat_s = np.zeros_like(flat)
    l.orc_sort_pairs(ctypes.c_int64(n_isects), ctypes.c_int(32 + tile_bits), _p(ids), _p(flat), _p(ids_s), _p(flat_s))
    offsets = np.zeros(n_tiles, np.int32)
    l.orc_tile_offsets(ctypes.c_int64(n_isects), _p(ids_s), ctypes.c_int(n_tiles), _p(offsets))
    return dict(tiles_per_gauss=tpg, n_isects=n_isects, isect_ids_unsorted=ids[:n_isects],
                flatten_ids_unsorted=flat[:n_isects], isect_ids=ids_s[:n_isects],
                flatten_ids=flat_s[:n_isects], isect_offsets=offsets.reshape(tile_h, tile_w),
                tile_width=tile_w, tile_height=tile_h)


def raster_fwd(means2d, conics, opacities, colors, backgrounds, width, height, isect_offsets, flatten_ids,
               tile_begin=0, tile_step=1):
    colors = _f32(colors)
    D = colors.shape[1]
    tile_h, tile_w = isect_offsets.shape
    out = np.zeros((height, width, D), np.float32)
    alphas = np.zeros((height, width), np.float32)
    last = np.zeros((height, width), np.int32)
    n_eval = ctypes.c_int64(0)
    n_blend = ctypes.c_int64(0)
    flat = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    bg = None if backgrounds is None else _f32(backgrounds)
    lib().orc_raster_fwd(ctypes.c_int(D), ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(tile_w),
                         ctypes.c_int(tile_h), _p(_f32(means2d)), _p(_f32(conics)), _p(_f32(opacities)),
                         _p(colors), _p(bg), _p(np.ascontiguousarray(isect_offsets, dtype=np.int32)), _p(flat),
                         ctypes.c_int64(len(flatten_ids)), ctypes.c_int(tile_begin), ctypes.c_int(tile_step),
                         _p(out), _p(alphas), _p(last), ctypes.byref(n_eval), ctypes.byref(n_blend))
    return out, alphas, last, dict(n_eval=n_eval.value, n_blend=n_blend.value)


def raster_fwd_acc64(means2d, conics, opacities, colors, backgrounds, width, height, isect_offsets, flatten_ids,
                     tile_begin=0, tile_step=1):
    """The forward with its colour sums formed in float64 on the SAME fp32 weights (orc_raster_fwd_acc64): the yardstick
    for fp32-equivalent contraction kernels.  Returns render_colors [H,W,D] float64 (tiles outside the subset: zeros)."""
    colors = _f32(colors)
    D = colors.shape[1]
    tile_h, tile_w = isect_offsets.shape
    out = np.zeros((height, width, D), np.float64)
    flat = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    bg = None if backgrounds is None else _f32(backgrounds)
    lib().orc_raster_fwd_acc64(ctypes.c_int(D), ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(tile_w),
                               ctypes.c_int(tile_h), _p(_f32(means2d)), _p(_f32(conics)), _p(_f32(opacities)),
                               _p(colors), _p(bg), _p(np.ascontiguousarray(isect_offsets, dtype=np.int32)), _p(flat),
                               ctypes.c_int64(len(flatten_ids)), ctypes.c_int(tile_begin), ctypes.c_int(tile_step),
                               _p(out))
    return out


def raster_bwd(means2d, conics, opacities, colors, backgrounds, width, height, isect_offsets, flatten_ids,
               render_alphas, last_ids, v_render_colors, v_render_alphas=None, colors_only=False,
               tile_begin=0, tile_step=1):
    colors = _f32(colors)
    N, D = colors.shape
    tile_h, tile_w = isect_offsets.shape
    v_colors = np.zeros((N, D), np.float32)
    if colors_only:
        v_opac = v_m2d = v_con = None
    else:
        v_opac = np.zeros(N, np.float32)
        v_m2d = np.zeros((N, 2), np.float32)
        v_con = np.zeros((N, 3), np.float32)
    flat = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    bg = None if backgrounds is None else _f32(backgrounds)
    va = None if v_render_alphas is None else _f32(v_render_alphas)
    lib().orc_raster_bwd(ctypes.c_int(D), ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(tile_w),
                         ctypes.c_int(tile_h), _p(_f32(means2d)), _p(_f32(conics)), _p(_f32(opacities)),
                         _p(colors), _p(bg), _p(np.ascontiguousarray(isect_offsets, dtype=np.int32)), _p(flat),
                         ctypes.c_int64(len(flatten_ids)), _p(_f32(render_alphas)),
                         _p(np.ascontiguousarray(last_ids, dtype=np.int32)), _p(_f32(v_render_colors)), _p(va),
                         ctypes.c_int(tile_begin), ctypes.c_int(tile_step),
                         _p(v_colors), _p(v_opac), _p(v_m2d), _p(v_con))
    return v_colors, v_opac, v_m2d, v_con


def raster_bwd_colors_fwdorder(means2d, conics, opacities, d, width, height, isect_offsets, flatten_ids,
                               v_render_colors, n, tile_begin=0, tile_step=1):
    """Colours-only gradient with alpha*T recomputed front to back (see gags_oracle.c)."""
    tile_h, tile_w = isect_offsets.shape
    v_colors = np.zeros((n, d), np.float32)
    flat = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    lib().orc_raster_bwd_colors_fwdorder(ctypes.c_int(d), ctypes.c_int(width), ctypes.c_int(height),
                                         ctypes.c_int(tile_w), ctypes.c_int(tile_h), _p(_f32(means2d)),
                                         _p(_f32(conics)), _p(_f32(opacities)),
                                         _p(np.ascontiguousarray(isect_offsets, dtype=np.int32)), _p(flat),
                                         ctypes.c_int64(len(flatten_ids)), _p(_f32(v_render_colors)),
                                         ctypes.c_int(tile_begin), ctypes.c_int(tile_step), _p(v_colors))
    return v_colors


def project_bwd(means, quats, scales, viewmat, K, width, height, radii, v_means2d, v_depths, v_conics, eps2d=0.3):
    means, quats, scales = _f32(means), _f32(quats), _f32(scales)
    viewmat, K = _f32(viewmat).reshape(4, 4), _f32(K).reshape(3, 3)
    N = means.shape[0]
    v_means = np.zeros((N, 3), np.float32)
    v_quats = np.zeros((N, 4), np.float32)
    v_scales = np.zeros((N, 3), np.float32)
    vd = None if v_depths is None else _f32(v_depths)
    lib().orc_project_bwd(ctypes.c_int(N), _p(means), _p(quats), _p(scales), _p(viewmat), _p(K),
                          ctypes.c_int(width), ctypes.c_int(height), ctypes.c_float(eps2d),
                          _p(np.ascontiguousarray(radii, dtype=np.int32)), _p(_f32(v_means2d)), _p(vd),
                          _p(_f32(v_conics)), _p(v_means), _p(v_quats), _p(v_scales))
    return v_means, v_quats, v_scales


def sh_fwd(deg, means, campos, coeffs, radii=None):
    means, coeffs, campos = _f32(means), _f32(coeffs), _f32(campos)
    N, Kc = coeffs.shape[0], coeffs.shape[1]
    out = np.zeros((N, 3), np.float32)
    r = None if radii is None else np.ascontiguousarray(radii, dtype=np.int32)
    lib().orc_sh_fwd(ctypes.c_int(N), ctypes.c_int(Kc), ctypes.c_int(deg), _p(means), _p(campos), _p(coeffs),
                     _p(r), _p(out))
    return out


def rasterization(means, quats, scales, opacities, colors, viewmat, K, backgrounds, width, height,
                  sh_degree=None, render_mode="RGB", tile_begin=0, tile_step=1):
    """CPU restatement of `gsplat.rasterization(...)` as the reference calls it
    (/root/reference/gaussian_renderer/__init__.py:56-70), one camera, packed=False.
    Returns (render_colors [H,W,D'], render_alphas [H,W], info dict with every intermediate)."""
    radii, means2d, depths, conics = project_fwd(means, quats, scales, viewmat, K, width, height)
    opacities = _f32(opacities).reshape(-1)
    if sh_degree is not None:
        vm = np.asarray(viewmat, np.float64).reshape(4, 4)
        campos = np.linalg.inv(vm)[:3, 3].astype(np.float32)
        cols = sh_fwd(sh_degree, means, campos, colors, radii)
    else:
        cols = _f32(colors)
    bg = None if backgrounds is None else _f32(backgrounds).reshape(-1)
    if render_mode in ("RGB+ED", "RGB+D"):
        cols = np.concatenate([cols, depths[:, None]], axis=1)
        if bg is not None:
            bg = np.concatenate([bg, np.zeros(1, np.float32)])
    elif render_mode in ("ED", "D"):
        cols = depths[:, None].copy()
        bg = None if bg is None else np.zeros(1, np.float32)
    b = tile_bin(means2d, radii, depths, width, height)
    out, alphas, last, stats = raster_fwd(means2d, conics, opacities, cols, bg, width, height,
                                          b["isect_offsets"], b["flatten_ids"], tile_begin, tile_step)
    if render_mode in ("RGB+ED", "ED"):
        out[..., -1] = out[..., -1] / np.maximum(alphas, np.float32(1e-10))
    info = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opacities,
                colors=cols, backgrounds=bg, last_ids=last, width=width, height=height, tile_size=TILE,
                n_cameras=1, **b, **stats)
    return out, alphas, info


def adam_step(p, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-15, step=1):
    """In-place Adam step on float32 numpy arrays (R9: scene/gaussian_model.py:208, train.py:221-223)."""
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    lib().orc_adam_step(ctypes.c_int64(p.size), _p(p), _p(g), _p(m), _p(v), ctypes.c_double(lr),
                        ctypes.c_double(beta1), ctypes.c_double(beta2), ctypes.c_double(eps), ctypes.c_int(step))
    return p, m, v
