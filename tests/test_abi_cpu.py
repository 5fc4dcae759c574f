"""The C-ABI shared library loads without a GPU and exports every symbol include/*.h declares;
the Python binding types exactly that set; argument validation returns error codes (no compute
is launched here).  Also: the product package never imports the oracle."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        if os.path.basename(h) == "gags_cpu.h":  # the CPU twins live in oracle/libgags_oracle.so (test infrastructure)
            continue
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(gags_\w+)\s*\(", src))
    return names


@pytest.fixture(scope="module")
def lib():
    from gags_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from gags_amd import _lib
    decl = _declared()
    assert len(decl) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(raw, name), f"{name} declared in include/gags_raster.h but not exported"
    assert set(_lib.SIGNATURES) == decl, "python binding and header disagree"


def test_library_exports_nothing_but_the_declared_c_abi(lib):
    """Built with -fvisibility=hidden: the dynamic symbol table holds the headers' gags_* functions and no internal
    C++ launcher (17 mangled _Z..._launch symbols used to leak)."""
    import subprocess
    from gags_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TW"}
    assert exported == _declared(), sorted(exported ^ _declared())


def test_abi_version_and_error_strings(lib):
    assert lib.gags_abi_version() == 2
    assert lib.gags_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert lib.gags_strerror(code) not in (b"ok", b"unknown error")
    assert lib.gags_strerror(-99) == b"unknown error"


def test_argument_validation_returns_codes_without_launching(lib):
    assert lib.gags_project_fwd(-1, *([None] * 5), 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 5), None) == -1
    assert lib.gags_project_fwd(8, *([None] * 5), 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 5), None) == -1
    assert lib.gags_raster_fwd(0, 4, 16, 16, *([None] * 7), 0, None, None, None, None, None, 0, None, 0, None) == -1
    assert lib.gags_raster_fwd_scratch_bytes(1000, 64, 48) >= 4 * 1000 * 260
    assert lib.gags_sort_pairs(4, 40, 0, None, None, None, None, None, 0, None) == -1
    assert lib.gags_depth_order(-1, None, None, None, None, None, 0, None) == -1
    assert lib.gags_depth_order_scratch_bytes(1000) > 2 * 1000 * 4
    assert lib.gags_sort_scratch_bytes(1000) > 1000 * 12
    assert lib.gags_scan_scratch_bytes(100000) >= 4
    assert lib.gags_cumsum_i32(-5, None, None, None, None, 0, None) == -1
    assert lib.gags_sh_fwd(4, 16, 7, None, None, None, None, None, None) == -1
    # round 4's entries: the stored-parameter projection, the permuted prefix sum, the row-list compaction, the split tiers
    assert lib.gags_project_fwd_raw(-1, *([None] * 4), 1.0, None, None, 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 9), None) == -1
    assert lib.gags_project_fwd_raw(8, *([None] * 4), 1.0, None, None, 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 9), None) == -1
    assert lib.gags_project_bwd_raw(8, *([None] * 4), 1.0, None, None, 16, 16, 0.3, *([None] * 9), None) == -1
    assert lib.gags_cumsum_gather_i32(-1, None, None, None, None, None, 0, None) == -1
    assert lib.gags_cumsum_gather_i32(8, None, None, None, None, None, 0, None) == -1
    assert lib.gags_compact_mask(-1, None, 0, None, None, None, 0, None) == -1
    assert lib.gags_compact_mask(8, None, 8, None, None, None, 0, None) == -1
    assert lib.gags_compact_mask_scratch_bytes(100000) >= 4 * 49
    # round 6's entries: list trimming, the inverse row map, the reduce stage's wire / persistent-buffer flavours
    assert lib.gags_raster_list_need(4, 16, 16, None, None, 10, None, 0, None, None) == -1
    assert lib.gags_raster_list_need(-1, 16, 16, None, None, 0, None, 0, None, None) == -1
    assert lib.gags_trim_lists(16, 16, None, None, None, None, None, None) == -1
    assert lib.gags_trim_lists(0, 16, None, None, None, None, None, None) == -1
    assert lib.gags_trim_last_ids(16, 16, None, None, None, None, None) == -1
    assert lib.gags_compact_mask_pos(-1, None, 0, None, None, None, None, 0, None) == -1
    assert lib.gags_compact_mask_pos(8, None, 1 << 31, None, None, None, None, 0, None) == -1   # capacity past int32 positions
    staged = (128, 8, 16, 16, None, 0, None, None, None, 0, None, 0, None, 0, None, 3, 0, 128)
    assert lib.gags_raster_bwd_colors_staged_wire(*staged, None, None, None, None, None) == -1
    assert lib.gags_raster_bwd_colors_staged_keep(*staged, None, None, None) == -1            # both flag arrays are required
    # the staged backward wants the forward's per-block slot counts 16-byte aligned (one scalar load per tile)
    import ctypes
    buf = (ctypes.c_int32 * 64)()
    base = ctypes.addressof(buf)
    al = base + (-base % 16)
    P = ctypes.c_void_p
    common = dict(pre=(128, 8, 16, 16, P(al), 0, P(al)), post=(P(al), 0, P(al), 0, P(al), 0, P(al), 3))
    assert lib.gags_raster_bwd_colors_staged(*common["pre"], P(al + 4), *common["post"], None) == -1
    assert lib.gags_raster_bwd_colors_staged(*common["pre"], P(al), *common["post"], None) == -3   # (aligned: on to the scratch-size check)
    assert lib.gags_decoder_layer_split(8, 4, 4, *([None] * 2), 4, *([None] * 2), 1, *([None] * 4), 4, 5, None) == -1   # terms = 5
    assert lib.gags_decoder_wgrad_split(8, 4, 4, None, 4, None, None, 4, None, None, None, 0, 1, None) == -1            # terms = 1
    # round 6, second half: the iteration's entries
    P8 = P(al)
    assert lib.gags_decoder_wgrad_out(8, 32, 32, P8, P8, None, P8, None, 33, 16, None, None, 0, None) == -1              # co > n_out
    assert lib.gags_decoder_wgrad_out_h16(8, 32, 32, P8, P8, None, P8, None, 16, 0, None, None, 0, None) == -1           # ci = 0
    assert lib.gags_decoder_pack_layers(13, P8, P8, P8, P8, P8, P8, P8, P8, P8, None) == -1                              # > 12 layers
    assert lib.gags_decoder_bwd_fused_scaled(8, 64, 512, P8, P8, P8, P8, None, None, None) == -1                         # c_in > 32
    assert lib.gags_scale_decoder_fwd_fused_head(8, 16, P8, P8, P8, None, None, None, None, None) == -1                  # no output at all
    assert lib.gags_softmax_head_bwd_y(8, 5, 32, P8, P8, P8, None) == -1                                                 # more than 4 channels
    assert lib.gags_pow2_scale(None, 1.0, 12.0, P8, None) == -1 and lib.gags_pow2_scale(P8, 0.0, 12.0, P8, None) == -1
    assert lib.gags_entropy_bwd_dev(8, P8, None, P8, None) == -1
    assert lib.gags_segment_loss(0, 4, 2, 1, 64, P8, P8, P8, P8, P8, P8, P8, P8, None, None) == -1                       # mode 0 is c == 1
    assert lib.gags_segment_loss(1, 4, 2, 1, 64, P8, P8, P8, P8, P8, P8, P8, P8, None, None) == -1                       # mode 1 needs `mean`
    assert lib.gags_region_var_bwd_add(8, 6, P8, P8, 4, P8, P8, P8, P8, None) == -1                                      # c % 4
    # (host arithmetic only) the run-length moments kernel: copies = workgroups, 0 = shape not served
    assert lib.gags_segment_stats_runs_copies(1920 * 1080, 16, 300, 1) == 512
    assert lib.gags_segment_stats_runs_copies(1920 * 1080, 1, 300, 0) == 254
    assert lib.gags_segment_stats_runs_copies(1000, 16, 300, 1) == 1
    assert lib.gags_segment_stats_runs_copies(1920 * 1080, 16, 300, 0) == 0        # channel-major 16 channels
    assert lib.gags_segment_stats_runs_copies(1920 * 1080, 16, 20000, 1) == 0      # table beyond LDS
    assert lib.gags_segment_stats_runs(64, 16, P8, P8, 300, 7, P8, P8, P8, 1, None) == -1   # copies must be what _copies says


def test_intersection_cap_covers_the_slot_space(lib):
    """GAGS_MAX_ISECTS alone is not the limit: the slot space of a view is 4 I + 64 tiles + 64 slots and its id table is
    indexed with 32-bit BYTE offsets (slots < 2^30).  Near the cap the per-tile slack would wrap them: the raster
    entries refuse such a count with GAGS_EINVAL before looking at anything else; a count that fits passes validation (and
    stops at the scratch-size check here: nothing is launched)."""
    import ctypes
    dummy = ctypes.create_string_buffer(64)
    P = ctypes.cast(dummy, ctypes.c_void_p)
    w = h = 16  # one tile: slots = 4 I + 128

    def fwd(n_isects):
        return lib.gags_raster_fwd(16, 4, w, h, P, P, P, P, None, P, P, n_isects, P, P, P, P, P, 0, P, 0, None)
    assert fwd((1 << 28) - 64) == -3      # GAGS_ESCRATCH: validated, scratch_bytes = 0 is too small
    assert fwd((1 << 28) - 1) == -1       # below GAGS_MAX_ISECTS, slot space 2^30 + 124
    assert fwd(1 << 28) == -1
    # at 1080p (8160 tiles) the slack is 522 304 slots: 130 576 intersections below the cap
    def fwd1080(n_isects):
        return lib.gags_raster_fwd(16, 4, 1920, 1080, P, P, P, P, None, P, P, n_isects, P, P, P, P, P, 0, P, 0, None)
    assert fwd1080((1 << 28) - 130_600) == -3
    assert fwd1080((1 << 28) - 130_500) == -1
    from gags_amd import rasterization
    with pytest.raises(RuntimeError, match="tile intersections"):
        rasterization._check_isects((1 << 28) - 130_500, 8160)
    rasterization._check_isects((1 << 28) - 130_600, 8160)


def test_missing_library_raises_loudly(monkeypatch, tmp_path):
    from gags_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgags_hip.so"))
    with pytest.raises(_lib.GagsLibraryError):
        _lib.load()


def test_product_never_touches_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "gags_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".h", "Makefile")):
            src = open(path).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
            assert "libgags_oracle" not in src and "gags_oracle.c" not in src, path


def test_cpu_tensors_are_rejected_not_rerouted():
    import torch
    from gags_amd.rasterization import rasterization
    with pytest.raises(RuntimeError, match="no CPU path"):
        rasterization(torch.zeros(4, 3), torch.zeros(4, 4), torch.zeros(4, 3), torch.zeros(4), torch.zeros(4, 3),
                      torch.eye(4)[None], torch.eye(3)[None], 16, 16)


def test_cpu_twins_share_the_c_abi_signatures(oracle):
    """"CPU twins with identical signatures".  oracle/libgags_oracle.so exports gags_cpu_<name> for the core
    entry points (include/gags_cpu.h); each is typed here with the ctypes signature _lib.SIGNATURES holds for gags_<name>
    -- the very table the product binds the GPU library with -- and the whole operator is driven through them on host
    pointers: projection, prefix sum, emit, sort, offsets, forward, backward.  Result == the oracle's own front-end."""
    import ctypes
    import re
    import numpy as np
    from gags_amd import _lib
    from helpers import scene_arrays
    here = os.path.dirname(os.path.abspath(__file__))
    strip = lambda txt: re.sub(r"/\*.*?\*/", "", txt, flags=re.S)  # noqa: E731
    header = strip(open(os.path.join(here, "..", "include", "gags_cpu.h")).read())
    gh = strip(open(os.path.join(here, "..", "include", "gags_raster.h")).read())
    names = sorted(set(re.findall(r"\b(gags_cpu_\w+)\s*\(", header)))
    assert len(names) >= 14
    cpu = ctypes.CDLL(os.path.join(here, "..", "oracle", "libgags_oracle.so"))
    fn = {}
    for name in names:
        gpu_name = name.replace("gags_cpu_", "gags_")
        assert gpu_name in _lib.SIGNATURES, f"{name} has no GPU counterpart in the C ABI"
        f = getattr(cpu, name)  # exported
        f.restype, f.argtypes = _lib.SIGNATURES[gpu_name]
        fn[gpu_name] = f
        # ... and the header declares the twin with the GPU entry's parameter list, token for token
        norm = lambda txt: re.sub(r"\s+", "", txt)  # noqa: E731
        decl_cpu = re.search(re.escape(name) + r"\s*\((.*?)\)\s*;", header, re.S).group(1)
        decl_gpu = re.search(r"\b" + re.escape(gpu_name) + r"\s*\((.*?)\)\s*;", gh, re.S).group(1)
        assert norm(decl_cpu) == norm(decl_gpu), name

    def P(a):
        return None if a is None else ctypes.c_void_p(a.ctypes.data)

    n, w, h, d = 1500, 96, 64, 16
    s = scene_arrays(n, d, w, h, seed=3, view=2, scale_mult=6.0)
    bg = np.full(d, 0.25, np.float32)
    tw, th = (w + 15) // 16, (h + 15) // 16
    radii, tiles = np.zeros(n, np.int32), np.zeros(n, np.int32)
    m2d, depths, conics = np.zeros((n, 2), np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    means, quats, scales, opac, cols = (f32(s[k]) for k in ("means", "quats", "scales", "opacities", "colors"))
    vm, K = f32(s["viewmat"]), f32(s["K"])
    assert fn["gags_project_fwd"](n, P(means), P(quats), P(scales), P(vm), P(K), w, h, 0.3, 0.01, 1e10, 0.0, P(radii), P(m2d),
                                  P(depths), P(conics), P(tiles), None) == 0
    cum, total = np.zeros(n, np.int32), np.zeros(1, np.int32)
    assert fn["gags_cumsum_i32"](n, P(tiles), P(cum), P(total), None, 0, None) == 0
    ni = int(total[0])
    ids, flat = np.zeros(ni, np.int64), np.zeros(ni, np.int32)
    assert fn["gags_tile_emit"](n, P(m2d), P(radii), P(depths), P(cum), None, tw, th, P(ids), P(flat), None) == 0
    ids_s, flat_s = np.zeros_like(ids), np.zeros_like(flat)
    assert fn["gags_sort_pairs"](ni, max(1, (tw * th).bit_length()), 0, P(ids), P(flat), P(ids_s), P(flat_s), None, 0, None) == 0
    offsets = np.zeros(tw * th + 1, np.int32)
    assert fn["gags_tile_offsets"](ni, P(ids_s), tw * th, P(offsets), None) == 0
    assert offsets[-1] == ni
    out, alphas, last = np.zeros((h, w, d), np.float32), np.zeros((h, w), np.float32), np.zeros((h, w), np.int32)
    assert fn["gags_raster_fwd"](d, n, w, h, P(m2d), P(conics), P(opac), P(cols), P(bg), P(offsets), P(flat_s), ni, None, P(out),
                                 P(alphas), P(last), None, 0, None, 0, None) == 0
    v_out = np.random.default_rng(5).standard_normal((h, w, d)).astype(np.float32)
    v_cols = np.zeros((n, d), np.float32)
    assert fn["gags_raster_bwd"](d, w, h, P(m2d), P(conics), P(opac), P(cols), P(bg), P(offsets), P(flat_s), ni, None, P(alphas),
                                 P(last), P(v_out), None, P(v_cols), None, None, None, _lib.GAGS_BWD_COLORS_ONLY, None) == 0
    o_out, o_alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"],
                                              s["K"], bg, w, h)
    np.testing.assert_array_equal(radii, oi["radii"])
    np.testing.assert_array_equal(ids_s, oi["isect_ids"])
    np.testing.assert_array_equal(flat_s, oi["flatten_ids"])
    np.testing.assert_array_equal(offsets[:-1].reshape(th, tw), oi["isect_offsets"])
    np.testing.assert_array_equal(out, o_out)
    np.testing.assert_array_equal(last, oi["last_ids"])
    o_vc, _, _, _ = oracle.raster_bwd(oi["means2d"], oi["conics"], s["opacities"], s["colors"], bg, w, h, oi["isect_offsets"],
                                      oi["flatten_ids"], o_alpha, oi["last_ids"], v_out, None, colors_only=True)
    np.testing.assert_allclose(v_cols, o_vc, rtol=0, atol=1e-5 * float(np.abs(o_vc).max()))  # (per-tile sums added in thread order)
    # error convention of the ABI: a bad argument is a code, not a crash
    assert fn["gags_raster_fwd"](0, n, w, h, None, None, None, None, None, None, None, 0, None, None, None, None, None, 0, None, 0,
                                 None) == -1


def test_python_side_raster_flags_are_distinct_bits():
    """render(..., raster_flags=) takes an OR of these: each selectable behaviour needs a bit of its own (two of them sharing one
    would silently select both kernels' worth of behaviour), and the retired GAGS_BWD_F16SPLIT stays 0."""
    from gags_amd import _lib
    names = ["GAGS_BWD_COLORS_ONLY", "GAGS_FWD_NO_MFMA", "GAGS_BWD_ATOMIC", "GAGS_FWD_FUSED", "GAGS_FEAT_F16", "GAGS_BWD_F32MFMA",
             "GAGS_FWD_F16MFMA", "GAGS_RECS_BY_GAUSSIAN", "GAGS_FWD_ONLY_WEIGHTS", "GAGS_FWD_ONLY_FEATURES", "GAGS_FWD_EXACT",
             "GAGS_BWD_BLOCKWAVES", "GAGS_BWD_EXACT_WEIGHTS"]
    vals = [getattr(_lib, n) for n in names]
    assert all(v > 0 and v & (v - 1) == 0 for v in vals), dict(zip(names, vals))
    assert len(set(vals)) == len(vals), dict(zip(names, vals))
    assert _lib.GAGS_BWD_F16SPLIT == 0
