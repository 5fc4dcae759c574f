"""Holds oracle/gags_oracle.c to gsplat ITSELF when tests/golden/gsplat_vectors.npz exists (written by
tests/golden/make_golden_gsplat.py on a machine where `import gsplat` works; the reference's rasterizer is that pip
package, absent from the reference tree and from the build container: SURVEY 8c).  Without the file
the gsplat comparison SKIPS, loudly, and the oracle stays "parity unpinned"; the reader itself is exercised either way
against a file of the same layout written by the oracle.

Bounds: integer tensors (radii, tiles_per_gauss, isect_ids' tile part, flatten_ids, isect_offsets) equal; projections 1e-5;
renders, alphas and gradients rel-L2 <= 1e-4 (gsplat evaluates exp with the GPU's fast __expf, the oracle with a
polynomial, and sums in a different order); last_ids equal on >= 99.9 % of the pixels (threshold knife edges)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "gsplat_vectors.npz")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _scene(z, name):
    pre = name + "/"
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


def _oracle_outputs(oracle, s):
    """What the oracle computes for a stored scene, in the fixture's vocabulary."""
    w, h = int(s["width"]), int(s["height"])
    sh_degree = None if int(s["sh_degree"]) < 0 else int(s["sh_degree"])
    mode = str(s["render_mode"])
    bg = s.get("backgrounds")
    bgo = None if bg is None else (np.concatenate([bg, np.zeros(1, np.float32)]) if mode == "RGB+ED" else bg)
    out, alpha, oi = oracle.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmat"], s["K"],
                                          bg, w, h, sh_degree=sh_degree, render_mode=mode)
    res = dict(render_colors=out, render_alphas=alpha, info_radii=oi["radii"][None], info_tiles_per_gauss=oi["tiles_per_gauss"][None],
               info_isect_ids=oi["isect_ids"], info_flatten_ids=oi["flatten_ids"], info_isect_offsets=oi["isect_offsets"][None],
               info_means2d=oi["means2d"][None], info_depths=oi["depths"][None], info_conics=oi["conics"][None],
               last_ids=oi["last_ids"])
    if sh_degree is None and mode == "RGB":
        geom = "v_means" in s
        vc, vo, vm2, vcon = oracle.raster_bwd(oi["means2d"], oi["conics"], s["opacities"], s["colors"], bg, w, h,
                                              oi["isect_offsets"], oi["flatten_ids"], alpha, oi["last_ids"], s["cotangent"], None,
                                              colors_only=not geom)
        res["v_colors"] = vc
        if geom:
            vmeans, vq, vs = oracle.project_bwd(s["means"], s["quats"], s["scales"], s["viewmat"], s["K"], w, h, oi["radii"], vm2,
                                                None, vcon)
            res.update(v_means=vmeans, v_quats=vq, v_scales=vs, v_opacities=vo)
    del bgo
    return res


def _compare(got, s, name):
    """`got` = the oracle's outputs, `s` = the stored (gsplat) ones."""
    vis = np.asarray(s["info_radii"]).reshape(-1) > 0
    np.testing.assert_array_equal(np.asarray(got["info_radii"]).reshape(-1), np.asarray(s["info_radii"]).reshape(-1), err_msg=name)
    for key in ("info_tiles_per_gauss", "info_flatten_ids", "info_isect_offsets"):
        if key in s:
            np.testing.assert_array_equal(np.asarray(got[key]).reshape(-1), np.asarray(s[key]).reshape(-1), err_msg=f"{name} {key}")
    if "info_isect_ids" in s:  # tile id in the high word; the low word holds the depth's bits (both must agree)
        np.testing.assert_array_equal(np.asarray(got["info_isect_ids"]) >> 32, np.asarray(s["info_isect_ids"]) >> 32, err_msg=name)
    for key in ("info_means2d", "info_depths", "info_conics"):  # gsplat leaves culled entries undefined
        if key in s:
            a = np.asarray(got[key]).reshape(vis.size, -1)[vis]
            b = np.asarray(s[key]).reshape(vis.size, -1)[vis]
            assert _rel(a, b) <= 1e-5, (name, key, _rel(a, b))
    assert _rel(got["render_alphas"], s["render_alphas"]) <= 1e-4, (name, "alphas")
    assert _rel(got["render_colors"], s["render_colors"]) <= 1e-4, (name, "render", _rel(got["render_colors"], s["render_colors"]))
    if "last_ids" in s and "last_ids" in got:
        same = np.mean(np.asarray(got["last_ids"]) == np.asarray(s["last_ids"]))
        assert same >= 0.999, (name, "last_ids", same)
    for key in ("v_colors", "v_means", "v_quats", "v_scales", "v_opacities"):
        if key in s and key in got:
            assert _rel(got[key], s[key]) <= (1e-4 if key == "v_colors" else 1e-3), (name, key, _rel(got[key], s[key]))


def test_oracle_against_gsplat_fixture(oracle):
    if not os.path.exists(FIXTURE):
        pytest.skip("tests/golden/gsplat_vectors.npz is absent: gsplat (the reference's rasterizer, an unpinned pip package) "
                    "is not installable here, so the oracle is NOT pinned to it -- **parity unpinned**.  Run "
                    "tests/golden/make_golden_gsplat.py where `import gsplat` works and commit the file.")
    z = np.load(FIXTURE, allow_pickle=False)
    print("gsplat version of the fixture:", z["gsplat_version"])
    for name in [str(n) for n in z["scenes"]]:
        s = _scene(z, name)
        _compare(_oracle_outputs(oracle, s), s, name)


def test_fixture_reader_on_an_oracle_written_file(oracle, tmp_path):
    """The same reader and bounds on a file of the fixture's layout whose 'gsplat' side was written by the oracle: proves the
    comparison code runs (keys, shapes, every branch) before a real fixture exists, and that it FAILS on a perturbed file."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden_gsplat as mg
    from helpers import scene_arrays
    out = {"gsplat_version": np.array("oracle-self-test"), "scenes": np.array([sc[0] for sc in mg.SCENES[:3] + mg.SCENES[4:]])}
    for name, n, w, h, d, seed, view, mult, bgv, sh_degree, mode, all_grads in mg.SCENES[:3] + mg.SCENES[4:]:
        n = min(n, 600)
        sa = scene_arrays(n, max(d, 1), w, h, seed=seed, view=view, scale_mult=mult, sh=True)
        colors = sa["sh"] if sh_degree is not None else sa["colors"][:, :d]
        d_out = (3 if sh_degree is not None else d) + (1 if mode == "RGB+ED" else 0)
        rec = dict(means=sa["means"], quats=sa["quats"], scales=sa["scales"], opacities=sa["opacities"], colors=colors,
                   viewmat=sa["viewmat"], K=sa["K"], width=np.int32(w), height=np.int32(h),
                   cotangent=np.random.default_rng(seed + 100).standard_normal((h, w, d_out)).astype(np.float32),
                   sh_degree=np.int32(-1 if sh_degree is None else sh_degree), render_mode=np.array(mode))
        if bgv is not None:
            rec["backgrounds"] = np.full(3 if sh_degree is not None else d, bgv, np.float32)
        if all_grads:
            rec["v_means"] = np.zeros(1)  # (marks "every gradient wanted"; overwritten below)
        rec.update(_oracle_outputs(oracle, rec))
        for k, v in rec.items():
            out[f"{name}/{k}"] = v
    path = os.path.join(tmp_path, "gsplat_vectors.npz")
    np.savez_compressed(path, **out)
    z = np.load(path, allow_pickle=False)
    for name in [str(n) for n in z["scenes"]]:
        s = _scene(z, name)
        _compare(_oracle_outputs(oracle, s), s, name)
    bad = dict(_scene(z, "d16"))
    bad["render_colors"] = bad["render_colors"] * np.float32(1.001)
    with pytest.raises(AssertionError):
        _compare(_oracle_outputs(oracle, _scene(z, "d16")), bad, "d16-perturbed")
