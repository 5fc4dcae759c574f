"""On-disk formats: PLY column order pinned by the reference's own construct_list_of_attributes,
byte-level header, round trips of PLY / checkpoint tuples / language-feature .npy files.  No GPU needed."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from gags_amd import io_formats as io
from gags_amd.scene import GaussianModel

REF = "/root/reference"


def _model(n=7, d=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    return GaussianModel.from_tensors(torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g),
                                      torch.randn(n, 1, generator=g), torch.randn(n, 1, 3, generator=g),
                                      torch.randn(n, 15, 3, generator=g), torch.randn(n, d, generator=g))


# scene/gaussian_model.py:222-237 for features_dc [N,1,3], features_rest [N,15,3], scaling [N,3], rotation [N,4], D = 16
GOLDEN_NAMES = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] +
                ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)] + [f"semantic_{i}" for i in range(16)])


def test_ply_columns_follow_the_reference_attribute_list():
    assert io.ply_attribute_names(3, 45, 3, 4, 16) == GOLDEN_NAMES
    if not os.path.isdir(REF):
        return  # the reference tree exists only in the build container: there the list is checked against its own code
    for name in ("plyfile", "simple_knn", "simple_knn._C"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.path.insert(0, REF)
    try:
        from scene.gaussian_model import GaussianModel as RefModel
        m = _model()
        ref = RefModel.__new__(RefModel)
        ref._features_dc, ref._features_rest = m._features_dc, m._features_rest
        ref._scaling, ref._rotation, ref._semantic_feature = m._scaling, m._rotation, m._semantic_feature
        assert RefModel.construct_list_of_attributes(ref) == GOLDEN_NAMES
    finally:
        sys.path.remove(REF)


def test_ply_round_trip_and_header(tmp_path):
    m = _model()
    path = str(tmp_path / "point_cloud" / "iteration_30000" / "point_cloud.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n") + 11].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 7\nproperty float x\n")
    assert head.count("property float") == len(GOLDEN_NAMES)
    assert len(raw) == len(head) + 7 * 4 * len(GOLDEN_NAMES)
    names, table = io.read_ply_table(path)
    assert names == GOLDEN_NAMES
    # f_dc / f_rest are stored channel-major (transpose(1, 2) before flatten): column f_rest_1 = coefficient 1 of R
    np.testing.assert_array_equal(table["f_rest_1"], m._features_rest[:, 1, 0].numpy())
    np.testing.assert_array_equal(table["f_rest_15"], m._features_rest[:, 0, 1].numpy())
    np.testing.assert_array_equal(table["nx"], np.zeros(7, np.float32))
    back = GaussianModel(3).load_ply(path, device="cpu")
    for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_semantic_feature"):
        assert torch.equal(getattr(back, a).detach(), getattr(m, a).detach()), a
    assert back.active_sh_degree == 3 and back._semantic_feature.requires_grad


def test_ply_without_semantic_columns_and_ascii(tmp_path):
    m = _model()
    p = str(tmp_path / "rgb.ply")
    io.write_ply(p, m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation, None)
    assert io.read_ply(p)["semantic_feature"] is None
    names, table = io.read_ply_table(p)
    q = str(tmp_path / "ascii.ply")
    with open(q, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 7\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n")
        for row in table:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
    np.testing.assert_allclose(io.read_ply(q)["xyz"], m._xyz.numpy(), rtol=1e-7)


def test_checkpoint_tuples(tmp_path):
    m = _model()
    m.training_setup()
    m._semantic_feature.grad = torch.ones_like(m._semantic_feature)
    m.optimizer.step()
    path = str(tmp_path / "chkpnt30000.pth")
    io.save_checkpoint(path, m, 30000)
    args, it = io.load_checkpoint(path)
    assert it == 30000 and len(args) == 13
    r = GaussianModel(3).restore(args)
    assert torch.equal(r._semantic_feature.detach(), m._semantic_feature.detach())
    st = r.optimizer.state[r._semantic_feature]
    assert int(st["step"]) == 1 and torch.equal(st["exp_avg"], m.optimizer.state[m._semantic_feature]["exp_avg"])
    # 12-tuple (a checkpoint of the RGB field): the feature table starts from zeros with the requested width
    r2 = GaussianModel(3).restore(args[:12], semantic_dim=16)
    assert r2._semantic_feature.shape == (7, 16) and float(r2._semantic_feature.detach().abs().max()) == 0.0
    assert not r2._xyz.requires_grad and r2._semantic_feature.requires_grad
    with pytest.raises(ValueError):
        GaussianModel(3).restore(args[:5])


def test_language_feature_files(tmp_path):
    f = np.random.default_rng(0).standard_normal((5, 512)).astype(np.float32)
    s = np.random.default_rng(1).integers(-1, 5, (4, 6, 8)).astype(np.float32)
    np.save(str(tmp_path / "frame_00001_f.npy"), f)
    np.save(str(tmp_path / "frame_00001_s.npy"), s)
    e, seg = io.load_language_features(str(tmp_path / "frame_00001"))
    assert torch.equal(e, torch.from_numpy(f)) and torch.equal(seg, torch.from_numpy(s))
    e, seg = io.load_language_features(str(tmp_path / "frame_00001"), render_hw=(12, 16))
    assert seg.shape == (4, 12, 16) and set(np.unique(seg.numpy())) <= set(np.unique(s))
    np.testing.assert_array_equal(seg[:, ::2, ::2].numpy(), s)
