"""Shared input builders for the parity tests (seeded, CPU-generated so that the oracle and
the GPU see bit-identical inputs)."""
import numpy as np
import torch

from gags_amd import synthetic as syn


def scene_arrays(n, d, width, height, seed=0, view=None, scale_mult=1.0, sh=False):
    """Activated parameters as numpy arrays + camera matrices (what crosses the rasterization boundary)."""
    cam = syn.make_camera(width, height, view=view, device="cpu")
    p = syn.make_gaussians(n, d, width, height, seed=seed, scale0=syn.SCALE0 * scale_mult)
    vm, K = syn.camera_matrices(cam)
    out = dict(
        means=p["xyz"].numpy(),
        quats=torch.nn.functional.normalize(p["rotation"]).numpy(),
        scales=p["scaling_log"].exp().numpy(),
        opacities=torch.sigmoid(p["opacity_logit"]).reshape(-1).numpy(),
        colors=None if d == 0 else p["semantic_feature"].numpy(),
        sh=torch.cat([p["features_dc"], p["features_rest"]], dim=1).numpy(),
        viewmat=vm.numpy().copy(), K=K, cam=cam, raw=p)
    return out


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def to_dev(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
