"""Host-side mirrors of the reference types that cross the render boundary.

Only what `render(...)` reads is mirrored:

* `GaussianModel` -- tensor layout and activation getters of
  /root/reference/scene/gaussian_model.py:48-61,116-139 (`_xyz [N,3]`, `_features_dc [N,1,3]`,
  `_features_rest [N,15,3]`, `_scaling [N,3]` log-space, `_rotation [N,4]` wxyz un-normalised,
  `_opacity [N,1]` logit, `_semantic_feature [N,D]` raw), and `training_setup` of :183-208
  (only `_semantic_feature` is optimised, Adam eps=1e-15, every geometry tensor frozen).
* `Camera` -- the five attributes render() touches (scene/cameras.py:17-61):
  `FoVx, FoVy, image_width, image_height` (mutable: render.py:115-116) and
  `world_view_transform` = W2C^T, built as utils/graphics_utils.py:38-49 does.

PLY / checkpoint I/O (`save_ply`, `load_ply`, `capture`, `restore`: SURVEY.md 8f N3) go through
gags_amd/io_formats.py; densification and COLMAP loading are out of scope.
"""
import math

import numpy as np
import torch
from torch import nn


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World-to-view 4x4 (float32) from camera rotation R (stored transposed, as COLMAP
    readers do) and translation t; same contract as utils/graphics_utils.py:38-49."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = np.asarray(t)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


class Camera:
    """Minimal stand-in for scene.cameras.Camera / MiniCam."""

    def __init__(self, R, T, FoVx, FoVy, image_width, image_height, device="cuda", uid=0):
        self.uid = uid
        self.R = np.asarray(R, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.FoVx = float(FoVx)
        self.FoVy = float(FoVy)
        self.image_width = int(image_width)
        self.image_height = int(image_height)
        self.zfar = 100.0
        self.znear = 0.01
        w2c = getWorld2View2(self.R, self.T)
        self.world_view_transform = torch.tensor(w2c).transpose(0, 1).contiguous().to(device)
        self.camera_center = torch.tensor(np.linalg.inv(w2c.astype(np.float64))[:3, 3], dtype=torch.float32).to(device)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianModel:
    """Parameter container with the reference's tensor names, shapes and getters."""

    def __init__(self, sh_degree=3):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self._semantic_feature = None
        self.optimizer = None

    # -- activations: scene/gaussian_model.py:36-42,116-139 -------------------------------
    def cache_activations(self, on=True):
        """Opt-in: keep the activations of FROZEN parameters (requires_grad False: the geometry during the reference's
        feature training, train.py:62-75) between renders instead of re-evaluating exp / normalize / sigmoid on every
        getter call as scene/gaussian_model.py:116-139 does (six small kernels per view).  OFF by default: a cached value
        cannot see writes that bypass autograd's version counter (`p.data.add_(...)`, `p.data.copy_(...)` -- idiomatic
        in 3DGS code for opacity resets, clamping and weight loading), so whoever turns this on promises to call
        `invalidate_activations()` after such a write.  In-place ops on the parameter itself, replacing the parameter,
        load_ply / restore, and requires_grad are noticed without help."""
        self._cache_on = bool(on)
        self.invalidate_activations()
        return self

    def invalidate_activations(self):
        self.__dict__["_act_cache"] = {}

    def _frozen(self, name, param, fn):
        """Activation of a parameter: evaluated afresh on every call (the reference's behaviour) unless
        cache_activations() was turned on and the parameter is frozen."""
        if (not getattr(self, "_cache_on", False) or param.requires_grad
                or torch.is_grad_enabled() and param.grad_fn is not None):
            return fn(param)
        cache = self.__dict__.setdefault("_act_cache", {})
        key = (id(param), param.data_ptr(), param._version, tuple(param.shape))
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, fn(param))
            cache[name] = hit
        return hit[1]

    @property
    def get_scaling(self):
        return self._frozen("scaling", self._scaling, torch.exp)

    @property
    def get_rotation(self):
        return self._frozen("rotation", self._rotation, torch.nn.functional.normalize)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self._frozen("opacity", self._opacity, torch.sigmoid)

    @property
    def get_semantic_feature(self):
        return self._semantic_feature

    def rewrite_semantic_feature(self, x):
        self._semantic_feature = x

    @classmethod
    def from_tensors(cls, xyz, scaling_log, rotation, opacity_logit, features_dc=None, features_rest=None,
                     semantic_feature=None, sh_degree=3, active_sh_degree=None):
        m = cls(sh_degree)
        n = xyz.shape[0]
        dev = xyz.device
        m._xyz = nn.Parameter(xyz.contiguous().float(), requires_grad=False)
        m._scaling = nn.Parameter(scaling_log.contiguous().float(), requires_grad=False)
        m._rotation = nn.Parameter(rotation.contiguous().float(), requires_grad=False)
        m._opacity = nn.Parameter(opacity_logit.reshape(n, 1).contiguous().float(), requires_grad=False)
        if features_dc is None:
            features_dc = torch.zeros(n, 1, 3, device=dev)
        if features_rest is None:
            features_rest = torch.zeros(n, (sh_degree + 1) ** 2 - 1, 3, device=dev)
        m._features_dc = nn.Parameter(features_dc.contiguous().float(), requires_grad=False)
        m._features_rest = nn.Parameter(features_rest.contiguous().float(), requires_grad=False)
        if semantic_feature is not None:
            m._semantic_feature = nn.Parameter(semantic_feature.contiguous().float(), requires_grad=True)
        m.active_sh_degree = sh_degree if active_sh_degree is None else active_sh_degree
        return m

    # -- on-disk formats: scene/gaussian_model.py:63-113,240-318 (gags_amd/io_formats.py) ----------------------
    def save_ply(self, path):
        from . import io_formats
        io_formats.write_ply(path, self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling,
                             self._rotation, self._semantic_feature)

    def load_ply(self, path, device="cuda"):
        self.invalidate_activations()
        from . import io_formats
        t = {k: (None if v is None else torch.from_numpy(np.array(v)).to(device)) for k, v in
             io_formats.read_ply(path, self.max_sh_degree).items()}
        new = GaussianModel.from_tensors(t["xyz"], t["scaling"], t["rotation"], t["opacity"], t["features_dc"],
                                         t["features_rest"], t["semantic_feature"], sh_degree=self.max_sh_degree)
        self.__dict__.update(new.__dict__)
        self.active_sh_degree = self.max_sh_degree
        return self

    def capture(self):
        """The reference's 13-tuple (scene/gaussian_model.py:63-78); the densification statistics it carries are
        empty here (densification is dead code in the feature flow)."""
        dev = self._xyz.device
        n = self._xyz.shape[0]
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, getattr(self, "max_radii2D", torch.zeros(n, device=dev)),
                getattr(self, "xyz_gradient_accum", torch.zeros(n, 1, device=dev)),
                getattr(self, "denom", torch.zeros(n, 1, device=dev)),
                self.optimizer.state_dict() if self.optimizer is not None else {},
                getattr(self, "spatial_lr_scale", 1.0), self._semantic_feature)

    def restore(self, model_args, semantic_feature_lr=0.001, semantic_dim=16):
        """12-tuple (RGB field: features start from zeros, train.py:82-94) or 13-tuple (feature field: features and
        optimizer state are taken over), as scene/gaussian_model.py:80-113."""
        self.invalidate_activations()
        if len(model_args) not in (12, 13):
            raise ValueError("checkpoint tuple must have 12 or 13 entries")
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
         self._opacity, self.max_radii2D, xyz_gradient_accum, denom, opt_dict, self.spatial_lr_scale) = model_args[:12]
        self._semantic_feature = model_args[12] if len(model_args) == 13 else None
        self.training_setup(semantic_feature_lr, semantic_dim)
        if len(model_args) == 13 and opt_dict:
            self.optimizer.load_state_dict(opt_dict)
        self.xyz_gradient_accum, self.denom = xyz_gradient_accum, denom
        return self

    def training_setup(self, semantic_feature_lr=0.001, semantic_dim=16):
        """Feature-only optimisation, as scene/gaussian_model.py:183-208."""
        n = self._xyz.shape[0]
        if self._semantic_feature is None or self._semantic_feature.shape[0] != n:
            self._semantic_feature = nn.Parameter(
                torch.zeros((n, semantic_dim), device=self._xyz.device).contiguous().requires_grad_(True))
        for p in (self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation):
            p.requires_grad_(False)
        # same constructor call as the reference; on the GPU the step is the single-pass HIP kernel
        groups = [{"params": [self._semantic_feature], "lr": semantic_feature_lr, "name": "semantic_feature"}]
        if self._semantic_feature.is_cuda:
            from .optim import FeatureAdam
            self.optimizer = FeatureAdam(groups, lr=0.0, eps=1e-15)
        else:  # host-side bookkeeping only (CPU unit tests): the stock optimizer the reference uses
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        return self.optimizer
