"""Drop-in for /root/reference/gaussian_renderer/__init__.py: `render(...)`.

Same signature, same attribute reads on `viewpoint_camera` (FoVx, FoVy, image_width,
image_height, world_view_transform) and `pc` (get_xyz, get_opacity, get_scaling,
get_rotation, get_semantic_feature, get_features, active_sh_degree), same returned dict
(`render` [D',H,W] as a permuted view of [H,W,D'], `viewspace_points` [1,N,2],
`visibility_filter` [N] bool, `radii` [N] int32).  `pipe` is accepted and ignored, as in the
reference (its PipelineParams are never read on this path).  The only difference is what
executes underneath: gags_amd.rasterization instead of gsplat.rasterization.
"""
import math

import torch

from .rasterization import default_context, rasterization
from .scene import GaussianModel

# The reference re-uploads K with torch.tensor(..., device="cuda") on every call
# (gaussian_renderer/__init__.py:31-38).  K depends only on (FoVx, FoVy, W, H); the device copy stays resident in the
# caller's RasterContext instead of paying a pageable H2D copy per iteration.
def _intrinsics(viewpoint_camera, device, cache):
    key = (float(viewpoint_camera.FoVx), float(viewpoint_camera.FoVy), int(viewpoint_camera.image_width),
           int(viewpoint_camera.image_height), str(device))
    K = cache.get(key)
    if K is None:
        tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
        tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
        focal_length_x = viewpoint_camera.image_width / (2 * tanfovx)
        focal_length_y = viewpoint_camera.image_height / (2 * tanfovy)
        K = torch.tensor(
            [[focal_length_x, 0, viewpoint_camera.image_width / 2.0],
             [0, focal_length_y, viewpoint_camera.image_height / 2.0],
             [0, 0, 1]], device=device)
        if len(cache) > 4096:
            cache.clear()
        cache[key] = K
    return K


# R2 folded into R4 (the getters are "4 elementwise kernels/iter over N (fusable into projection)"): when `pc`
# stores its parameters the way scene/gaussian_model.py:48-61 does and activates them the way :36-42 does (exp, sigmoid,
# F.normalize), render() hands the STORED tensors to the projection kernel, which applies those getters itself -- bit for
# bit what torch computes (tools/micro/actprobe.py; tests/test_parity_gpu.py::test_raw_parameter_projection_*).  Any other
# `pc` goes through its getters, as in the reference.  False: always the getters.
RAW_PARAMS = True


def _stored_parameters(pc):
    """(rotation, scaling_log, opacity_logit) when `pc` provably activates its parameters the way
    scene/gaussian_model.py:36-42,116-139 does, else None (the getters are called, as in the reference).

    "Provably": `pc` is this package's GaussianModel with its getters NOT overridden (a subclass that clamps, masks or
    re-activates in get_scaling / get_opacity / get_rotation goes through its getters), or any other model that carries the
    reference's three activation attributes and they ARE torch.exp / torch.sigmoid / F.normalize (the reference's
    GaussianModel.setup_functions).  A duck-typed model without those attributes is never assumed to use them."""
    if not RAW_PARAMS:
        return None
    try:
        rot, scal, opac = pc._rotation, pc._scaling, pc._opacity
    except AttributeError:
        return None
    F = torch.nn.functional
    cls = type(pc)
    own = isinstance(pc, GaussianModel) and all(
        getattr(cls, name, None) is getattr(GaussianModel, name) for name in ("get_scaling", "get_rotation", "get_opacity"))
    if not own:
        try:
            if (pc.scaling_activation is not torch.exp or pc.opacity_activation is not torch.sigmoid
                    or pc.rotation_activation is not F.normalize):
                return None
        except AttributeError:
            return None
        if isinstance(pc, GaussianModel):  # (a subclass of ours with overridden getters: the attributes do not describe them)
            return None
    n = pc.get_xyz.shape[0]
    if not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in (rot, scal, opac)):
        return None
    if tuple(rot.shape) != (n, 4) or tuple(scal.shape) != (n, 3) or tuple(opac.shape) != (n, 1):
        return None
    return rot, scal, opac


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, feature_mode=True, scaling_modifier=1.0,
           override_color=None, render_mode="RGB", raster_flags=0, context=None):
    """Render the scene.  Background tensor (bg_color) must be on GPU!
    raster_flags / context are this package's additions to the reference's signature (kernel selection; the RasterContext
    holding this caller's hooks, capacities and caches -- None = the calling thread's default)."""
    rctx = context if context is not None else default_context()
    means3D = pc.get_xyz
    K = _intrinsics(viewpoint_camera, means3D.device, rctx.k_cache)
    stored = _stored_parameters(pc)
    if stored is not None:
        rotations, scales, opacity = stored
    else:
        opacity = pc.get_opacity
        scales = pc.get_scaling * scaling_modifier
        rotations = pc.get_rotation
    if feature_mode:
        colors = pc.get_semantic_feature  # [N, D]
        sh_degree = None
        bg_color = bg_color[0].repeat(colors.shape[-1])
    elif override_color is not None:
        colors = override_color  # [N, 3]
        sh_degree = None
    else:
        colors = pc.get_features  # [N, K, 3]
        sh_degree = pc.active_sh_degree

    viewmat = viewpoint_camera.world_view_transform.transpose(0, 1)  # [4, 4]
    render_colors, render_alphas, info = rasterization(
        means=means3D, quats=rotations, scales=scales, opacities=opacity.squeeze(-1), colors=colors,
        viewmats=viewmat[None], Ks=K[None], backgrounds=bg_color[None],
        width=int(viewpoint_camera.image_width), height=int(viewpoint_camera.image_height),
        packed=False, sh_degree=sh_degree, render_mode=render_mode, raster_flags=raster_flags,
        raw_params=stored is not None, scaling_modifier=float(scaling_modifier), context=rctx)

    # squeeze (not [0]): its backward is a view, [0]'s is a zero-fill + copy of the whole map
    rendered_image = render_colors.squeeze(0).permute(2, 0, 1)  # [1,H,W,D'] -> [D',H,W]
    radii = info["radii"].squeeze(0)  # [N,]
    try:
        info["means2d"].retain_grad()  # [1, N, 2]
    except Exception:
        pass
    return {"render": rendered_image,
            "viewspace_points": info["means2d"],
            "visibility_filter": radii > 0,
            "radii": radii,
            "alphas": render_alphas[0, ..., 0],
            "info": info}
