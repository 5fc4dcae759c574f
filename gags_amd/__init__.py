"""gags_amd -- MI355X-native drop-in for GAGS's feature-rasterization hot path.

Public surface (mirrors the reference's):
    gags_amd.gaussian_renderer.render(viewpoint_camera, pc, pipe, bg_color, feature_mode=True, ...)
    gags_amd.rasterization.rasterization(means, quats, scales, opacities, colors, viewmats, Ks, ...)
    gags_amd.scene.GaussianModel / Camera
All device work goes through the C-ABI library gags_amd/csrc/libgags_hip.so
(include/gags_raster.h); there is no CPU or PyTorch fallback -- a missing library raises.
"""
__version__ = "0.1.0"
