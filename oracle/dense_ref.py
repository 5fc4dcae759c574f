"""Dense float64 PyTorch-autograd restatement of the path.

TEST INFRASTRUCTURE ONLY (same rules as oracle.py).  It is an *independent* second
statement of the algorithm: all-pairs alpha, explicit tile-membership mask, cumulative
products instead of a sequential loop, true exp(), float64 throughout, gradients by
autograd.  It is used on CPU to validate gags_oracle.c (values and analytic gradients) at
small sizes.  The integer decisions that are defined in fp32 by the algorithm (radius,
tile AABB, depth order) are taken from the fp32 oracle so that both statements walk the
same Gaussian lists.
"""
import numpy as np
import torch

TILE = 16


def quat_to_rotmat(q):
    """wxyz, normalised here; convention of /root/reference/utils/general_utils.py:78-98."""
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def project(means, quats, scales, viewmat, K, width, height, eps2d=0.3):
    """A1-A4 (no culling; the caller masks with the oracle's radii).  Differentiable."""
    Rcw, t = viewmat[:3, :3], viewmat[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    R = quat_to_rotmat(quats)
    M = R * scales[:, None, :]
    cov = M @ M.transpose(1, 2)
    p = means @ Rcw.T + t
    cov_c = Rcw @ cov @ Rcw.T
    x, y, z = p.unbind(-1)
    tanx, tany = 0.5 * width / fx, 0.5 * height / fy
    lxp, lxn = (width - cx) / fx + 0.3 * tanx, cx / fx + 0.3 * tanx
    lyp, lyn = (height - cy) / fy + 0.3 * tany, cy / fy + 0.3 * tany
    rz = 1.0 / z
    tx = z * torch.minimum(lxp, torch.maximum(-lxn, x * rz))
    ty = z * torch.minimum(lyp, torch.maximum(-lyn, y * rz))
    zero = torch.zeros_like(z)
    J = torch.stack([fx * rz, zero, -fx * tx * rz * rz, zero, fy * rz, -fy * ty * rz * rz], -1).reshape(-1, 2, 3)
    cov2 = J @ cov_c @ J.transpose(1, 2)
    a = cov2[:, 0, 0] + eps2d
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + eps2d
    det = a * c - b * b
    conics = torch.stack([c / det, -b / det, a / det], -1)
    means2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], -1)
    return means2d, z, conics


def composite(means2d, conics, opacities, colors, backgrounds, width, height, radii, order, v_mask_fn=None):
    """A6-A9 dense: every pixel against every Gaussian (float64).

    radii: int array [N] from the fp32 oracle; order: Gaussian indices in global depth order
    (stable), from the fp32 oracle.  Returns colors [H,W,D], alphas [H,W], last (sorted pos)."""
    dev = means2d.device
    order_t = torch.as_tensor(order, dtype=torch.long, device=dev)
    m2, cn, op, col = means2d[order_t], conics[order_t], opacities[order_t], colors[order_t]
    rad = torch.as_tensor(np.asarray(radii)[order], dtype=torch.float64, device=dev)
    tile_w, tile_h = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    # tile AABB per Gaussian (fp32 semantics reproduced on the detached values)
    m2f = m2.detach().to(torch.float32)
    radf = rad.to(torch.float32)
    xmin = torch.clamp(torch.floor(m2f[:, 0] / TILE - radf / TILE), 0, tile_w)
    xmax = torch.clamp(torch.ceil(m2f[:, 0] / TILE + radf / TILE), 0, tile_w)
    ymin = torch.clamp(torch.floor(m2f[:, 1] / TILE - radf / TILE), 0, tile_h)
    ymax = torch.clamp(torch.ceil(m2f[:, 1] / TILE + radf / TILE), 0, tile_h)
    ii, jj = torch.meshgrid(torch.arange(height, device=dev), torch.arange(width, device=dev), indexing="ij")
    ii, jj = ii.reshape(-1), jj.reshape(-1)
    tyi, txi = (ii // TILE).to(torch.float32), (jj // TILE).to(torch.float32)
    member = ((txi[:, None] >= xmin[None]) & (txi[:, None] < xmax[None]) &
              (tyi[:, None] >= ymin[None]) & (tyi[:, None] < ymax[None]) & (rad[None] > 0))
    px, py = jj.to(torch.float64) + 0.5, ii.to(torch.float64) + 0.5
    dx = m2[None, :, 0] - px[:, None]
    dy = m2[None, :, 1] - py[:, None]
    sigma = 0.5 * (cn[None, :, 0] * dx * dx + cn[None, :, 2] * dy * dy) + cn[None, :, 1] * dx * dy
    alpha = torch.clamp(op[None] * torch.exp(-sigma), max=0.999)
    valid = member & (sigma >= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    next_T = torch.cumprod(one_m, dim=1)
    stop = valid & (next_T.detach() <= 1e-4)
    n = stop.shape[1]
    idx = torch.arange(n, device=dev)[None].expand_as(stop)
    first_stop = torch.where(stop, idx, torch.full_like(idx, n)).min(dim=1).values
    include = valid & (idx < first_stop[:, None])
    a_inc = torch.where(include, alpha, torch.zeros_like(alpha))
    one_m_inc = 1.0 - a_inc
    T_incl = torch.cumprod(one_m_inc, dim=1)
    T_before = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    w = a_inc * T_before
    T_final = T_incl[:, -1] if n > 0 else torch.ones(px.shape[0], dtype=torch.float64, device=dev)
    out = w @ col
    if backgrounds is not None:
        out = out + T_final[:, None] * backgrounds[None]
    alphas = 1.0 - T_final
    last_local = torch.where(include, idx, torch.full_like(idx, -1)).max(dim=1).values
    D = col.shape[1]
    return (out.reshape(height, width, D), alphas.reshape(height, width),
            last_local.reshape(height, width), include.sum().item())


def sh_colors(deg, coeffs, means, campos):
    """float64, differentiable: colour [N,3] = clamp_min(SH(normalize(means - campos)) + 0.5, 0) with the basis and signs
    of /root/reference/utils/sh_utils.py:57-112; coeffs [N,K,3] (this repository's layout).  Pinned against autograd of
    the reference's own eval_sh by tests/golden/shgrad_vectors.npz (values and d / d means, d / d coeffs)."""
    d = means - campos
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = coeffs
    r = 0.28209479177387814 * c[:, 0]
    if deg > 0:
        C1 = 0.4886025119029199
        r = r - C1 * y * c[:, 1] + C1 * z * c[:, 2] - C1 * x * c[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + 1.0925484305920792 * xy * c[:, 4] - 1.0925484305920792 * yz * c[:, 5]
             + 0.31539156525252005 * (2.0 * zz - xx - yy) * c[:, 6] - 1.0925484305920792 * xz * c[:, 7]
             + 0.5462742152960396 * (xx - yy) * c[:, 8])
    if deg > 2:
        r = (r - 0.5900435899266435 * y * (3 * xx - yy) * c[:, 9] + 2.890611442640554 * xy * z * c[:, 10]
             - 0.4570457994644658 * y * (4 * zz - xx - yy) * c[:, 11]
             + 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy) * c[:, 12]
             - 0.4570457994644658 * x * (4 * zz - xx - yy) * c[:, 13] + 1.445305721320277 * z * (xx - yy) * c[:, 14]
             - 0.5900435899266435 * x * (xx - 3 * yy) * c[:, 15])
    return torch.clamp_min(r + 0.5, 0.0)
