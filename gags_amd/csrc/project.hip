// K1 + K4: fused 3D->2D projection and tile counting, one lane per Gaussian.
// HBM-bound O(N) stage: reads 40 B (+ the shared 21-float camera block via scalar loads),
// writes 32 B per Gaussian.  Follows SURVEY.md Appendix A1-A6; the operation order is part
// of the numerical contract (bit-exact radii / tile counts against the oracle), so this
// translation unit must be built with -ffp-contract=off.
#include "common.h"
#include "raster_mfma_common.h"

namespace {

struct Cam {
    float R00, R01, R02, t0, R10, R11, R12, t1, R20, R21, R22, t2;
    float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ vm, const float *__restrict__ K)
{
    Cam c;
    c.R00 = vm[0]; c.R01 = vm[1]; c.R02 = vm[2]; c.t0 = vm[3];
    c.R10 = vm[4]; c.R11 = vm[5]; c.R12 = vm[6]; c.t1 = vm[7];
    c.R20 = vm[8]; c.R21 = vm[9]; c.R22 = vm[10]; c.t2 = vm[11];
    c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
    return c;
}

// The activations of scene/gaussian_model.py:116-139 as torch evaluates them on this stack, bit for bit
// (tools/micro/actprobe.py: 0 of 4 M values differ): get_scaling = exp(_scaling) (* scaling_modifier in render()),
// get_rotation = F.normalize(_rotation) = q / max(|q|, 1e-12) with |q|^2 summed pairwise, get_opacity = sigmoid(_opacity).
__device__ __forceinline__ float4 gags_act_rotation(float4 q)
{
    const float nrm = fmaxf(sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w)), 1e-12f);
    return make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);
}
__device__ __forceinline__ float gags_act_scaling(float s, float modifier) { return expf(s) * modifier; }
__device__ __forceinline__ float gags_act_opacity(float o) { return 1.0f / (1.0f + expf(-o)); }

// RAW: quats / scales are the stored parameters (_rotation un-normalised, _scaling in log space) and the kernel applies
// the getters itself (gags_project_fwd_raw); the activated opacity -- and, for a later backward, the activated quats and
// scales when asked for -- are written on the way.
template <bool RAW>
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int n, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmat, const float *__restrict__ Kmat,
    int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip,
    int tile_w, int tile_h,
    int32_t *__restrict__ radii, float *__restrict__ means2d, float *__restrict__ depths,
    float *__restrict__ conics, int32_t *__restrict__ tiles_per_gauss,
    const float *__restrict__ opacity_logits, float scaling_modifier, float *__restrict__ opacities_out,
    float *__restrict__ quats_out, float *__restrict__ scales_out, gags_mfma::GRec *__restrict__ grec_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 q = reinterpret_cast<const float4 *>(quats)[i];
    float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
    float opac = 0.f;
    if constexpr (RAW) {
        q = gags_act_rotation(q);
        s0 = gags_act_scaling(s0, scaling_modifier); s1 = gags_act_scaling(s1, scaling_modifier); s2 = gags_act_scaling(s2, scaling_modifier);
        opac = gags_act_opacity(opacity_logits[i]);
        opacities_out[i] = opac;
        if (quats_out) reinterpret_cast<float4 *>(quats_out)[i] = q;
        if (scales_out) { scales_out[3 * i] = s0; scales_out[3 * i + 1] = s1; scales_out[3 * i + 2] = s2; }
    }
    const Cam c = load_cam(viewmat, Kmat);
    const float fw = (float)width, fh = (float)height;
    const float tan_fovx = 0.5f * fw / c.fx;
    const float tan_fovy = 0.5f * fh / c.fy;
    const float lim_x_pos = (fw - c.cx) / c.fx + 0.3f * tan_fovx;
    const float lim_x_neg = c.cx / c.fx + 0.3f * tan_fovx;
    const float lim_y_pos = (fh - c.cy) / c.fy + 0.3f * tan_fovy;
    const float lim_y_neg = c.cy / c.fy + 0.3f * tan_fovy;

    int32_t o_rad = 0, o_tiles = 0;
    float o_mx = 0.f, o_my = 0.f, o_z = 0.f, o_a = 0.f, o_b = 0.f, o_c = 0.f;

    const float px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];
    const float x = ((c.R00 * px + c.R01 * py) + c.R02 * pz) + c.t0;
    const float y = ((c.R10 * px + c.R11 * py) + c.R12 * pz) + c.t1;
    const float z = ((c.R20 * px + c.R21 * py) + c.R22 * pz) + c.t2;
    if (!(z < near_plane || z > far_plane)) {
        float qw = q.x, qx = q.y, qy = q.z, qz = q.w;
        const float inv_norm = 1.0f / sqrtf(((qx * qx + qy * qy) + qz * qz) + qw * qw);
        qw *= inv_norm; qx *= inv_norm; qy *= inv_norm; qz *= inv_norm;
        const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
        const float wx = qw * qx, wy = qw * qy, wz = qw * qz;
        const float r00 = 1.f - 2.f * (y2 + z2), r01 = 2.f * (xy - wz), r02 = 2.f * (xz + wy);
        const float r10 = 2.f * (xy + wz), r11 = 1.f - 2.f * (x2 + z2), r12 = 2.f * (yz - wx);
        const float r20 = 2.f * (xz - wy), r21 = 2.f * (yz + wx), r22 = 1.f - 2.f * (x2 + y2);
        const float m00 = r00 * s0, m01 = r01 * s1, m02 = r02 * s2;
        const float m10 = r10 * s0, m11 = r11 * s1, m12 = r12 * s2;
        const float m20 = r20 * s0, m21 = r21 * s1, m22 = r22 * s2;
        const float c00 = (m00 * m00 + m01 * m01) + m02 * m02;
        const float c01 = (m00 * m10 + m01 * m11) + m02 * m12;
        const float c02 = (m00 * m20 + m01 * m21) + m02 * m22;
        const float c11 = (m10 * m10 + m11 * m11) + m12 * m12;
        const float c12 = (m10 * m20 + m11 * m21) + m12 * m22;
        const float c22 = (m20 * m20 + m21 * m21) + m22 * m22;
        const float a00 = (c.R00 * c00 + c.R01 * c01) + c.R02 * c02;
        const float a01 = (c.R00 * c01 + c.R01 * c11) + c.R02 * c12;
        const float a02 = (c.R00 * c02 + c.R01 * c12) + c.R02 * c22;
        const float a10 = (c.R10 * c00 + c.R11 * c01) + c.R12 * c02;
        const float a11 = (c.R10 * c01 + c.R11 * c11) + c.R12 * c12;
        const float a12 = (c.R10 * c02 + c.R11 * c12) + c.R12 * c22;
        const float a20 = (c.R20 * c00 + c.R21 * c01) + c.R22 * c02;
        const float a21 = (c.R20 * c01 + c.R21 * c11) + c.R22 * c12;
        const float a22 = (c.R20 * c02 + c.R21 * c12) + c.R22 * c22;
        const float v00 = (a00 * c.R00 + a01 * c.R01) + a02 * c.R02;
        const float v01 = (a00 * c.R10 + a01 * c.R11) + a02 * c.R12;
        const float v02 = (a00 * c.R20 + a01 * c.R21) + a02 * c.R22;
        const float v11 = (a10 * c.R10 + a11 * c.R11) + a12 * c.R12;
        const float v12 = (a10 * c.R20 + a11 * c.R21) + a12 * c.R22;
        const float v22 = (a20 * c.R20 + a21 * c.R21) + a22 * c.R22;

        const float rz = 1.f / z;
        const float rz2 = rz * rz;
        const float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
        const float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
        const float j00 = c.fx * rz, j02 = -c.fx * tx * rz2;
        const float j11 = c.fy * rz, j12 = -c.fy * ty * rz2;
        const float b00 = j00 * v00 + j02 * v02;
        const float b01 = j00 * v01 + j02 * v12;
        const float b02 = j00 * v02 + j02 * v22;
        const float b11 = j11 * v11 + j12 * v12;
        const float b12 = j11 * v12 + j12 * v22;
        float s00 = b00 * j00 + b02 * j02;
        const float s01 = b01 * j11 + b02 * j12;
        float s11 = b11 * j11 + b12 * j12;
        const float m2x = c.fx * x * rz + c.cx;
        const float m2y = c.fy * y * rz + c.cy;

        s00 += eps2d; s11 += eps2d;
        const float det = s00 * s11 - s01 * s01;
        if (det > 0.f) {
            const float inv_det = 1.f / det;
            const float hb = 0.5f * (s00 + s11);
            const float v1 = hb + sqrtf(fmaxf(0.01f, hb * hb - det));
            const float radius = ceilf(3.f * sqrtf(v1));
            const bool off = (radius <= radius_clip) || (m2x + radius <= 0.f) || (m2x - radius >= fw) ||
                             (m2y + radius <= 0.f) || (m2y - radius >= fh);
            if (!off) {
                o_rad = (int32_t)radius;
                o_mx = m2x; o_my = m2y; o_z = z;
                o_a = s11 * inv_det; o_b = -s01 * inv_det; o_c = s00 * inv_det;
                int x0, x1, y0, y1;
                gags_tile_aabb(m2x, m2y, o_rad, tile_w, tile_h, x0, x1, y0, y1);
                o_tiles = (y1 - y0) * (x1 - x0);
            }
        }
    }
    radii[i] = o_rad;
    reinterpret_cast<float2 *>(means2d)[i] = make_float2(o_mx, o_my);
    depths[i] = o_z;
    conics[3 * i] = o_a; conics[3 * i + 1] = o_b; conics[3 * i + 2] = o_c;
    tiles_per_gauss[i] = o_tiles;
    // K8b on the way (the activated opacity is at hand here): the per-Gaussian record of the matrix-core raster kernels
    // (gags_pack_isects with packed = NULL writes the same bytes from the arrays above)
    if constexpr (RAW) {
        if (grec_out && o_rad > 0) grec_out[i] = gags_mfma::make_grec_from(o_mx, o_my, o_a, o_b, o_c, opac);
    }
}

}  // namespace

extern "C" int gags_project_fwd(int n, const float *means, const float *quats, const float *scales,
                                const float *viewmat, const float *K, int width, int height,
                                float eps2d, float near_plane, float far_plane, float radius_clip,
                                int32_t *radii, float *means2d, float *depths, float *conics,
                                int32_t *tiles_per_gauss, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || width <= 0 || height <= 0) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!means || !quats || !scales || !viewmat || !K || !radii || !means2d || !depths || !conics ||
        !tiles_per_gauss)
        return GAGS_EINVAL;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    hipLaunchKernelGGL(project_fwd_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, means,
                       quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip,
                       tile_w, tile_h, radii, means2d, depths, conics, tiles_per_gauss, nullptr, 1.0f, nullptr, nullptr, nullptr, nullptr);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_project_fwd_raw(int n, const float *means, const float *rotation, const float *scaling_log,
                                    const float *opacity_logit, float scaling_modifier, const float *viewmat, const float *K,
                                    int width, int height, float eps2d, float near_plane, float far_plane,
                                    float radius_clip, int32_t *radii, float *means2d, float *depths, float *conics,
                                    int32_t *tiles_per_gauss, float *opacities, float *quats_act, float *scales_act,
                                    void *grec, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || width <= 0 || height <= 0) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!means || !rotation || !scaling_log || !opacity_logit || !viewmat || !K || !radii || !means2d || !depths ||
        !conics || !tiles_per_gauss || !opacities)
        return GAGS_EINVAL;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    hipLaunchKernelGGL(project_fwd_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, means,
                       rotation, scaling_log, viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip,
                       tile_w, tile_h, radii, means2d, depths, conics, tiles_per_gauss, opacity_logit, scaling_modifier,
                       opacities, quats_act, scales_act, reinterpret_cast<gags_mfma::GRec *>(grec));
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

namespace {

// K2: projection backward, one lane per Gaussian (same operation order as the forward).
// RAW (gags_project_bwd_raw): quats / scales are the stored parameters; the getters are re-applied (bit for bit the
// forward's), and the gradients are taken one step further, to the stored parameters: d/d _scaling = v_s * exp(_scaling) *
// modifier, d/d _rotation through q / max(|q|, 1e-12), d/d _opacity = v_o * o (1 - o) for EVERY Gaussian (the opacity's
// gradient comes from the rasterizer, not from the projection, so it is not gated by radii).
template <bool RAW>
__global__ __launch_bounds__(256) void project_bwd_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmat, const float *__restrict__ Kmat, int width, int height, float eps2d,
    const int32_t *__restrict__ radii, const float *__restrict__ v_means2d, const float *__restrict__ v_depths,
    const float *__restrict__ v_conics, float *__restrict__ v_means, float *__restrict__ v_quats,
    float *__restrict__ v_scales, const float *__restrict__ opacity_logits, float scaling_modifier,
    const float *__restrict__ v_opacities, float *__restrict__ v_opacity_logits)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if constexpr (RAW) {
        if (v_opacity_logits) {
            const float o = gags_act_opacity(opacity_logits[i]);
            v_opacity_logits[i] = v_opacities ? (v_opacities[i] * (1.0f - o)) * o : 0.f;  // (torch: grad * (1 - y) * y)
        }
    }
    _Pragma("unroll") for (int k = 0; k < 3; ++k) { v_means[3 * i + k] = 0.f; v_scales[3 * i + k] = 0.f; }
    _Pragma("unroll") for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = 0.f;
    if (radii[i] <= 0) return;
    const float Rc[3][3] = {{viewmat[0], viewmat[1], viewmat[2]},
                            {viewmat[4], viewmat[5], viewmat[6]},
                            {viewmat[8], viewmat[9], viewmat[10]}};
    const float tc[3] = {viewmat[3], viewmat[7], viewmat[11]};
    const float fx = Kmat[0], cx = Kmat[2], fy = Kmat[4], cy = Kmat[5];
    const float fw = (float)width, fh = (float)height;
    const float tan_fovx = 0.5f * fw / fx, tan_fovy = 0.5f * fh / fy;
    const float lim_x_pos = (fw - cx) / fx + 0.3f * tan_fovx, lim_x_neg = cx / fx + 0.3f * tan_fovx;
    const float lim_y_pos = (fh - cy) / fy + 0.3f * tan_fovy, lim_y_neg = cy / fy + 0.3f * tan_fovy;
        /* ---- recompute forward intermediates ---- */
        const float mu[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        float p[3];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) p[r] = ((Rc[r][0] * mu[0] + Rc[r][1] * mu[1]) + Rc[r][2] * mu[2]) + tc[r];
        const float x = p[0], y = p[1], z = p[2];
        float q0 = quats[4 * i], q1 = quats[4 * i + 1], q2 = quats[4 * i + 2], q3 = quats[4 * i + 3];
        float raw_nrm = 1.f;
        if constexpr (RAW) {
            raw_nrm = fmaxf(sqrtf((q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3)), 1e-12f);
            const float4 qa = gags_act_rotation(make_float4(q0, q1, q2, q3));
            q0 = qa.x; q1 = qa.y; q2 = qa.z; q3 = qa.w;
        }
        const float inv_norm = 1.0f / sqrtf(((q1 * q1 + q2 * q2) + q3 * q3) + q0 * q0);
        const float qw = q0 * inv_norm, qx = q1 * inv_norm, qy = q2 * inv_norm, qz = q3 * inv_norm;
        float R[3][3];
        R[0][0] = 1.f - 2.f * (qy * qy + qz * qz); R[0][1] = 2.f * (qx * qy - qw * qz); R[0][2] = 2.f * (qx * qz + qw * qy);
        R[1][0] = 2.f * (qx * qy + qw * qz); R[1][1] = 1.f - 2.f * (qx * qx + qz * qz); R[1][2] = 2.f * (qy * qz - qw * qx);
        R[2][0] = 2.f * (qx * qz - qw * qy); R[2][1] = 2.f * (qy * qz + qw * qx); R[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
        float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        if constexpr (RAW) { _Pragma("unroll") for (int k = 0; k < 3; ++k) s[k] = gags_act_scaling(s[k], scaling_modifier); }
        float M[3][3], S3[3][3], A[3][3], Sc[3][3];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c) M[r][c] = R[r][c] * s[c];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c)
            S3[r][c] = (M[r][0] * M[c][0] + M[r][1] * M[c][1]) + M[r][2] * M[c][2];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c)
            A[r][c] = (Rc[r][0] * S3[0][c] + Rc[r][1] * S3[1][c]) + Rc[r][2] * S3[2][c];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c)
            Sc[r][c] = (A[r][0] * Rc[c][0] + A[r][1] * Rc[c][1]) + A[r][2] * Rc[c][2];
        const float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        const float xr = x * rz, yr = y * rz;
        const int x_in = (xr <= lim_x_pos) && (xr >= -lim_x_neg);
        const int y_in = (yr <= lim_y_pos) && (yr >= -lim_y_neg);
        const float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
        const float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
        const float J[2][3] = {{fx * rz, 0.f, -fx * tx * rz2}, {0.f, fy * rz, -fy * ty * rz2}};
        float B[2][3];
        _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c)
            B[r][c] = (J[r][0] * Sc[0][c] + J[r][1] * Sc[1][c]) + J[r][2] * Sc[2][c];
        const float s00 = ((B[0][0] * J[0][0] + B[0][1] * J[0][1]) + B[0][2] * J[0][2]) + eps2d;
        const float s01 = (B[0][0] * J[1][0] + B[0][1] * J[1][1]) + B[0][2] * J[1][2];
        const float s11 = ((B[1][0] * J[1][0] + B[1][1] * J[1][1]) + B[1][2] * J[1][2]) + eps2d;
        const float det = s00 * s11 - s01 * s01;
        const float inv_det = 1.f / det;
        const float ca = s11 * inv_det, cb = -s01 * inv_det, cc = s00 * inv_det;
        /* ---- 1. conic -> cov2d:  G = -X Gx X,  X = [[ca,cb],[cb,cc]], Gx = [[va, vb/2],[vb/2, vc]] ---- */
        const float va = v_conics[3 * i], vb = 0.5f * v_conics[3 * i + 1], vc = v_conics[3 * i + 2];
        const float t00 = ca * va + cb * vb, t01 = ca * vb + cb * vc;
        const float t10 = cb * va + cc * vb, t11 = cb * vb + cc * vc;
        const float G00 = -(t00 * ca + t01 * cb), G01 = -(t00 * cb + t01 * cc);
        const float G10 = -(t10 * ca + t11 * cb), G11 = -(t10 * cb + t11 * cc);
        const float G[2][2] = {{G00, G01}, {G10, G11}};
        /* ---- 2. cov2d = J Sc J^T:  G_Sc = J^T G J ; G_J = G J Sc^T + G^T J Sc ---- */
        float GJ[2][3], GSc[3][3], vJ[2][3];
        _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c) GJ[r][c] = G[r][0] * J[0][c] + G[r][1] * J[1][c];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c) GSc[r][c] = J[0][r] * GJ[0][c] + J[1][r] * GJ[1][c];
        _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c) {
            const float gt0 = G[0][r], gt1 = G[1][r]; /* G^T row r */
            const float a1 = (GJ[r][0] * Sc[c][0] + GJ[r][1] * Sc[c][1]) + GJ[r][2] * Sc[c][2];
            const float gtj0 = gt0 * J[0][0] + gt1 * J[1][0], gtj1 = gt0 * J[0][1] + gt1 * J[1][1],
                        gtj2 = gt0 * J[0][2] + gt1 * J[1][2];
            const float a2 = (gtj0 * Sc[0][c] + gtj1 * Sc[1][c]) + gtj2 * Sc[2][c];
            vJ[r][c] = a1 + a2;
        }
        /* ---- 3. J and mean2d/depth -> camera-space point ---- */
        const float vm2x = v_means2d[2 * i], vm2y = v_means2d[2 * i + 1];
        float vp[3];
        vp[0] = fx * rz * vm2x;
        vp[1] = fy * rz * vm2y;
        vp[2] = -(fx * x * vm2x + fy * y * vm2y) * rz2;
        if (v_depths) vp[2] += v_depths[i];
        if (x_in) vp[0] += -fx * rz2 * vJ[0][2]; else vp[2] += -fx * rz3 * vJ[0][2] * tx;
        if (y_in) vp[1] += -fy * rz2 * vJ[1][2]; else vp[2] += -fy * rz3 * vJ[1][2] * ty;
        vp[2] += -fx * rz2 * vJ[0][0] - fy * rz2 * vJ[1][1] + 2.f * fx * tx * rz3 * vJ[0][2] + 2.f * fy * ty * rz3 * vJ[1][2];
        /* ---- 4. camera -> world ---- */
        _Pragma("unroll") for (int c = 0; c < 3; ++c) v_means[3 * i + c] = (Rc[0][c] * vp[0] + Rc[1][c] * vp[1]) + Rc[2][c] * vp[2];
        float T1[3][3], GS[3][3];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c)
            T1[r][c] = (Rc[0][r] * GSc[0][c] + Rc[1][r] * GSc[1][c]) + Rc[2][r] * GSc[2][c];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c)
            GS[r][c] = (T1[r][0] * Rc[0][c] + T1[r][1] * Rc[1][c]) + T1[r][2] * Rc[2][c];
        /* ---- 5. Sigma = M M^T: G_M = (G_S + G_S^T) M ---- */
        float GM[3][3];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int c = 0; c < 3; ++c)
            GM[r][c] = ((GS[r][0] + GS[0][r]) * M[0][c] + (GS[r][1] + GS[1][r]) * M[1][c]) + (GS[r][2] + GS[2][r]) * M[2][c];
        /* ---- 6. M = R diag(s) ---- */
        float GR[3][3];
        _Pragma("unroll") for (int c = 0; c < 3; ++c) {
            const float vs = (R[0][c] * GM[0][c] + R[1][c] * GM[1][c]) + R[2][c] * GM[2][c];
            v_scales[3 * i + c] = RAW ? vs * s[c] : vs;  // RAW: d/d log-scale (torch: grad * modifier, then * exp(_scaling))
            _Pragma("unroll") for (int r = 0; r < 3; ++r) GR[r][c] = GM[r][c] * s[c];
        }
        /* ---- 7. R(q^) -> q^ ---- */
        const float vqw = 2.f * (-qz * GR[0][1] + qy * GR[0][2] + qz * GR[1][0] - qx * GR[1][2] - qy * GR[2][0] + qx * GR[2][1]);
        const float vqx = 2.f * (qy * GR[0][1] + qz * GR[0][2] + qy * GR[1][0] - 2.f * qx * GR[1][1] - qw * GR[1][2] +
                                 qz * GR[2][0] + qw * GR[2][1] - 2.f * qx * GR[2][2]);
        const float vqy = 2.f * (-2.f * qy * GR[0][0] + qx * GR[0][1] + qw * GR[0][2] + qx * GR[1][0] + qz * GR[1][2] -
                                 qw * GR[2][0] + qz * GR[2][1] - 2.f * qy * GR[2][2]);
        const float vqz = 2.f * (-2.f * qz * GR[0][0] - qw * GR[0][1] + qx * GR[0][2] + qw * GR[1][0] - 2.f * qz * GR[1][1] +
                                 qy * GR[1][2] + qx * GR[2][0] + qy * GR[2][1]);
        /* ---- 8. normalisation q^ = q/|q| ---- */
        const float dotp = ((qw * vqw + qx * vqx) + qy * vqy) + qz * vqz;
        float g0 = (vqw - qw * dotp) * inv_norm, g1 = (vqx - qx * dotp) * inv_norm;
        float g2 = (vqy - qy * dotp) * inv_norm, g3 = (vqz - qz * dotp) * inv_norm;
        if constexpr (RAW) {
            // through the getter q_a = q / n, n = max(|q|, 1e-12):  v_q = (g - q_a <q_a, g>) / n  (q_a = (q0..q3) here)
            const float d2 = ((q0 * g0 + q1 * g1) + q2 * g2) + q3 * g3;
            g0 = (g0 - q0 * d2) / raw_nrm; g1 = (g1 - q1 * d2) / raw_nrm;
            g2 = (g2 - q2 * d2) / raw_nrm; g3 = (g3 - q3 * d2) / raw_nrm;
        }
        v_quats[4 * i] = g0; v_quats[4 * i + 1] = g1; v_quats[4 * i + 2] = g2; v_quats[4 * i + 3] = g3;
}

}  // namespace

extern "C" int gags_project_bwd(int n, const float *means, const float *quats, const float *scales,
                                const float *viewmat, const float *K, int width, int height, float eps2d,
                                const int32_t *radii, const float *conics, const float *v_means2d,
                                const float *v_depths, const float *v_conics, float *v_means, float *v_quats,
                                float *v_scales, void *stream)
{
    GAGS_CLEAR_ERR();
    (void)conics;  // recomputed in-kernel with the forward's operation order
    if (n < 0 || width <= 0 || height <= 0) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!means || !quats || !scales || !viewmat || !K || !radii || !v_means2d || !v_conics || !v_means ||
        !v_quats || !v_scales)
        return GAGS_EINVAL;
    hipLaunchKernelGGL(project_bwd_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, means, quats,
                       scales, viewmat, K, width, height, eps2d, radii, v_means2d, v_depths, v_conics, v_means,
                       v_quats, v_scales, nullptr, 1.0f, nullptr, nullptr);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_project_bwd_raw(int n, const float *means, const float *rotation, const float *scaling_log,
                                    const float *opacity_logit, float scaling_modifier, const float *viewmat, const float *K,
                                    int width, int height, float eps2d, const int32_t *radii, const float *v_means2d,
                                    const float *v_depths, const float *v_conics, const float *v_opacities, float *v_means,
                                    float *v_rotation, float *v_scaling_log, float *v_opacity_logit, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || width <= 0 || height <= 0) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!means || !rotation || !scaling_log || !viewmat || !K || !radii || !v_means2d || !v_conics || !v_means ||
        !v_rotation || !v_scaling_log || (v_opacity_logit && !opacity_logit))
        return GAGS_EINVAL;
    hipLaunchKernelGGL(project_bwd_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, means, rotation,
                       scaling_log, viewmat, K, width, height, eps2d, radii, v_means2d, v_depths, v_conics, v_means,
                       v_rotation, v_scaling_log, opacity_logit, scaling_modifier, v_opacities, v_opacity_logit);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
