// K3: spherical-harmonics colour fwd/bwd (feature_mode=False, no override colour:
// gaussian_renderer/__init__.py:51-53).  One lane per (Gaussian, channel); HBM-bound.
// Basis and sign convention: utils/sh_utils.py:57-112 (degree 0..3).
#include "common.h"

namespace {

__constant__ const float SH_C0 = 0.28209479177387814f;
__constant__ const float SH_C1 = 0.4886025119029199f;

__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float *b /*16*/)
{
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = 1.0925484305920792f * xy;
            b[5] = -1.0925484305920792f * yz;
            b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
            b[7] = -1.0925484305920792f * xz;
            b[8] = 0.5462742152960396f * (xx - yy);
            if (deg > 2) {
                b[9] = -0.5900435899266435f * y * (3.f * xx - yy);
                b[10] = 2.890611442640554f * xy * z;
                b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
                b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
                b[14] = 1.445305721320277f * z * (xx - yy);
                b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
            }
        }
    }
}

__device__ __forceinline__ void unit_dir(const float *means, const float *campos, int i, float &x, float &y, float &z)
{
    x = means[3 * i] - campos[0];
    y = means[3 * i + 1] - campos[1];
    z = means[3 * i + 2] - campos[2];
    const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
    x *= inorm; y *= inorm; z *= inorm;
}

__global__ __launch_bounds__(256) void sh_fwd_kernel(int n, int kc, int deg, const float *__restrict__ means,
                                                     const float *__restrict__ campos,
                                                     const float *__restrict__ coeffs,
                                                     const int32_t *__restrict__ radii, float *__restrict__ out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * 3) return;
    const int i = idx / 3, c = idx - 3 * i;
    if (radii && radii[i] <= 0) { out[idx] = 0.f; return; }
    float x, y, z;
    unit_dir(means, campos, i, x, y, z);
    float b[16];
    sh_basis(deg, x, y, z, b);
    const int nb = (deg + 1) * (deg + 1);
    const float *sh = coeffs + (size_t)i * kc * 3 + c;
    float r = 0.f;
    for (int k = 0; k < nb; ++k) r += b[k] * sh[k * 3];
    out[idx] = fmaxf(r + 0.5f, 0.f);
}

// v_coeffs[i,k,c] = basis_k(dir_i) * v_out[i,c] * [colour not clamped]; zero for k >= (deg+1)^2
__global__ __launch_bounds__(256) void sh_bwd_kernel(int n, int kc, int deg, const float *__restrict__ means,
                                                     const float *__restrict__ campos,
                                                     const int32_t *__restrict__ radii,
                                                     const float *__restrict__ colors_out,
                                                     const float *__restrict__ v_out, float *__restrict__ v_coeffs)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * 3) return;
    const int i = idx / 3, c = idx - 3 * i;
    float *vsh = v_coeffs + (size_t)i * kc * 3 + c;
    const bool live = !(radii && radii[i] <= 0) && colors_out[idx] > 0.f;
    if (!live) {
        for (int k = 0; k < kc; ++k) vsh[k * 3] = 0.f;
        return;
    }
    float x, y, z;
    unit_dir(means, campos, i, x, y, z);
    float b[16];
    sh_basis(deg, x, y, z, b);
    const int nb = (deg + 1) * (deg + 1);
    const float v = v_out[idx];
    for (int k = 0; k < kc; ++k) vsh[k * 3] = (k < nb) ? b[k] * v : 0.f;
}

// d colour / d means through the view direction (gsplat propagates it; reached by gaussian_renderer/__init__.py:51-53
// with feature_mode=False and trainable xyz, train.py:142 without --feature_mode).  One lane per Gaussian:
//   g = sum_c [colour_c not clamped] v_out[c] * sum_k sh[k, c] * d basis_k / d (x, y, z)      (polynomials of sh_basis)
//   v_means = (g - <g, n> n) / |mean - campos|                                                 (through the normalisation)
__global__ __launch_bounds__(256) void sh_bwd_dirs_kernel(int n, int kc, int deg, const float *__restrict__ means,
                                                          const float *__restrict__ campos,
                                                          const float *__restrict__ coeffs,
                                                          const int32_t *__restrict__ radii,
                                                          const float *__restrict__ colors_out,
                                                          const float *__restrict__ v_out, float *__restrict__ v_means)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const bool vis = !(radii && radii[i] <= 0);
    float dx = 0.f, dy = 0.f, dz = 0.f, inorm = 0.f;
    if (vis && deg > 0) {
        dx = means[3 * i] - campos[0];
        dy = means[3 * i + 1] - campos[1];
        dz = means[3 * i + 2] - campos[2];
        inorm = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
        const float x = dx * inorm, y = dy * inorm, z = dz * inorm;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        for (int c = 0; c < 3; ++c) {
            if (!(colors_out[3 * i + c] > 0.f)) continue;  // clamped at 0: no gradient
            const float v = v_out[3 * i + c];
            const float *sh = coeffs + (size_t)i * kc * 3 + c;
#define SH(k) sh[(k) * 3]
            float ax = -SH_C1 * SH(3), ay = -SH_C1 * SH(1), az = SH_C1 * SH(2);
            if (deg > 1) {
                const float c2a = 1.0925484305920792f, c2b = 0.31539156525252005f, c2c = 0.5462742152960396f;
                ax += c2a * y * SH(4) - 2.f * c2b * x * SH(6) - c2a * z * SH(7) + 2.f * c2c * x * SH(8);
                ay += c2a * x * SH(4) - c2a * z * SH(5) - 2.f * c2b * y * SH(6) - 2.f * c2c * y * SH(8);
                az += -c2a * y * SH(5) + 4.f * c2b * z * SH(6) - c2a * x * SH(7);
                if (deg > 2) {
                    const float c3a = 0.5900435899266435f, c3b = 2.890611442640554f, c3c = 0.4570457994644658f,
                                c3d = 0.3731763325901154f, c3e = 1.445305721320277f;
                    ax += -6.f * c3a * xy * SH(9) + c3b * yz * SH(10) + 2.f * c3c * xy * SH(11) - 6.f * c3d * xz * SH(12)
                          - c3c * (4.f * zz - 3.f * xx - yy) * SH(13) + 2.f * c3e * xz * SH(14) - 3.f * c3a * (xx - yy) * SH(15);
                    ay += -3.f * c3a * (xx - yy) * SH(9) + c3b * xz * SH(10) - c3c * (4.f * zz - xx - 3.f * yy) * SH(11)
                          - 6.f * c3d * yz * SH(12) + 2.f * c3c * xy * SH(13) - 2.f * c3e * yz * SH(14) + 6.f * c3a * xy * SH(15);
                    az += c3b * xy * SH(10) - 8.f * c3c * yz * SH(11) + c3d * (6.f * zz - 3.f * xx - 3.f * yy) * SH(12)
                          - 8.f * c3c * xz * SH(13) + c3e * (xx - yy) * SH(14);
                }
            }
#undef SH
            gx += v * ax; gy += v * ay; gz += v * az;
        }
        const float dot = (gx * x + gy * y) + gz * z;
        gx = (gx - dot * x) * inorm; gy = (gy - dot * y) * inorm; gz = (gz - dot * z) * inorm;
    }
    v_means[3 * i] = gx; v_means[3 * i + 1] = gy; v_means[3 * i + 2] = gz;
}

}  // namespace

extern "C" int gags_sh_bwd_dirs(int n, int kc, int degree, const float *means, const float *campos, const float *coeffs,
                                const int32_t *radii, const float *colors_out, const float *v_out, float *v_means,
                                void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || degree < 0 || degree > 3 || kc < (degree + 1) * (degree + 1)) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!means || !campos || !coeffs || !colors_out || !v_out || !v_means) return GAGS_EINVAL;
    hipLaunchKernelGGL(sh_bwd_dirs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, kc, degree, means,
                       campos, coeffs, radii, colors_out, v_out, v_means);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_sh_fwd(int n, int kc, int degree, const float *means, const float *campos, const float *coeffs,
                           const int32_t *radii, float *out, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || degree < 0 || degree > 3 || kc < (degree + 1) * (degree + 1)) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!means || !campos || !coeffs || !out) return GAGS_EINVAL;
    hipLaunchKernelGGL(sh_fwd_kernel, dim3((n * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, kc, degree,
                       means, campos, coeffs, radii, out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_sh_bwd(int n, int kc, int degree, const float *means, const float *campos, const int32_t *radii,
                           const float *colors_out, const float *v_out, float *v_coeffs, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || degree < 0 || degree > 3 || kc < (degree + 1) * (degree + 1)) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!means || !campos || !colors_out || !v_out || !v_coeffs) return GAGS_EINVAL;
    hipLaunchKernelGGL(sh_bwd_kernel, dim3((n * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, kc, degree,
                       means, campos, radii, colors_out, v_out, v_coeffs);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
