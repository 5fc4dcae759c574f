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

}  // namespace

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
