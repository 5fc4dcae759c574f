// R9: Adam step of the feature parameter [N, D] (scene/gaussian_model.py:192-208: torch.optim.Adam,
// eps = 1e-15, one group; stepped at train.py:221-223).  One pass over the four tensors: 16 B read + 12 B
// written per element, pure HBM streaming (the torch optimizer makes several passes over the same 3 GB
// tensors).  Same operation order as torch's single-tensor Adam; scalars formed in double on the host.
#include "common.h"

namespace {

struct AdamScalars {
    float w1, b2, w2, step_size, bc2_sqrt, eps;
};

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamScalars &s)
{
    const float mi = m + s.w1 * (g - m);
    const float vi = v * s.b2 + (s.w2 * g) * g;
    const float denom = sqrtf(vi) / s.bc2_sqrt + s.eps;  // IEEE sqrt / divide (no fast-math in this build)
    p = p - s.step_size * (mi / denom);
    m = mi;
    v = vi;
}

__global__ __launch_bounds__(256) void adam_step_kernel(int64_t n, float *__restrict__ p, const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, AdamScalars s)
{
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], mm = reinterpret_cast<float4 *>(m)[i],
               vv = reinterpret_cast<float4 *>(v)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        adam1(pp.x, gg.x, mm.x, vv.x, s);
        adam1(pp.y, gg.y, mm.y, vv.y, s);
        adam1(pp.z, gg.z, mm.z, vv.z, s);
        adam1(pp.w, gg.w, mm.w, vv.w, s);
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    const int64_t tail = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;  // n % 4 elements
    if (tail < n) adam1(p[tail], g[tail], m[tail], v[tail], s);
}

}  // namespace

extern "C" int gags_adam_step(int64_t numel, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                              double lr, double beta1, double beta2, double eps, int step, void *stream)
{
    if (numel < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return GAGS_EINVAL;
    if (numel == 0) return GAGS_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return GAGS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return GAGS_EINVAL;  // float4 accesses
    GAGS_CLEAR_ERR();
    AdamScalars s;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    s.w1 = (float)(1.0 - beta1); s.b2 = (float)beta2; s.w2 = (float)(1.0 - beta2);
    s.step_size = (float)(lr / bc1); s.bc2_sqrt = (float)sqrt(bc2); s.eps = (float)eps;
    const int64_t n4 = numel >> 2;
    const int64_t want = (n4 + 255) / 256;
    const int grid = (int)(want < 1 ? 1 : (want > 256 * 32 ? 256 * 32 : want));  // grid-stride: 32 blocks / CU
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, numel, param, grad, exp_avg,
                       exp_avg_sq, s);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
