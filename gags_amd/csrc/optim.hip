// R9: Adam step of the feature parameter [N, D] (scene/gaussian_model.py:192-208: torch.optim.Adam,
// eps = 1e-15, one group; stepped at train.py:221-223).  One pass over the four tensors: 16 B read + 12 B
// written per element, pure HBM streaming (the torch optimizer makes several passes over the same 3 GB
// tensors).  Same operation order as torch's single-tensor Adam; scalars formed in double on the host.
#include "common.h"
#include "gags_next.h"

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

// ---- harness helper: loss = <render, G>, the terminal loss of the synthetic step -------------
// Two 4.25 GB streams at C3: one read each, fp32 partial sums per thread, double across threads / blocks; fixed
// grid and order => reproducible.  (rocBLAS sdot runs the same reduction at ~4.8 TB/s; this one is bandwidth-bound.)
namespace {

constexpr int DOT_BLOCKS = 256 * 8;

__global__ __launch_bounds__(256) void dot_partial_kernel(int64_t n, const float *__restrict__ x,
                                                          const float *__restrict__ y, double *__restrict__ partial)
{
    __shared__ double sm[4];
    // every workgroup owns ONE contiguous chunk of both streams (round 6): 6.3 TB/s where the grid-stride traversal reached
    // 6.0 (tools/micro/dot2.hip) -- a workgroup's requests then walk DRAM pages in order instead of touching 2 x 4 pages
    // 8 MB apart per wave
    const int64_t n4 = n >> 2, stride = 256;
    const int64_t chunk = (n4 + DOT_BLOCKS - 1) / DOT_BLOCKS;
    const int64_t i_end = min(n4, (int64_t)(blockIdx.x + 1) * chunk);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t i = (int64_t)blockIdx.x * chunk + threadIdx.x;
    for (; i + 3 * stride < i_end; i += 4 * stride) {  // eight 16-byte requests per thread in flight
        float4 u[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u[q] = reinterpret_cast<const float4 *>(x)[i + q * stride];
            v[q] = reinterpret_cast<const float4 *>(y)[i + q * stride];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0 = fmaf(u[q].x, v[q].x, a0); a1 = fmaf(u[q].y, v[q].y, a1); a2 = fmaf(u[q].z, v[q].z, a2); a3 = fmaf(u[q].w, v[q].w, a3);
        }
    }
    for (; i < i_end; i += stride) {
        const float4 u = reinterpret_cast<const float4 *>(x)[i], v = reinterpret_cast<const float4 *>(y)[i];
        a0 = fmaf(u.x, v.x, a0); a1 = fmaf(u.y, v.y, a1); a2 = fmaf(u.z, v.z, a2); a3 = fmaf(u.w, v.w, a3);
    }
    const int64_t tail = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tail < n) a0 = fmaf(x[tail], y[tail], a0);
    double s = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void dot_final_kernel(const double *__restrict__ partial, float *__restrict__ out)
{
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < DOT_BLOCKS; i += 256) s += partial[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
}

}  // namespace

extern "C" int64_t gags_dot_scratch_bytes(void) { return (int64_t)DOT_BLOCKS * sizeof(double); }

extern "C" int gags_dot_f32(int64_t numel, const float *x, const float *y, float *out, void *scratch,
                            int64_t scratch_bytes, void *stream)
{
    if (numel < 0 || !out || !scratch || scratch_bytes < gags_dot_scratch_bytes()) return GAGS_EINVAL;
    if (numel > 0 && (!x || !y)) return GAGS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return GAGS_EINVAL;
    GAGS_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, numel, x, y, (double *)scratch);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, (const double *)scratch, out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// ---- the f16 decoder tier's gradient scale (gags_amd/decoders.py::_pow2_scale) ----------------------------------------------
// out[0] = S = 2^floor(target_log2 - log2(amax[0] / div)) (exponent clamped to +-100; 1 when amax is 0 or not finite),
// out[1] = 1 / S: one launch of one thread instead of a dozen element-wise launches on a scalar, no host readback.
namespace {
__global__ void pow2_scale_kernel(const float *__restrict__ amax, float div, float target_log2, float *__restrict__ out)
{
    const float a = amax[0] / div;
    float s = 1.0f;
    if (isfinite(a) && a > 0.f) {
        const float e = fminf(fmaxf(floorf(target_log2 - log2f(a)), -100.f), 100.f);
        s = ldexpf(1.0f, (int)e);
    }
    out[0] = s;
    out[1] = 1.0f / s;
}
}  // namespace

extern "C" int gags_pow2_scale(const float *amax, float div, float target_log2, float *out, void *stream)
{
    if (!amax || !out || !(div > 0.f)) return GAGS_EINVAL;
    GAGS_CLEAR_ERR();
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax, div, target_log2, out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
