// Which fp32 formulas reproduce torch's GaussianModel getters bit for bit on this stack?  (tools/micro/actprobe.py)
// Candidates for normalize(q) = q / max(|q|, 1e-12) differ in the order the four squares are added; exp / sigmoid go through
// the device libm the way a plain HIP kernel compiled with -ffp-contract=off gets them.
#include <hip/hip_runtime.h>
#include <cstdint>

template <int V>
__global__ void norm_kernel(int n, const float *__restrict__ q, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = q[4 * i], b = q[4 * i + 1], c = q[4 * i + 2], d = q[4 * i + 3];
    float s;
    if (V == 0) s = ((a * a + b * b) + c * c) + d * d;                 // sequential, separate roundings
    else if (V == 1) s = (a * a + c * c) + (b * b + d * d);             // shuffle tree, offset 2 then 1
    else if (V == 2) s = (a * a + b * b) + (c * c + d * d);             // pairwise
    else if (V == 3) s = fmaf(d, d, fmaf(c, c, fmaf(b, b, a * a)));     // sequential, contracted
    else if (V == 4) s = fmaf(b, b, a * a) + fmaf(d, d, c * c);
    else s = fmaf(c, c, a * a) + fmaf(d, d, b * b);
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
    out[4 * i] = a / nrm; out[4 * i + 1] = b / nrm; out[4 * i + 2] = c / nrm; out[4 * i + 3] = d / nrm;
}

__global__ void exp_kernel(int n, const float *__restrict__ x, float m, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = expf(x[i]) * m;
}

template <int V>
__global__ void sigmoid_kernel(int n, const float *__restrict__ x, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (V == 0) out[i] = 1.0f / (1.0f + expf(-x[i]));
    else out[i] = 1.0f / (1.0f + __expf(-x[i]));
}

extern "C" {
void probe_norm(int v, int n, const float *q, float *out, void *st)
{
    dim3 g((n + 255) / 256), b(256);
    hipStream_t s = (hipStream_t)st;
    switch (v) {
    case 0: hipLaunchKernelGGL(norm_kernel<0>, g, b, 0, s, n, q, out); break;
    case 1: hipLaunchKernelGGL(norm_kernel<1>, g, b, 0, s, n, q, out); break;
    case 2: hipLaunchKernelGGL(norm_kernel<2>, g, b, 0, s, n, q, out); break;
    case 3: hipLaunchKernelGGL(norm_kernel<3>, g, b, 0, s, n, q, out); break;
    case 4: hipLaunchKernelGGL(norm_kernel<4>, g, b, 0, s, n, q, out); break;
    default: hipLaunchKernelGGL(norm_kernel<5>, g, b, 0, s, n, q, out); break;
    }
}
void probe_exp(int n, const float *x, float m, float *out, void *st)
{
    hipLaunchKernelGGL(exp_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, n, x, m, out);
}
void probe_sigmoid(int v, int n, const float *x, float *out, void *st)
{
    if (v == 0) hipLaunchKernelGGL(sigmoid_kernel<0>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, n, x, out);
    else hipLaunchKernelGGL(sigmoid_kernel<1>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, n, x, out);
}
}
