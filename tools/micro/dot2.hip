// Microbenchmark: the harness loss <x, y> over two 4.25 GB fp32 streams (optim.hip: dot_partial_kernel, 5.7 TB/s) -- which
// traversal gets closest to the 6.4 TB/s a single-stream read reaches (mall_pc.hip)?
//   0  grid-stride, 4 + 4 float4 in flight per thread (the shipped kernel)
//   1  the same with 8 + 8 in flight
//   2  blocked: workgroup b owns one contiguous chunk of both streams
//   3  grid-stride, y's traversal shifted by half the grid (x[i] y[i] are still paired: the shift is in WHICH i a wave takes
//      for its y loads first -- no: pairs must meet in one thread; instead x and y come from allocations whose bases differ
//      by 1 MiB + 4 KiB (bank / channel bits of paired requests differ)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int Q, bool BLOCKED>
__global__ __launch_bounds__(256) void dotk(int64_t n4, const float4 *__restrict__ x, const float4 *__restrict__ y, double *partial)
{
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t i, stride, end;
    if (BLOCKED) {
        const int64_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
        i = (int64_t)blockIdx.x * chunk + threadIdx.x;
        end = min(n4, (int64_t)(blockIdx.x + 1) * chunk);
        stride = 256;
    } else {
        i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        end = n4;
        stride = (int64_t)gridDim.x * 256;
    }
    for (; i + (Q - 1) * stride < end; i += Q * stride) {
        float4 u[Q], v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { u[q] = x[i + q * stride]; v[q] = y[i + q * stride]; }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            a0 = fmaf(u[q].x, v[q].x, a0); a1 = fmaf(u[q].y, v[q].y, a1); a2 = fmaf(u[q].z, v[q].z, a2); a3 = fmaf(u[q].w, v[q].w, a3);
        }
    }
    for (; i < end; i += stride) {
        const float4 u = x[i], v = y[i];
        a0 = fmaf(u.x, v.x, a0); a1 = fmaf(u.y, v.y, a1); a2 = fmaf(u.z, v.z, a2); a3 = fmaf(u.w, v.w, a3);
    }
    double s = ((double)a0 + a1) + ((double)a2 + a3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&partial[blockIdx.x & 1023], s);
}

template <int Q, bool BLOCKED>
static void run(const char *name, int grid, int64_t n4, const float4 *x, const float4 *y, double *partial)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((dotk<Q, BLOCKED>), dim3(grid), dim3(256), 0, 0, n4, x, y, partial);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-52s grid %5d: %.3f ms  %.2f TB/s\n", name, grid, best, 2.0 * n4 * 16 / best * 1e-9);
}

int main()
{
    const int64_t n = (int64_t)1920 * 1080 * 512, n4 = n / 4;
    char *buf;
    double *partial;
    CK(hipMalloc(&buf, (size_t)n * 8 + (64 << 20)));
    CK(hipMalloc(&partial, 8192));
    CK(hipMemset(buf, 0, (size_t)n * 8 + (64 << 20)));
    CK(hipMemset(partial, 0, 8192));
    const float4 *x = (const float4 *)buf;
    const float4 *y0 = (const float4 *)(buf + (size_t)n * 4);                       // back to back (2 MiB-aligned like two tensors)
    const float4 *y1 = (const float4 *)(buf + (size_t)n * 4 + (1 << 20) + 4096);     // shifted
    for (int grid : {2048, 4096, 8192}) {
        run<4, false>("grid-stride, 4 + 4 in flight (shipped)", grid, n4, x, y0, partial);
        run<8, false>("grid-stride, 8 + 8 in flight", grid, n4, x, y0, partial);
        run<2, false>("grid-stride, 2 + 2 in flight", grid, n4, x, y0, partial);
        run<4, true>("blocked chunks, 4 + 4 in flight", grid, n4, x, y0, partial);
        run<4, false>("grid-stride, 4 + 4, y shifted by 1 MiB + 4 KiB", grid, n4, x, y1, partial);
    }
    return 0;
}
