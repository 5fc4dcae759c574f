// Microbenchmark: does the 256 MiB Infinity Cache serve a producer -> consumer hand-over between two kernels?
// Kernel W writes X bytes (streaming float4 stores), kernel R reads them back (streaming float4 loads, sum).  Three patterns:
//   fresh   : every W/R pair uses a different region of a 16 GiB arena (nothing can be cached across pairs)
//   reuse   : every W/R pair uses the SAME region (dirty lines can be overwritten in the cache before they are evicted)
//   reuse2  : two regions alternating (double-buffered bands)
// The time of the pair W+R per byte says where the bytes went: HBM both ways (~X/5.5 TB/s each) or the on-die cache.
// Also: R alone on a region written long ago (HBM read reference) and W alone (HBM write reference).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void wk(float4 *dst, size_t n4, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = make_float4(v, v + 1.f, v + 2.f, (float)i);
}

__global__ __launch_bounds__(256) void rk(const float4 *src, size_t n4, float *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 u = src[i];
        s += u.x + u.y + u.z + u.w;
    }
    if (s == 123.456f) sink[0] = s;
}

int main()
{
    const size_t arena = (size_t)16 << 30;
    char *buf;
    float *sink;
    CK(hipMalloc(&buf, arena));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, arena));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    const size_t sizes_mb[] = {32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096};
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        const int pairs = (int)((arena / bytes) < 64 ? (arena / bytes) : 64);
        float ms[5];
        for (int pat = 0; pat < 5; ++pat) {
            // 0 fresh W+R, 1 reuse W+R, 2 reuse2 W+R, 3 R only over fresh regions (written in the memset / earlier: cold), 4 W only fresh
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                for (int p = 0; p < pairs; ++p) {
                    size_t off = 0;
                    if (pat == 0 || pat == 3 || pat == 4) off = (size_t)p * bytes;
                    if (pat == 2) off = (size_t)(p & 1) * bytes;
                    float4 *r = reinterpret_cast<float4 *>(buf + off);
                    if (pat != 3) wk<<<grid, 256>>>(r, n4, (float)p);
                    if (pat != 4) rk<<<grid, 256>>>(r, n4, sink);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[pat], e0, e1));
            }
            ms[pat] /= pairs;
        }
        const double gb = bytes / 1e9;
        printf("X = %5zu MiB: W+R fresh %.3f ms (%.2f TB/s per direction)  reuse %.3f ms (%.2f)  reuse2 %.3f ms (%.2f)  | R cold %.3f ms (%.2f TB/s)  W %.3f ms (%.2f TB/s)\n",
               mb, ms[0], 2 * gb / ms[0], ms[1], 2 * gb / ms[1], ms[2], 2 * gb / ms[2], ms[3], gb / ms[3], ms[4], gb / ms[4]);
    }
    return 0;
}
