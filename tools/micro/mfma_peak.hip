// Microbenchmark: achievable fp32 MFMA rate on this box (register-only v_mfma_f32_32x32x2_f32 chains).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; prints TFLOP/s for 1..4 waves / SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(64) void mfma_loop(int iters, float *out)
{
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.f) out[0] = s;
}

template <int NACC>
static void run(int waves_per_simd)
{
    float *out;
    hipMalloc(&out, 4);
    const int iters = 4000;
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(64), 0, 0, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(64), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * iters * 8 * NACC * 4096.0;
    printf("nacc=%d waves/simd=%d: %.3f ms  %.1f TFLOP/s\n", NACC, waves_per_simd, ms, flop / ms * 1e-9);
    hipFree(out);
}

// sustained: back-to-back launches for ~0.5 s; prints the rate of every 10th launch (clock / power throttling shows as decay)
static void sustained()
{
    float *out;
    (void)hipMalloc(&out, 4);
    const int iters = 4000, blocks = 256 * 4 * 2, n = 120;
    hipEvent_t ev[n + 1];
    for (int i = 0; i <= n; ++i) (void)hipEventCreate(&ev[i]);
    (void)hipEventRecord(ev[0]);
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(64), 0, 0, iters, out);
        (void)hipEventRecord(ev[i + 1]);
    }
    (void)hipDeviceSynchronize();
    const double flop = (double)blocks * iters * 8 * 4 * 4096.0;
    for (int i = 0; i < n; i += 10) {
        float ms;
        (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
        float t;
        (void)hipEventElapsedTime(&t, ev[0], ev[i + 1]);
        printf("t=%7.1f ms  launch %3d: %.3f ms  %.1f TFLOP/s\n", t, i, ms, flop / ms * 1e-9);
    }
}

int main()
{
    for (int w = 1; w <= 4; ++w) run<4>(w);
    sustained();
    return 0;
}
