// Microbenchmark: how fast does the dispatcher refill wave slots?  N single-wave workgroups, each spinning for
// `spin` shader cycles; VGPR footprint 256 (2 waves / SIMD) or small (8 waves / SIMD).  Reports the achieved
// fraction of the ideal time (N * spin / resident slots).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// spin time varies per workgroup when `spread` != 0: spin * (1 + spread * u), u in [0, 1) hashed from the id
template <int BIG, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void spin_kernel(long long spin, long long *out, int spread = 0)
{
    const long long t0 = __builtin_readcyclecounter();
    if (spread) {
        unsigned h = blockIdx.x * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        spin += spin * spread * (long long)(h & 1023) / 1024;
    }
    if (BIG) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = t0;
}

template <int BIG, int WAVES>
static void run(int n, long long spin)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((spin_kernel<BIG, WAVES>), dim3(1024), dim3(64 * WAVES), 0, 0, 100, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((spin_kernel<BIG, WAVES>), dim3(n / WAVES), dim3(64 * WAVES), 0, 0, spin, nullptr);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int slots = 1024 * (BIG ? 2 : 8);
    const double ideal_ms = (double)n * spin / 2.4e9 * 1e3 / slots;  // assuming 2.4 GHz
    printf("big=%d waves/wg=%d n=%d spin=%lld cyc (%.1f us): %.3f ms, ideal %.3f ms, launch rate %.1f waves/us\n", BIG, WAVES, n,
           spin, spin / 2400.0, ms, ideal_ms, n / (ms * 1e3));
}

// variable-duration waves: is the slot refill still perfect?  (ideal = sum of spins / slots)
static void run_var(int n, long long spin, int spread)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((spin_kernel<1, 1>), dim3(1024), dim3(64), 0, 0, 100, nullptr, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((spin_kernel<1, 1>), dim3(n), dim3(64), 0, 0, spin, nullptr, spread);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mean_spin = spin * (1.0 + spread * 0.5);
    const double ideal_ms = (double)n * mean_spin / 2.4e9 * 1e3 / 2048;
    printf("variable: n=%d spin=%.0f..%.0f us: %.3f ms, ideal %.3f ms -> slot utilisation %.2f\n", n, spin / 2400.0,
           spin * (1.0 + spread) / 2400.0, ms, ideal_ms, ideal_ms / ms);
}

int main()
{
    run_var(131072, 60000, 0);
    run_var(131072, 60000, 1);
    run_var(131072, 60000, 3);
    run_var(131072, 30000, 7);
    return 0;
    const int n = 131072;
    for (long long spin : {2400ll, 24000ll, 120000ll, 480000ll}) {
        run<1, 1>(n, spin);
        run<1, 4>(n, spin);
        run<0, 1>(n, spin);
        run<0, 4>(n, spin);
    }
    return 0;
}
