// Microbenchmark: what a wave can do while its SIMD partner streams fp32 MFMAs.
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) issue back-to-back v_mfma_f32_32x32x2_f32 (4 independent
// accumulators, the rows kernels' burst), waves 4-7 (their SIMD partners) run one of several "filler" loops.  Reports
// the fillers' cycles per instruction with the partner idle and with the partner multiplying, and the MFMA wave's
// slowdown.  Build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o bin/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void k(int mode, int mfma_on, int iters, float *buf, long long *tout, int swap, int prio)
{
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    const bool mf = swap ? (wave >= 4) : (wave < 4);
    if (!mf && prio) __builtin_amdgcn_s_setprio(3);
    if (mf) {
        if (mfma_on) {
            f32x16 acc[4];
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            float a = lane * 1e-3f, b = lane * 2e-3f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 32; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
            float s = 0.f;
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 16; ++r) s += acc[j][r];
            if (s == 12345.f) buf[0] = s;
        }
    } else {
        float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
        if (mode == 0) {  // independent VALU adds (4 chains)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(1.0f));
                }
            }
        } else if (mode == 1) {  // LDS reads b128
            int addr = (lane * 16) & 0x7fff;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    float4 v;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((addr + u * 1024) & 0x7ff0));
                    if (u == 31) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x0 += v.x; }
                }
            }
        } else if (mode == 2) {  // global stores dwordx4 (1 KB per instruction), no waits in between
            float4 *dst = reinterpret_cast<float4 *>(buf) + ((size_t)blockIdx.x * 4 + (wave & 3)) * 64 * 128 + lane;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 32; ++u) dst[(u & 127) * 64] = make_float4(x0, x1, x2, x3);
            }
        } else if (mode == 3) {  // SALU
            int s = blockIdx.x;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 128; ++u) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
            }
            x0 += s;
        }
        x0 += lds[(wave * 64 + lane) & 8191] * 0.f;
        if (x0 + x1 + x2 + x3 == 12345.f) buf[1] = x0;
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) tout[blockIdx.x * 8 + (mf ? (wave & 3) : 4 + (wave & 3))] = t1 - t0;
}

int main()
{
    float *buf; long long *tout;
    const int blocks = 256;
    hipMalloc(&buf, (size_t)blocks * 4 * 64 * 128 * 16 + 64);
    hipMalloc(&tout, blocks * 8 * 8);
    long long *h = (long long *)malloc(blocks * 8 * 8);
    const char *names[] = {"v_add_f32 x4 indep", "ds_read_b128", "global_store_dwordx4", "s_add_u32"};
    const int per_iter[] = {128, 32, 32, 128};
    const int iters = 200;
    for (int cfg = 0; cfg < 4; ++cfg)
    for (int mode = 0; mode < 4; ++mode)
        for (int on = (cfg ? 1 : 0); on < 2; ++on) {
            const int swap = cfg & 1, prio = cfg >> 1;
            if (mode == 0 && on == 1) printf("-- MFMA on the %s waves of the workgroup, filler s_setprio %d\n", swap ? "younger (4-7)" : "older (0-3)", prio ? 3 : 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, on, iters, buf, tout, swap, prio);
            hipDeviceSynchronize();
            hipMemcpy(h, tout, blocks * 8 * 8, hipMemcpyDeviceToHost);
            double tm = 0, tf = 0;
            for (int b = 0; b < blocks; ++b)
                for (int w = 0; w < 8; ++w) (w < 4 ? tm : tf) += h[b * 8 + w];
            tm /= blocks * 4; tf /= blocks * 4;
            printf("%-22s partner MFMA %s: filler %.1f cycles/instr", names[mode], on ? "on " : "off", tf / ((double)iters * per_iter[mode]));
            if (on) printf(", MFMA wave %.1f cycles/MFMA", tm / ((double)iters * 128));
            printf("\n");
        }
    return 0;
}
