// Microbenchmark: do global stores overlap with fp32 MFMA work on a SIMD?
//   mode 0: 128 MFMAs per tile                       (rows-kernel tile without memory)
//   mode 1: 16 x global_store_dwordx4 per tile       (1 KB per store and wave)
//   mode 2: 128 MFMAs, then the 16 stores            (the rows kernel's order)
//   mode 3: one store after every 8 MFMAs            (interleaved)
//   mode 4: 128 MFMAs, then 224 independent VALU ops, then the 16 stores   (VALU block after the MFMAs)
//   mode 6: like 2, but the accumulators are zeroed per tile and the stores write the accumulators (true dependency)
//   mode 7: like 6 plus 8 x global_load_dwordx4 of A operands per tile (consumed by the next tile's MFMAs)
//   mode 8: like 7 with the accumulators in AccVGPRs (inline-asm MFMA, "a" constraint)
//   mode 9: like 8 with the accumulators in arch VGPRs
//   mode 5: 128 MFMAs with 2 VALU ops in the shadow of each, then the 16 stores
// 2048 waves (2 per SIMD, 256 VGPRs each) or 1024 waves; `stream` = 1: every tile stores to fresh memory (HBM),
// 0: the wave re-writes its own 16 KB (L2).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64, 2) void k(int tiles, int stream, float *buf, size_t wave_stride, float *sink)
{
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    f32x4 data = {a, b, a, b};
    float *base = buf + (size_t)blockIdx.x * wave_stride + threadIdx.x * 4;
    float bj[4] = {b, b + 1.f, b + 2.f, b + 3.f};
    f32x4 Areg[8];
    for (int i = 0; i < 8; ++i) Areg[i] = data;
    for (int m = 0; m < tiles; ++m) {
        float *dst = base + (stream ? (size_t)m * 4096 : 0);
        if (MODE >= 6) {
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
        if (MODE == 7) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(dst + 4096);  // next tile's region: 8 x 1 KB
            f32x4 An[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) An[i] = src[i * 64];
#pragma unroll
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Areg[t >> 2][t & 3], bj[j], acc[j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) Areg[i] = An[i];
        }
        if (MODE == 9) {  // as 8, accumulators in arch VGPRs: separates "AccVGPR" from "asm-pinned order"
            const f32x4 *src = reinterpret_cast<const f32x4 *>(dst + 4096);
            f32x4 An[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) An[i] = src[i * 64];
#pragma unroll
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(Areg[t >> 2][t & 3]), "v"(bj[j]));
#pragma unroll
            for (int i = 0; i < 8; ++i) Areg[i] = An[i];
        }
        if (MODE == 8) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(dst + 4096);
            f32x4 An[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) An[i] = src[i * 64];
#pragma unroll
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(Areg[t >> 2][t & 3]), "v"(bj[j]));
#pragma unroll
            for (int i = 0; i < 8; ++i) Areg[i] = An[i];
        }
        if (MODE == 6) {
#pragma unroll
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bj[j], acc[j], 0, 0, 0);
        }
        if (MODE >= 6) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                f32x4 o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + r * 256), "v"(o) : "memory");
            }
            continue;
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (MODE != 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                    if (MODE == 5) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1" : "+v"(data.x), "+v"(data.y) : "v"(b), "v"(data.y));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (MODE == 3 && (t & 1)) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + (t >> 1) * 256), "v"(data) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 4) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 56; ++r)
                asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4"
                             : "+v"(data.x), "+v"(data.y), "+v"(data.z), "+v"(data.w) : "v"(b));
        }
        if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + r * 256), "v"(data) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.f) sink[0] = s;
}

// Big tiles, ONE wave per SIMD: 256 MFMAs (8 accumulator tiles = 256 channels) per 32-slot tile, the next tile's
// 8 KB of A operands requested before the burst and waited for after it, then 32 row stores (true dependency).
__global__ __launch_bounds__(64, 1) void kbig(int tiles, float *buf, size_t wave_stride, float *sink)
{
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    float bj[8];
    for (int j = 0; j < 8; ++j) bj[j] = b + j;
    f32x4 Areg[8];
    for (int i = 0; i < 8; ++i) Areg[i] = f32x4{a, b, a, b};
    float *base = buf + (size_t)blockIdx.x * wave_stride + threadIdx.x * 4;
    float s = 0.f;
    for (int m = 0; m < tiles; ++m) {
        float *dst = base + (size_t)m * 8192;  // 32 KB of rows per tile
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(dst + 8192);
        f32x4 An[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) An[i] = src[i * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 32; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(Areg[t >> 2][t & 3]), "v"(bj[j]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) Areg[i] = An[i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x4 o0 = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]}, o1 = {acc[4][r], acc[5][r], acc[6][r], acc[7][r]};
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + r * 512), "v"(o0) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + r * 512 + 256), "v"(o1) : "memory");
        }
    }
    if (s == 12345.f) sink[0] = s;
}

static void run_big(int tiles, float *buf, float *sink)
{
    const int waves = 1024;
    const size_t wave_stride = (size_t)(tiles + 1) * 8192;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kbig, dim3(waves), dim3(64), 0, 0, 2, buf, wave_stride, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kbig, dim3(waves), dim3(64), 0, 0, tiles, buf, wave_stride, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)waves * tiles * 256 * 4096.0, bytes = (double)waves * tiles * 32768.0;
    printf("big tiles, 1 wave/SIMD: %.3f ms  %.1f TFLOP/s  %.2f TB/s stored\n", ms, flop / ms * 1e-9, bytes / ms * 1e-9);
}

template <int MODE>
static void run(int waves, int tiles, int stream, float *buf, size_t wave_stride, float *sink)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, 2, stream, buf, wave_stride, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, tiles, stream, buf, wave_stride, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = MODE == 1 ? 0 : (double)waves * tiles * 128 * 4096.0;
    const double bytes = MODE == 0 ? 0 : (double)waves * tiles * 16384.0;
    printf("mode %d waves %d stream %d: %.3f ms  %.1f TFLOP/s  %.2f TB/s stored\n", MODE, waves, stream, ms, flop / ms * 1e-9,
           bytes / ms * 1e-9);
}

int main()
{
    const int tiles = 256;
    const size_t wave_stride = (size_t)tiles * 4096;  // floats: 16 KB per tile
    float *buf, *sink;
    (void)hipMalloc(&buf, 2112 * wave_stride * 4);  // + slack: every mode reads one tile past its last  // 8 GB
    (void)hipMalloc(&sink, 4);
    for (int stream = 1; stream < 2; ++stream)
        for (int waves : {2048}) {
            run<0>(waves, tiles, stream, buf, wave_stride, sink);
            run<1>(waves, tiles, stream, buf, wave_stride, sink);
            run<2>(waves, tiles, stream, buf, wave_stride, sink);
            run<3>(waves, tiles, stream, buf, wave_stride, sink);
            run<4>(waves, tiles, stream, buf, wave_stride, sink);
            run<5>(waves, tiles, stream, buf, wave_stride, sink);
            run<6>(waves, tiles, stream, buf, wave_stride, sink);
            run<7>(waves, tiles, stream, buf, wave_stride, sink);
            run<8>(waves, tiles, stream, buf, wave_stride, sink);
            run<9>(waves, tiles, stream, buf, wave_stride, sink);
        }
    run_big(256, buf, sink);
    return 0;
}
