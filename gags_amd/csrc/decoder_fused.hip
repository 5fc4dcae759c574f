// CNN_decoder (models/networks.py:139-218) as TWO fused kernels in the fast bf16 mode: the nine 1x1
// convolutions of the forward -- and the nine input-gradient GEMMs of the backward -- run per 64-pixel tile with the
// activations resident in LDS; only what the other direction needs crosses HBM (forward: each layer's output once, for the
// ReLU masks and the weight gradients; backward: each layer's dz once, for the weight gradients).  Layer by layer
// (csrc/decoder.hip) every GEMM read its input from HBM and wrote its output back: 15 launches, 12 of 30 ms per iteration.
//
// Tile = 64 pixels, workgroup = 4 waves, two workgroups per CU.  A layer is  out^T[256 n x 64 p] = W[256 x K] act^T[K x 64]:
// the MFMA A operand is the weight matrix (row n, 16-byte rows straight from global memory / L2: 128 KB per layer, shared by
// every tile), the B operand the activation tile (row p of the LDS buffer), so a lane of the accumulator owns one pixel and
// runs of four consecutive channels -- the epilogue writes 8-byte pieces of pixel-major rows.  Wave w owns channels
// 64 w .. 64 w + 63.  Two LDS buffers ping-pong; the residual sums x1 + x2 and x3 + x4 are formed in place.
// Arithmetic is that of gags_decoder_layer (bf16 operands, fp32 accumulate in ascending k, bias + ReLU in fp32, one
// rounding to bf16): the results are BIT-IDENTICAL to the layer-by-layer path (tests/test_decoders_gpu.py).
#include "common.h"
#include "gags_next.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fbf16x2_t __attribute__((ext_vector_type(2)));
typedef float ff32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned fpack(float lo, float hi)
{
    const ff32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, fbf16x2_t));
}
__device__ __forceinline__ float flo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float fhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

constexpr int FT = 64;        // pixels per tile
constexpr int FH = 256;       // hidden width
constexpr int FLD = FH + 8;   // LDS row pitch in bf16 (528 B: 16-byte aligned rows, consecutive rows 4 banks apart)
constexpr int FPD = 4;        // weight fragments are requested this many K-steps ahead

typedef unsigned short (*Tile)[FLD];

// acc[i][j] (i: 32 channels 64 w + 32 i.., j: 32 pixels 32 j..) = sum over K = 16 ksteps of W[n][k] in[p][k]
__device__ __forceinline__ void layer_mma(f32x16 (&acc)[2][2], const unsigned short *__restrict__ W, int ldw, int n_base, int ksteps,
                                          Tile in, int lane, bool zero)
{
    if (zero) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const unsigned short *w0 = W + (size_t)(n_base + (lane & 31)) * ldw + 8 * (lane >> 5);
    const unsigned short *w1 = w0 + (size_t)32 * ldw;
    bf16x8 a0[FPD], a1[FPD];
#pragma unroll
    for (int q = 0; q < FPD; ++q) {
        const int ks = min(q, ksteps - 1);
        a0[q] = *reinterpret_cast<const bf16x8 *>(w0 + 16 * ks);
        a1[q] = *reinterpret_cast<const bf16x8 *>(w1 + 16 * ks);
    }
    for (int k0 = 0; k0 < ksteps; k0 += FPD) {
#pragma unroll
        for (int q = 0; q < FPD; ++q) {
            const int ks = k0 + q;
            const bf16x8 c0 = a0[q], c1 = a1[q];
            const int kn = min(ks + FPD, ksteps - 1);  // (past the end: a harmless re-read)
            a0[q] = *reinterpret_cast<const bf16x8 *>(w0 + 16 * kn);
            a1[q] = *reinterpret_cast<const bf16x8 *>(w1 + 16 * kn);
            if (ks < ksteps) {  // (uniform)
                const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(&in[(lane & 31)][16 * ks + 8 * (lane >> 5)]);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(&in[32 + (lane & 31)][16 * ks + 8 * (lane >> 5)]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch FPD steps ahead (the scheduler sinks loads to their use)
        }
    }
}

// accumulator element (i, j, 4 g + e) of lane (p = lane & 31, h = lane >> 5): pixel 32 j + p, channel n_base + 32 i + 8 g + 4 h + e
// hidden-layer epilogue: out[p][n] = bf16(relu(acc + bias[n]))
__device__ __forceinline__ void epilogue_hidden(const f32x16 (&acc)[2][2], const float *__restrict__ bias, int n_base, Tile out,
                                                int lane)
{
    const int p = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n_base + 32 * i + 8 * g + 4 * h;
            const float4 b = *reinterpret_cast<const float4 *>(bias + n);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float v0 = fmaxf(acc[i][j][4 * g] + b.x, 0.f), v1 = fmaxf(acc[i][j][4 * g + 1] + b.y, 0.f);
                const float v2 = fmaxf(acc[i][j][4 * g + 2] + b.z, 0.f), v3 = fmaxf(acc[i][j][4 * g + 3] + b.w, 0.f);
                *reinterpret_cast<uint2 *>(&out[32 * j + p][n]) = make_uint2(fpack(v0, v1), fpack(v2, v3));
            }
        }
}

// the workgroup copies a [64][256] bf16 tile from LDS to its rows of a pixel-major tensor (16 bytes per lane, whole rows)
__device__ __forceinline__ void store_tile(unsigned short *__restrict__ dst, int64_t p0, int64_t P, Tile src, int tid)
{
    if (!dst) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + 256 * q, row = id >> 5, c = (id & 31) * 8;
        if (p0 + row < P) *reinterpret_cast<uint4 *>(dst + (size_t)(p0 + row) * FH + c) = *reinterpret_cast<const uint4 *>(&src[row][c]);
    }
}

// dst[p][:] = bf16(dst + add) element-wise (fp32 add, one rounding: what gags_decoder_layer does with two sources)
__device__ __forceinline__ void add_tile(Tile dst, Tile add, int tid)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + 256 * q, row = id >> 5, c = (id & 31) * 8;
        const uint4 x = *reinterpret_cast<const uint4 *>(&dst[row][c]), y = *reinterpret_cast<const uint4 *>(&add[row][c]);
        *reinterpret_cast<uint4 *>(&dst[row][c]) =
            make_uint4(fpack(flo(x.x) + flo(y.x), fhi(x.x) + fhi(y.x)), fpack(flo(x.y) + flo(y.y), fhi(x.y) + fhi(y.y)),
                       fpack(flo(x.z) + flo(y.z), fhi(x.z) + fhi(y.z)), fpack(flo(x.w) + flo(y.w), fhi(x.w) + fhi(y.w)));
    }
}

struct FwdArgs {
    const float *x;              // [P, c_in] fp32 pixel-major (the rasterizer's own output), c_in <= 32
    const unsigned short *W[9];  // bf16, K contiguous, padded: [256, 32], 7 x [256, 256], [n_last, 256]
    const float *b[9];
    unsigned short *act[9];      // a0 [P, 32], then x1, t1, x2, x3, t4, x4, t6, t7 [P, 256]: null = not kept (inference)
    float *logits;               // [P, n_last] fp32
    int64_t P;
    int c_in, n_last;            // n_last % 256 == 0
};

__global__ __launch_bounds__(256, 2) void decoder_fwd_fused_kernel(FwdArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned short bufA[FT][FLD];
    __shared__ __attribute__((aligned(16))) unsigned short bufB[FT][FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * FT;
    const int n_base = 64 * wave;
    f32x16 acc[2][2];

    // input tile -> bufB[p][0..31] bf16 (zero-padded), also kept as a0 for the first layer's weight gradient
    for (int e = tid; e < FT * 32; e += 256) {
        const int row = e >> 5, c = e & 31;
        const int64_t p = p0 + row;
        const float v = (p < a.P && c < a.c_in) ? a.x[p * a.c_in + c] : 0.f;
        const unsigned short hv = (unsigned short)(fpack(v, 0.f) & 0xffffu);
        bufB[row][c] = hv;
        if (a.act[0] && p < a.P) a.act[0][p * 32 + c] = hv;
    }
    __syncthreads();
    // L0: a0 (B) -> x1 (A)
    layer_mma(acc, a.W[0], 32, n_base, 2, bufB, lane, true);
    epilogue_hidden(acc, a.b[0], n_base, bufA, lane);
    __syncthreads();
    store_tile(a.act[1], p0, a.P, bufA, tid);
    // L1: x1 (A) -> t1 (B)
    layer_mma(acc, a.W[1], FH, n_base, 16, bufA, lane, true);
    epilogue_hidden(acc, a.b[1], n_base, bufB, lane);
    __syncthreads();
    store_tile(a.act[2], p0, a.P, bufB, tid);
    // L2: t1 (B) -> x2, written over t1 once every wave is done reading it; then A = x1 + x2
    layer_mma(acc, a.W[2], FH, n_base, 16, bufB, lane, true);
    __syncthreads();
    epilogue_hidden(acc, a.b[2], n_base, bufB, lane);
    __syncthreads();
    store_tile(a.act[3], p0, a.P, bufB, tid);
    add_tile(bufA, bufB, tid);
    __syncthreads();
    // L3: x1 + x2 (A) -> x3 (B)
    layer_mma(acc, a.W[3], FH, n_base, 16, bufA, lane, true);
    epilogue_hidden(acc, a.b[3], n_base, bufB, lane);
    __syncthreads();
    store_tile(a.act[4], p0, a.P, bufB, tid);
    // L4: x3 (B) -> t4 (A)
    layer_mma(acc, a.W[4], FH, n_base, 16, bufB, lane, true);
    epilogue_hidden(acc, a.b[4], n_base, bufA, lane);
    __syncthreads();
    store_tile(a.act[5], p0, a.P, bufA, tid);
    // L5: t4 (A) -> x4 over t4; then B = x3 + x4
    layer_mma(acc, a.W[5], FH, n_base, 16, bufA, lane, true);
    __syncthreads();
    epilogue_hidden(acc, a.b[5], n_base, bufA, lane);
    __syncthreads();
    store_tile(a.act[6], p0, a.P, bufA, tid);
    add_tile(bufB, bufA, tid);
    __syncthreads();
    // L6: x3 + x4 (B) -> t6 (A)
    layer_mma(acc, a.W[6], FH, n_base, 16, bufB, lane, true);
    epilogue_hidden(acc, a.b[6], n_base, bufA, lane);
    __syncthreads();
    store_tile(a.act[7], p0, a.P, bufA, tid);
    // L7: t6 (A) -> t7 (B)
    layer_mma(acc, a.W[7], FH, n_base, 16, bufA, lane, true);
    epilogue_hidden(acc, a.b[7], n_base, bufB, lane);
    __syncthreads();
    store_tile(a.act[8], p0, a.P, bufB, tid);
    // L8: t7 (B) -> fp32 logits [P, n_last], 256 channels per pass, straight from the accumulators (32 contiguous bytes per
    // pixel and store instruction; the four instructions of a channel block complete its 128-byte line)
    const int p = lane & 31, h = lane >> 5;
    for (int nb = 0; nb < a.n_last; nb += FH) {
        layer_mma(acc, a.W[8] + (size_t)nb * FH, FH, n_base, 16, bufB, lane, true);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + n_base + 32 * i + 8 * g + 4 * h;
                const float4 b = *reinterpret_cast<const float4 *>(a.b[8] + n);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t pg = p0 + 32 * j + p;
                    if (pg < a.P)
                        *reinterpret_cast<float4 *>(a.logits + (size_t)pg * a.n_last + n) =
                            make_float4(acc[i][j][4 * g] + b.x, acc[i][j][4 * g + 1] + b.y, acc[i][j][4 * g + 2] + b.z,
                                        acc[i][j][4 * g + 3] + b.w);
                }
            }
    }
}

}  // namespace

extern "C" int gags_decoder_fwd_fused(int64_t n_pix, int c_in, int n_last, const float *x, const void *const *w_bf16,
                                      const float *const *bias, void *const *acts_bf16, float *logits, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c_in <= 0 || c_in > 32 || n_last <= 0 || n_last % FH != 0 || !w_bf16 || !bias || !logits || (n_pix > 0 && !x))
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    FwdArgs a;
    a.x = x; a.logits = logits; a.P = n_pix; a.c_in = c_in; a.n_last = n_last;
    for (int i = 0; i < 9; ++i) {
        if (!w_bf16[i] || !bias[i]) return GAGS_EINVAL;
        a.W[i] = (const unsigned short *)w_bf16[i];
        a.b[i] = bias[i];
        a.act[i] = acts_bf16 ? (unsigned short *)acts_bf16[i] : nullptr;
    }
    hipLaunchKernelGGL(decoder_fwd_fused_kernel, dim3((unsigned)((n_pix + FT - 1) / FT)), dim3(256), 0, (hipStream_t)stream, a);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
