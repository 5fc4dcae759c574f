// N1: the per-pixel decoders of models/networks.py:109-248 -- stacks of 1x1 convolutions, i.e. a
// [pixels, C_in] x [C_in, C_out] GEMM per layer over 2.07 M pixels at 1080p -- on the 16-bit matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  The reference runs these convolutions through cuDNN, which by
// PyTorch's default uses TF32 (10-bit mantissa) on its RTX 4090; bf16 keeps 8 bits, fp32 accumulation is the same.
//
// Layout: activations are PIXEL-major [P, C] bf16 -- what the rasterizer writes is [H, W, D] already, so the
// 16-channel render feeds layer 0 without the reference's permute -- weights [C_out, C_in] bf16 (C_in contiguous:
// both MFMA operands are then 16-byte rows).  One kernel serves every layer:
//     Y[p, n] = act( sum_k (A1[p, k] (+ A2[p, k])) * W[n, k] + bias[n] ) (* mask) (+ E[p, n])
// with the second source for the residual sums (x1 + x2, x3 + x4 of CNN_decoder.forward) and, in the backward, the
// ReLU mask of the layer below and the residual's gradient in the epilogue.
#include "common.h"
#include "gags_next.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

constexpr int TM = 128, TN = 128, TK = 32;  // workgroup tile: 128 pixels x 128 outputs, K step 32
constexpr int LDK = TK + 8;                  // LDS row pitch in bf16 (80 B: 16-byte aligned, spreads the banks)

struct GemmArgs {
    const unsigned short *A1, *A2;  // [P, K] bf16, A2 optional (summed with A1 in fp32, rounded once)
    const unsigned short *W;        // [N, K] bf16
    const float *bias;              // [N] or null
    const unsigned short *mask_src; // [P, N] bf16 or null: output multiplied by (mask_src > 0)
    const unsigned short *E;        // [P, N] bf16 or null: added after the mask
    unsigned short *Y;              // [P, N] bf16 or null
    float *Yf;                      // [P, N] fp32 or null
    int64_t P;
    int N, K, relu;
};

__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned short As[TM][LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[TN][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;  // 2 x 2 waves, 64 x 64 outputs each
    const int64_t p0 = (int64_t)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging identity: row tid / 2, 16 consecutive k at 16 * (tid % 2)
    const int sr = tid >> 1, sh = (tid & 1) * 16;
    const int64_t arow = min(p0 + sr, a.P - 1);
    const int brow = min(n0 + sr, a.N - 1);
    for (int k0 = 0; k0 < a.K; k0 += TK) {
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.A1 + arow * a.K + k0 + sh);
            uint4 u0 = src[0], u1 = src[1];
            if (a.A2) {  // residual sum of two activations: add in fp32, round once
                const uint4 *s2 = reinterpret_cast<const uint4 *>(a.A2 + arow * a.K + k0 + sh);
                const uint4 v0 = s2[0], v1 = s2[1];
                auto add2 = [](unsigned x, unsigned y) {
                    const float lo = bf2f((unsigned short)(x & 0xffff)) + bf2f((unsigned short)(y & 0xffff));
                    const float hi = bf2f((unsigned short)(x >> 16)) + bf2f((unsigned short)(y >> 16));
                    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
                };
                u0 = make_uint4(add2(u0.x, v0.x), add2(u0.y, v0.y), add2(u0.z, v0.z), add2(u0.w, v0.w));
                u1 = make_uint4(add2(u1.x, v1.x), add2(u1.y, v1.y), add2(u1.z, v1.z), add2(u1.w, v1.w));
            }
            *reinterpret_cast<uint4 *>(&As[sr][sh]) = u0;
            *reinterpret_cast<uint4 *>(&As[sr][sh + 8]) = u1;
            const uint4 *wsrc = reinterpret_cast<const uint4 *>(a.W + (size_t)brow * a.K + k0 + sh);
            *reinterpret_cast<uint4 *>(&Bs[sr][sh]) = wsrc[0];
            *reinterpret_cast<uint4 *>(&Bs[sr][sh + 8]) = wsrc[1];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < TK; ks += 16) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const bf16x8 *>(&As[wy * 64 + i * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j] = *reinterpret_cast<const bf16x8 *>(&Bs[wx * 64 + j * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // The operands were swapped above (weights as "A", activations as "B"): an accumulator's lane then owns ONE pixel
    // (column = lane & 31) and 16 output channels (rows): channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the tile.
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t p = p0 + wy * 64 + i * 32 + (lane & 31);
        if (p >= a.P) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {  // four consecutive channels per group of accumulator registers
                const int n = n0 + wx * 64 + j * 32 + 8 * rq + 4 * (lane >> 5);
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = acc[i][j][4 * rq + q] + (a.bias ? a.bias[n + q] : 0.f);
                    if (a.relu) t = fmaxf(t, 0.f);
                    v[q] = t;
                }
                const size_t o = (size_t)p * a.N + n;
                if (a.mask_src) {
                    const uint2 m = *reinterpret_cast<const uint2 *>(a.mask_src + o);
                    v[0] = bf2f((unsigned short)(m.x & 0xffff)) > 0.f ? v[0] : 0.f;
                    v[1] = bf2f((unsigned short)(m.x >> 16)) > 0.f ? v[1] : 0.f;
                    v[2] = bf2f((unsigned short)(m.y & 0xffff)) > 0.f ? v[2] : 0.f;
                    v[3] = bf2f((unsigned short)(m.y >> 16)) > 0.f ? v[3] : 0.f;
                }
                if (a.E) {
                    const uint2 e = *reinterpret_cast<const uint2 *>(a.E + o);
                    v[0] += bf2f((unsigned short)(e.x & 0xffff)); v[1] += bf2f((unsigned short)(e.x >> 16));
                    v[2] += bf2f((unsigned short)(e.y & 0xffff)); v[3] += bf2f((unsigned short)(e.y >> 16));
                }
                if (a.Y)
                    *reinterpret_cast<uint2 *>(a.Y + o) = make_uint2((unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16),
                                                                     (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16));
                if (a.Yf) *reinterpret_cast<float4 *>(a.Yf + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// fp32 [P, C] (pixel-major: the rasterizer's own [H, W, D] output) -> bf16 [P, Cp], zero-padded to Cp >= C
__global__ __launch_bounds__(256) void to_bf16_pad_kernel(int64_t P, int C, int Cp, const float *__restrict__ x,
                                                          unsigned short *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * Cp) return;
    const int64_t p = i / Cp;
    const int c = (int)(i - p * Cp);
    y[i] = c < C ? f2bf(x[p * C + c]) : (unsigned short)0;
}

// head of a decoder: pixel-major fp32 logits x[P, C] -> CHANNEL-major out[C, P] (the reference's [C, H, W]),
// mode 0: F.normalize(dim=0) = x / max(||x||_2, 1e-12)   (CNN_decoder, models/networks.py:192)
// mode 1: softmax over the channels                      (CNN_scale_decoder, :242)
// One workgroup = 64 pixels; the transpose goes through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void head_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                   float *__restrict__ out)
{
    __shared__ float tile[64][65];
    __shared__ float stat[64][2];
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int tid = threadIdx.x;
    // pass 1: per-pixel statistic (thread = pixel x quarter of the channels)
    {
        const int px = tid >> 2, q = tid & 3;
        const int64_t p = min(p0 + px, P - 1);
        float s = 0.f, m = -3.0e38f;
        for (int c = q; c < C; c += 4) {
            const float v = x[p * ld + c];
            s = fmaf(v, v, s);
            m = fmaxf(m, v);
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2));
        float z = 0.f;
        if (mode == 1) {
            for (int c = q; c < C; c += 4) z += expf(x[p * ld + c] - m);
            z += __shfl_xor(z, 1); z += __shfl_xor(z, 2);
        }
        if (q == 0) { stat[px][0] = mode == 0 ? fmaxf(sqrtf(s), 1e-12f) : m; stat[px][1] = z; }
    }
    __syncthreads();
    for (int cb = 0; cb < C; cb += 64) {
        // load [64 px][64 ch] pixel-major (channels fastest), store channel-major (pixels fastest)
        for (int e = tid; e < 64 * 64; e += 256) {
            const int px = e >> 6, c = e & 63;
            const int64_t p = min(p0 + px, P - 1);
            float v = 0.f;
            if (cb + c < C) {
                v = x[p * ld + cb + c];
                v = mode == 0 ? v / stat[px][0] : expf(v - stat[px][0]) / stat[px][1];
            }
            tile[px][c] = v;
        }
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += 256) {
            const int c = e >> 6, px = e & 63;
            if (cb + c < C && p0 + px < P) out[(size_t)(cb + c) * P + p0 + px] = tile[px][c];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int gags_decoder_pack_input(int64_t n_pix, int c, int c_pad, const float *x, void *y_bf16, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || c_pad < c || c_pad % 32 != 0 || (n_pix > 0 && (!x || !y_bf16))) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(to_bf16_pad_kernel, dim3((unsigned)((n_pix * c_pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_pix, c, c_pad, x, (unsigned short *)y_bf16);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_decoder_layer(int64_t n_pix, int n_out, int k_in, const void *a1, const void *a2, const void *w,
                                  const float *bias, int relu, const void *mask_src, const void *residual, void *y_bf16,
                                  float *y_f32, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || n_out <= 0 || k_in <= 0 || k_in % 32 != 0 || n_out % 4 != 0 || !a1 || !w || (!y_bf16 && !y_f32))
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    GemmArgs g;
    g.A1 = (const unsigned short *)a1; g.A2 = (const unsigned short *)a2; g.W = (const unsigned short *)w; g.bias = bias;
    g.mask_src = (const unsigned short *)mask_src; g.E = (const unsigned short *)residual;
    g.Y = (unsigned short *)y_bf16; g.Yf = y_f32; g.P = n_pix; g.N = n_out; g.K = k_in; g.relu = relu;
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3((unsigned)((n_pix + TM - 1) / TM), (unsigned)((n_out + TN - 1) / TN)), dim3(256), 0,
                       (hipStream_t)stream, g);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_decoder_head(int64_t n_pix, int c, int ld, int mode, const float *x, float *out, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || ld < c || (mode != 0 && mode != 1) || (n_pix > 0 && (!x || !out))) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)((n_pix + 63) / 64)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld, mode, x, out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
