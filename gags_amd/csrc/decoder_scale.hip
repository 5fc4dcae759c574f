// CNN_scale_decoder (models/networks.py:220-248: 16 -> 64 -> 128 -> 64 -> 32 -> 16 -> 3, ReLU between) as ONE kernel per
// direction in the fast bf16 mode.  Layer by layer (csrc/decoder.hip) the six small GEMMs are bound by the HBM traffic of
// their activations (0.6 KB per pixel and direction): 1.05 ms forward, ~1.6 ms for the input-gradient chain at 1080p.
// All its matrices together are 40 KB, so here a WAVE owns a tile of 32 pixels and walks the six layers alone: activations
// ping-pong between two LDS buffers of its own (no workgroup barrier anywhere), weights arrive in MFMA-fragment order
// straight from L1 / L2 (one coalesced kilobyte per MFMA), every layer's output leaves once as the contiguous
// [32 pixels x N] chunk it is in the pixel-major activation tensor (the weight gradients read those), plus the ReLU
// decisions as bits for the backward.
// Arithmetic is that of gags_decoder_layer (bf16 operands, fp32 accumulation in ascending k, bias + ReLU in fp32, one
// rounding to bf16): BIT-IDENTICAL to the layer-by-layer chain (tests/test_decoders_gpu.py).
// Padded widths (gags_amd/decoders.py: _pack_weights, every dimension to a multiple of 32):
//   K = 32, 64, 128, 64, 32, 32   N = 64, 128, 64, 32, 32, 32   (real: 16 -> 64 -> 128 -> 64 -> 32 -> 16 -> 3)
#include "common.h"
#include "half16.h"
#include "gags_next.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
using gags_h16::h16_mfma;
typedef float sf32x2_t __attribute__((ext_vector_type(2)));

constexpr int SP = 32;          // pixels per tile = per wave
constexpr int SLD = 128 + 8;    // LDS row pitch in bf16 (272 B: 16-byte aligned rows, consecutive rows 4 banks apart)
constexpr int SNL = 6;          // layers
constexpr int SK[SNL] = {32, 64, 128, 64, 32, 32};
constexpr int SN[SNL] = {64, 128, 64, 32, 32, 32};
constexpr int SBOFF[SNL] = {0, 64, 192, 256, 288, 320};  // offsets of the layers' biases in the staged array (352 floats)
constexpr int SMW[SNL] = {0, 2, 6, 8, 9, 10};            // offsets of the layers' ReLU mask words (N / 32 words each): 11 per pixel

typedef unsigned short (*STile)[SLD];

struct SFwdArgs {
    const float *x;              // [P, c_in] fp32 pixel-major, c_in <= 32
    const unsigned short *W[SNL];  // bf16, fragment order [N / 32][K / 16][64][8]
    const float *b[SNL];         // fp32 [N]
    unsigned short *act[SNL];    // a0 [P, 32], a1 [P, 64], a2 [P, 128], a3 [P, 64], a4 [P, 32], a5 [P, 32] (null: not kept)
    unsigned *mask;              // [P, 11] ReLU bits of a1 .. a5 (null: not kept)
    float *logits;               // [P, 32] fp32 (3 real columns), or null with `soft`
    float *soft;                 // [3, P] fp32 channel-major: softmax over the three real logits (the head fused in), or null
    int64_t P;
    int c_in;
};

struct SBwdArgs {
    const unsigned short *dz5;   // [P, 32] bf16: gradient of the logits (from the head's backward)
    const unsigned short *Wt[SNL];  // bf16 TRANSPOSED padded matrices [K, N] in fragment order [K / 32][N / 16][64][8] (layers 1 .. 5)
    const unsigned *mask;        // [P, 11]
    unsigned short *dz[SNL];     // dz0 [P, 64], dz1 [P, 128], dz2 [P, 64], dz3 [P, 32], dz4 [P, 32] (what the weight gradients contract)
    int64_t P;
};

// acc[t] (+)= W[32 t .. 32 t + 31][:] in[p][:]: lane (p = lane & 31, h = lane >> 5) of the accumulator owns pixel p and
// channels 32 t + 8 g + 4 h + e (element 4 g + e)
template <int K, int NT>
__device__ __forceinline__ void slayer_mma(f32x16 (&acc)[4], const unsigned short *__restrict__ Wf, STile in, int lane)
{
    constexpr int KS = K / 16;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&in[lane & 31][16 * ks + 8 * (lane >> 5)]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Wf + ((size_t)(t * KS + ks) * 64 + lane) * 8);
            acc[t] = h16_mfma(a, b, acc[t]);
        }
    }
}

// the tile [32][N] of an LDS buffer -> its contiguous place in a pixel-major [P, N] tensor, 16 bytes per lane
template <int N>
__device__ __forceinline__ void stile_store(unsigned short *__restrict__ dst, int64_t p0, int64_t P, STile src, int lane)
{
    if (!dst) return;
    constexpr int C8 = N / 8;  // 16-byte pieces per row
#pragma unroll
    for (int q = 0; q < SP * C8 / 64; ++q) {
        const int id = lane + 64 * q, row = id / C8, c = (id - row * C8) * 8;
        if (p0 + row < P) {  // (streaming store: read next by a weight-gradient kernel, from HBM either way)
            typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
            const uint4 v = *reinterpret_cast<const uint4 *>(&src[row][c]);
            __builtin_nontemporal_store(nt_u4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u4 *>(dst + (size_t)(p0 + row) * N + c));
        }
    }
}

// hidden-layer epilogue: out[p][n] = bf16(relu(acc + bias[n])), and the ReLU decisions as bits (word t of the layer, bit
// n % 32).  Packed instructions as in csrc/decoder_fused.hip: v_pk_add_f32, v_cvt_pk_bf16_f32, ReLU = v_pk_max_i16 with 0
// on the packed pair, decisions = v_pk_min_u16(pair, 1).
template <int NT>
__device__ __forceinline__ void sepilogue(const f32x16 (&acc)[4], const float *__restrict__ bias, STile out, int lane, bool keep /* uniform */,
                                          unsigned *__restrict__ mask_px /* this lane's pixel's words of the layer (null past the image) */)
{
    const int p = lane & 31, h = lane >> 5;
    const unsigned one2 = 0x00010001u;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        unsigned bits = 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 32 * t + 8 * g + 4 * h;
            const float4 b = *reinterpret_cast<const float4 *>(bias + n);
            const sf32x2_t v01 = sf32x2_t{acc[t][4 * g], acc[t][4 * g + 1]} + sf32x2_t{b.x, b.y};
            const sf32x2_t v23 = sf32x2_t{acc[t][4 * g + 2], acc[t][4 * g + 3]} + sf32x2_t{b.z, b.w};
            unsigned u0 = gags_h16::h16_pack_sat(v01[0], v01[1]);
            unsigned u1 = gags_h16::h16_pack_sat(v23[0], v23[1]);
            unsigned m0, m1;
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(u0) : "v"(u0));
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(u1) : "v"(u1));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(m0) : "v"(u0), "v"(one2));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(m1) : "v"(u1), "v"(one2));
            *reinterpret_cast<uint2 *>(&out[p][n]) = make_uint2(u0, u1);
            const unsigned nib = ((m0 | (m0 >> 15)) & 3u) | (((m1 | (m1 >> 15)) & 3u) << 2);
            bits |= nib << (8 * g);
        }
        if (keep) {  // this half-wave's nibbles sit at bits 8 g + 4 h of the word
            const unsigned mine = bits << (4 * h);
            const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
            if (h == 0 && mask_px) mask_px[t] = sw[0] | sw[1];
        }
    }
}

__global__ __launch_bounds__(256, 2) void sdec_fwd_fused_kernel(SFwdArgs a)
{
    gags_h16::h16_saturate_mode();  // (f16 tier: conversions saturate in hardware; half16.h)
    __shared__ __attribute__((aligned(16))) unsigned short buf[4][2][SP][SLD];  // per wave: two ping-pong tiles (2 x 8.5 KB)
    __shared__ __attribute__((aligned(16))) float bias_s[352];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        float bv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q;
            int l = 0;
#pragma unroll
            for (int i = 1; i < SNL; ++i) l = e >= SBOFF[i] ? i : l;
            bv[q] = e < 352 ? a.b[l][e - SBOFF[l]] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (tid + 256 * q < 352) bias_s[tid + 256 * q] = bv[q];
    }
    __syncthreads();  // (the only one: from here on a wave works alone)
    STile A = buf[wave][0], B = buf[wave][1];
    const int64_t n_tiles = (a.P + SP - 1) / SP;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t p0 = tile * SP;
        {   // input tile -> A[p][0..31] bf16 (zero-padded), also kept as a0 for the first layer's weight gradient
            float xv[SP * 32 / 64];
#pragma unroll
            for (int q = 0; q < SP * 32 / 64; ++q) {
                const int e = lane + 64 * q, row = e >> 5, c = e & 31;
                xv[q] = a.x[min(p0 + row, a.P - 1) * a.c_in + min(c, a.c_in - 1)];
            }
#pragma unroll
            for (int q = 0; q < SP * 32 / 64; ++q) {
                const int e = lane + 64 * q, row = e >> 5, c = e & 31;
                const float v = (p0 + row < a.P && c < a.c_in) ? xv[q] : 0.f;
                const sf32x2_t pr = {v, 0.f};
                A[row][c] = (unsigned short)(gags_h16::h16_pack_sat(pr[0], pr[1]) & 0xffffu);
            }
        }
        __builtin_amdgcn_wave_barrier();
        stile_store<32>(a.act[0], p0, a.P, A, lane);
        f32x16 acc[4];
        const int64_t pg = min(p0 + (lane & 31), a.P - 1);
        unsigned *mp = (a.mask && p0 + (lane & 31) < a.P) ? a.mask + pg * 11 : nullptr;
#define GAGS_SLAYER(L, IN, OUT)                                                                                             \
        slayer_mma<SK[L], SN[L] / 32>(acc, a.W[L], IN, lane);                                                               \
        sepilogue<SN[L] / 32>(acc, bias_s + SBOFF[L], OUT, lane, a.mask != nullptr, mp ? mp + SMW[L] : nullptr);             \
        __builtin_amdgcn_wave_barrier();                                                                                    \
        stile_store<SN[L]>(a.act[L + 1], p0, a.P, OUT, lane);
        GAGS_SLAYER(0, A, B)
        GAGS_SLAYER(1, B, A)
        GAGS_SLAYER(2, A, B)
        GAGS_SLAYER(3, B, A)
        GAGS_SLAYER(4, A, B)
#undef GAGS_SLAYER
        // L5: a5 (B) -> fp32 logits [P, 32]
        slayer_mma<SK[5], 1>(acc, a.W[5], B, lane);
        {
            const int p = lane & 31, h = lane >> 5;
            if (a.soft) {
                // CNN_scale_decoder's head (models/networks.py:248, softmax over the 3 channels) on the accumulator itself: the
                // three real logits of pixel p sit in lane p (h == 0), elements 0..2.  Same operations in the same order as
                // head_small_kernel (csrc/decoder.hip) on the stored logits: the same bits, without writing 128 B per pixel
                // of padded fp32 logits and reading them back (round 6)
                if (h == 0 && p0 + p < a.P) {
                    const float e0 = acc[0][0] + bias_s[SBOFF[5]], e1 = acc[0][1] + bias_s[SBOFF[5] + 1], e2 = acc[0][2] + bias_s[SBOFF[5] + 2];
                    const float m = fmaxf(fmaxf(fmaxf(-3.0e38f, e0), e1), e2);
                    float z = 0.f;
                    z += expf(e0 - m); z += expf(e1 - m); z += expf(e2 - m);
                    a.soft[p0 + p] = expf(e0 - m) / z;
                    a.soft[(size_t)a.P + p0 + p] = expf(e1 - m) / z;
                    a.soft[2 * (size_t)a.P + p0 + p] = expf(e2 - m) / z;
                }
            } else if (p0 + p < a.P) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 8 * g + 4 * h;
                    const float4 b = *reinterpret_cast<const float4 *>(bias_s + SBOFF[5] + n);
                    *reinterpret_cast<float4 *>(a.logits + (size_t)(p0 + p) * 32 + n) =
                        make_float4(acc[0][4 * g] + b.x, acc[0][4 * g + 1] + b.y, acc[0][4 * g + 2] + b.z, acc[0][4 * g + 3] + b.w);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// input-gradient epilogue: out[p][n] = bf16(acc) where the activation's ReLU bit is set, else 0 (no bias)
template <int NT>
__device__ __forceinline__ void sepilogue_dgrad(const f32x16 (&acc)[4], const unsigned (&mw)[4], STile out, int lane)
{
    const int p = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 32 * t + 8 * g + 4 * h;
            const unsigned nib = mw[t] >> (8 * g + 4 * h);
            const float v0 = (nib & 1u) ? acc[t][4 * g] : 0.f, v1 = (nib & 2u) ? acc[t][4 * g + 1] : 0.f;
            const float v2 = (nib & 4u) ? acc[t][4 * g + 2] : 0.f, v3 = (nib & 8u) ? acc[t][4 * g + 3] : 0.f;
            const unsigned u0 = gags_h16::h16_pack_sat(v0, v1);
            const unsigned u1 = gags_h16::h16_pack_sat(v2, v3);
            *reinterpret_cast<uint2 *>(&out[p][n]) = make_uint2(u0, u1);
        }
}

// The five input-gradient GEMMs dz5 -> dz4 -> ... -> dz0 (dz_{i-1} = (dz_i W_i) . [a_i > 0]), same tiling as the forward;
// layer i's "weight" is the transposed matrix Wt_i [K_i, N_i]: output width SK[i], contraction over SN[i].
__global__ __launch_bounds__(256, 2) void sdec_bwd_fused_kernel(SBwdArgs a)
{
    gags_h16::h16_saturate_mode();  // (f16 tier: conversions saturate in hardware; half16.h)
    __shared__ __attribute__((aligned(16))) unsigned short buf[4][2][SP][SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    STile A = buf[wave][0], B = buf[wave][1];
    const int64_t n_tiles = (a.P + SP - 1) / SP;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t p0 = tile * SP;
        const int64_t pg = min(p0 + (lane & 31), a.P - 1);
        // this lane's pixel's ReLU words (a1: 0-1, a2: 2-5, a3: 6-7, a4: 8, a5: 9) and the dz5 tile: all requested at once
        unsigned mk[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) mk[q] = a.mask[pg * 11 + q];
        uint4 d5[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int id = lane + 64 * q, row = id >> 2, c = (id & 3) * 8;
            d5[q] = *reinterpret_cast<const uint4 *>(a.dz5 + (size_t)min(p0 + row, a.P - 1) * 32 + c);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int id = lane + 64 * q, row = id >> 2, c = (id & 3) * 8;
            *reinterpret_cast<uint4 *>(&A[row][c]) = d5[q];
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 acc[4];
        unsigned mw[4];
        // (L = the forward layer whose transposed matrix is applied; output = dz[L - 1], width SK[L], mask = a_L's words at SMW[L - 1])
#define GAGS_SDGRAD(L, IN, OUT)                                                                                             \
        slayer_mma<SN[L], SK[L] / 32>(acc, a.Wt[L], IN, lane);                                                              \
        _Pragma("unroll") for (int t = 0; t < SK[L] / 32; ++t) mw[t] = mk[SMW[L - 1] + t];                                  \
        sepilogue_dgrad<SK[L] / 32>(acc, mw, OUT, lane);                                                                    \
        __builtin_amdgcn_wave_barrier();                                                                                    \
        stile_store<SK[L]>(a.dz[L - 1], p0, a.P, OUT, lane);
        GAGS_SDGRAD(5, A, B)
        GAGS_SDGRAD(4, B, A)
        GAGS_SDGRAD(3, A, B)
        GAGS_SDGRAD(2, B, A)
        GAGS_SDGRAD(1, A, B)
#undef GAGS_SDGRAD
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int GAGS_DEC(gags_scale_decoder_bwd_fused)(int64_t n_pix, const void *dz_last_bf16, const void *const *wt_bf16, const void *masks,
                                            void *const *dz_bf16, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || !dz_last_bf16 || !wt_bf16 || !masks || !dz_bf16) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    SBwdArgs a;
    a.dz5 = (const unsigned short *)dz_last_bf16; a.mask = (const unsigned *)masks; a.P = n_pix;
    a.Wt[0] = nullptr; a.dz[5] = nullptr;
    for (int i = 1; i < SNL; ++i) {
        if (!wt_bf16[i] || !dz_bf16[i - 1]) return GAGS_EINVAL;
        a.Wt[i] = (const unsigned short *)wt_bf16[i];
        a.dz[i - 1] = (unsigned short *)dz_bf16[i - 1];
    }
    const int64_t n_tiles = (n_pix + SP - 1) / SP;
    const int64_t want = (n_tiles + 3) / 4;
    const unsigned grid = (unsigned)(want < 256 * 2 * 8 ? want : 256 * 2 * 8);
    hipLaunchKernelGGL(sdec_bwd_fused_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int GAGS_DEC(gags_scale_decoder_fwd_fused_head)(int64_t n_pix, int c_in, const float *x, const void *const *w_bf16,
                                                 const float *const *bias, void *const *acts_bf16, void *masks, float *logits,
                                                 float *softmax3, void *stream);

extern "C" int GAGS_DEC(gags_scale_decoder_fwd_fused)(int64_t n_pix, int c_in, const float *x, const void *const *w_bf16,
                                            const float *const *bias, void *const *acts_bf16, void *masks, float *logits,
                                            void *stream)
{
    if (!logits) return GAGS_EINVAL;
    return GAGS_DEC(gags_scale_decoder_fwd_fused_head)(n_pix, c_in, x, w_bf16, bias, acts_bf16, masks, logits, nullptr, stream);
}

extern "C" int GAGS_DEC(gags_scale_decoder_fwd_fused_head)(int64_t n_pix, int c_in, const float *x, const void *const *w_bf16,
                                                 const float *const *bias, void *const *acts_bf16, void *masks, float *logits,
                                                 float *softmax3, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c_in <= 0 || c_in > 32 || !w_bf16 || !bias || (!logits && !softmax3) || (n_pix > 0 && !x)) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    SFwdArgs a;
    a.x = x; a.logits = logits; a.soft = softmax3; a.P = n_pix; a.c_in = c_in; a.mask = (unsigned *)masks;
    for (int i = 0; i < SNL; ++i) {
        if (!w_bf16[i] || !bias[i]) return GAGS_EINVAL;
        a.W[i] = (const unsigned short *)w_bf16[i];
        a.b[i] = bias[i];
        a.act[i] = acts_bf16 ? (unsigned short *)acts_bf16[i] : nullptr;
    }
    const int64_t n_tiles = (n_pix + SP - 1) / SP;
    const int64_t want = (n_tiles + 3) / 4;
    const unsigned grid = (unsigned)(want < 256 * 2 * 8 ? want : 256 * 2 * 8);  // grid-stride beyond eight rounds of two workgroups per CU
    hipLaunchKernelGGL(sdec_fwd_fused_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
