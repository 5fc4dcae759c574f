// N1: the per-pixel decoders of models/networks.py:109-248 -- stacks of 1x1 convolutions, i.e. a
// [pixels, C_in] x [C_in, C_out] GEMM per layer over 2.07 M pixels at 1080p -- on the 16-bit matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  The reference runs these convolutions through cuDNN, which by
// PyTorch's default uses TF32 (10-bit mantissa) on its RTX 4090; bf16 keeps 8 bits, fp32 accumulation is the same.
//
// Layout: activations are PIXEL-major [P, C] bf16 -- what the rasterizer writes is [H, W, D] already, so the
// 16-channel render feeds layer 0 without the reference's permute -- weights [C_out, C_in] bf16 (C_in contiguous:
// both MFMA operands are then 16-byte rows).  One kernel serves every layer:
//     Y[p, n] = ( act( sum_k (A1[p, k] (+ A2[p, k])) * W[n, k] + bias[n] ) (+ E[p, n]) ) (* mask)
// with the second source for the residual sums (x1 + x2, x3 + x4 of CNN_decoder.forward) and, in the backward, the
// ReLU mask of the layer below and the residual's gradient in the epilogue.
#include "common.h"
#include "half16.h"
#include "gags_next.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// the 16-bit operand type (bfloat16, or IEEE half when compiled with -DGAGS_H16): csrc/half16.h
using gags_h16::h16_mfma;
__device__ __forceinline__ unsigned short f2bf(float f) { return gags_h16::h16_from(f); }  // (NaN stays NaN, round to nearest even)
__device__ __forceinline__ float bf2f(unsigned short h) { return gags_h16::h16_to(h); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return gags_h16::h16_pack(lo, hi); }
__device__ __forceinline__ float bf_lo(unsigned u) { return gags_h16::h16_lo(u); }
__device__ __forceinline__ float bf_hi(unsigned u) { return gags_h16::h16_hi(u); }

constexpr int TM = 128, TK = 64;             // workgroup tile: 128 pixels x (64 NJ) outputs, K step 64
constexpr int LDK = TK + 8;                  // LDS row pitch in bf16 (144 B: 16-byte aligned, spreads the banks)

struct GemmArgs {
    const unsigned short *A1, *A2;  // [P, K] bf16, A2 optional (summed with A1 in fp32, rounded once)
    const unsigned short *W;        // [N, K] bf16
    const float *bias;              // [N] or null
    const unsigned short *mask_src; // [P, N] bf16 or null: output multiplied by (mask_src > 0)
    const unsigned short *E;        // [P, N] bf16 or null: added before the mask
    unsigned short *Y;              // [P, N] bf16 or null
    unsigned short *Ypre;           // [P, N] bf16 or null: the value before the mask (a skip connection's gradient)
    float *Yf;                      // [P, N] fp32 or null
    int64_t P;
    int N, K, relu;
};

__device__ __forceinline__ unsigned add_bf16x2(unsigned x, unsigned y)
{
    return pack_bf16(bf_lo(x) + bf_lo(y), bf_hi(x) + bf_hi(y));
}

// NJ = 32-column blocks per wave: the workgroup tile is 128 pixels x TN = 64 NJ outputs (NJ = 4: a 256-wide layer in
// ONE tile, so the activations are read once).  Wider layers take several column tiles per pixel tile; the workgroup
// index is decoded so that those land on the SAME XCD back to back (hardware deals workgroups round-robin over the 8
// XCDs, each with its own L2): the second one finds the activation tile in that L2 instead of re-reading HBM.
template <int NJ, bool TWO>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs a, int n_tiles, unsigned p_tiles)
{
    constexpr int TN = 64 * NJ;
    __shared__ __attribute__((aligned(16))) unsigned short smem[(TM + TN) * LDK];
    unsigned short (*As)[LDK] = reinterpret_cast<unsigned short (*)[LDK]>(smem);
    unsigned short (*Bs)[LDK] = reinterpret_cast<unsigned short (*)[LDK]>(smem + TM * LDK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;  // 2 x 2 waves, 64 x (32 NJ) outputs each
    unsigned pt = blockIdx.x;
    int nt = 0;
    if (n_tiles > 1) {
        const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pt = (slot / n_tiles) * 8 + xcd;
        nt = slot % n_tiles;
        if (pt >= p_tiles) return;
    }
    const int64_t p0 = (int64_t)pt * TM;
    const int n0 = nt * TN;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging identity: 16-byte piece tid % 8 of rows tid / 8 + 32 q (eight consecutive lanes read one row's 128
    // contiguous bytes: fully coalesced); the next K step's operands are requested into registers before the current
    // one is multiplied (the K loop is only 4-8 steps long: no deeper pipeline).  Every address is clamped into the
    // operand, so the loads are unconditional (a predicated load is a branch and a wait of its own) and what lies
    // beyond K is zeroed by a select; the second activation source is added when the tile goes to LDS, not when it is
    // requested.
    const int sr = tid >> 3, sh = (tid & 7) * 8;
    uint4 ra[4], ra2[TWO ? 4 : 1], rb[2 * NJ];
    unsigned aoff[4], boff[2 * NJ];  // element offsets: P * K < 2^31 is checked by the entry
#pragma unroll
    for (int q = 0; q < 4; ++q) aoff[q] = (unsigned)(min(p0 + sr + 32 * q, a.P - 1) * a.K);
#pragma unroll
    for (int q = 0; q < 2 * NJ; ++q) boff[q] = (unsigned)(min(n0 + sr + 32 * q, a.N - 1) * a.K);
    bool fetched_ok = true;
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        fetched_ok = k0 + sh < a.K;  // K is a multiple of 32, the step is 64
        const int kc = fetched_ok ? k0 + sh : 0;
#pragma unroll
        for (int q = 0; q < 2 * NJ; ++q) rb[q] = *reinterpret_cast<const uint4 *>(a.W + boff[q] + kc);
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const uint4 *>(a.A1 + aoff[q] + kc);
        if constexpr (TWO) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ra2[q] = *reinterpret_cast<const uint4 *>(a.A2 + aoff[q] + kc);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        const unsigned keep = fetched_ok ? 0xffffffffu : 0u;  // (a select between two uint4 lvalues would pin them in memory)
        if constexpr (TWO) {  // residual sum of two activations: add in fp32, round once
#pragma unroll
            for (int q = 0; q < 4; ++q)
                ra[q] = make_uint4(add_bf16x2(ra[q].x, ra2[q].x), add_bf16x2(ra[q].y, ra2[q].y), add_bf16x2(ra[q].z, ra2[q].z),
                                   add_bf16x2(ra[q].w, ra2[q].w));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4 *>(&As[sr + 32 * q][sh]) =
                make_uint4(ra[q].x & keep, ra[q].y & keep, ra[q].z & keep, ra[q].w & keep);
#pragma unroll
        for (int q = 0; q < 2 * NJ; ++q) *reinterpret_cast<uint4 *>(&Bs[sr + 32 * q][sh]) =
                make_uint4(rb[q].x & keep, rb[q].y & keep, rb[q].z & keep, rb[q].w & keep);
    };
    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += TK) {
        commit();
        __syncthreads();
        if (k0 + TK < a.K) fetch(k0 + TK);
#pragma unroll
        for (int ks = 0; ks < TK; ks += 16) {
            bf16x8 af[2], bf[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const bf16x8 *>(&As[wy * 64 + i * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bf[j] = *reinterpret_cast<const bf16x8 *>(&Bs[wx * 32 * NJ + j * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = h16_mfma(bf[j], af[i], acc[i][j]);
        }
        __syncthreads();
    }
    // The operands were swapped above (weights as "A", activations as "B"): an accumulator's lane owns ONE pixel
    // (column = lane & 31) and 16 output channels (rows): channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of a 32-block.
    // Stored from there, every instruction would scatter 8-byte pieces 2 N bytes apart and HBM sees partial lines
    // (measured: 2.4 x the bytes, 25 B per write request).  Instead each wave passes 32 pixels x 64 channels at a time
    // through its own LDS patch in fp32 (bias and ReLU applied on the way in) and reads it back with 8 consecutive
    // channels per lane: the skip gradient E and the mask are then 16-byte reads and the results 16 / 32-byte stores,
    // eight lanes per 128-byte (bf16) or 256-byte (fp32) run of one pixel.
    constexpr int SP = 68;  // patch pitch in floats (272 B: conflict-free for both the b128 writes and reads)
    static_assert(4 * 32 * SP * 4 <= (TM + TN) * LDK * 2, "epilogue patches fit in the operand tiles' LDS");
    float *stg = reinterpret_cast<float *>(smem) + wave * 32 * SP;
    const int epx = lane & 31, ehalf = lane >> 5, cg = lane & 7;
    const bool full = p0 + TM <= a.P && n0 + TN <= a.N;  // workgroup-uniform: the interior needs no per-lane tests
    const float floor_ = a.relu ? 0.f : -3.4e38f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp) {
            const int nb = n0 + wx * 32 * NJ + jp * 64;  // first channel of this pass (wave-uniform)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int r0 = 4 * rq;
                    const f32x16 &c = acc[i][2 * jp + jj];
                    *reinterpret_cast<float4 *>(stg + epx * SP + jj * 32 + 8 * rq + 4 * ehalf) =
                        make_float4(c[r0], c[r0 + 1], c[r0 + 2], c[r0 + 3]);
                }
            __builtin_amdgcn_wave_barrier();
            const int n = nb + cg * 8;           // this lane's 8 channels: the same in all four pixel groups
            const int nc = min(n, a.N - 8);
            float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
                const float4 b0 = *reinterpret_cast<const float4 *>(a.bias + nc), b1 = *reinterpret_cast<const float4 *>(a.bias + nc + 4);
                bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
            }
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                const int px = t2 * 8 + (lane >> 3);
                const float4 lo = *reinterpret_cast<const float4 *>(stg + px * SP + cg * 8);
                const float4 hi = *reinterpret_cast<const float4 *>(stg + px * SP + cg * 8 + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q] + bs[q], floor_);
                const int64_t p = p0 + wy * 64 + i * 32 + px;
                const bool live = full || (p < a.P && n < a.N);
                const size_t o = (size_t)min(p, a.P - 1) * a.N + nc;  // always a valid address: loads are unconditional
                if (a.E) {
                    const uint4 e = *reinterpret_cast<const uint4 *>(a.E + o);
                    v[0] += bf_lo(e.x); v[1] += bf_hi(e.x); v[2] += bf_lo(e.y); v[3] += bf_hi(e.y);
                    v[4] += bf_lo(e.z); v[5] += bf_hi(e.z); v[6] += bf_lo(e.w); v[7] += bf_hi(e.w);
                }
                if (a.Ypre && live)
                    *reinterpret_cast<uint4 *>(a.Ypre + o) =
                        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
                if (a.mask_src) {
                    const uint4 m = *reinterpret_cast<const uint4 *>(a.mask_src + o);
                    const unsigned mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] = bf_lo(mw[q]) > 0.f ? v[2 * q] : 0.f;
                        v[2 * q + 1] = bf_hi(mw[q]) > 0.f ? v[2 * q + 1] : 0.f;
                    }
                }
                if (a.Y && live)
                    *reinterpret_cast<uint4 *>(a.Y + o) =
                        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
                if (a.Yf && live) {
                    *reinterpret_cast<float4 *>(a.Yf + o) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4 *>(a.Yf + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
            __builtin_amdgcn_wave_barrier();
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

// The heads for ld <= 512, ld % 32 == 0 (every decoder of the reference): one workgroup = 32 pixels, thread = pixel
// (tid >> 3) x channels (tid & 7) * 4 + 32 j held in registers, so the logits are read ONCE; the statistics are
// reduced over the eight lanes of a pixel; the transpose goes through a [channel][pixel ^ (channel & 31)] LDS tile
// (64 KB, conflict-free on the row side) and leaves as 128-byte rows.
constexpr int HP = 32;      // pixels per workgroup
constexpr int HJ = 16;      // float4 per thread (ld <= 512)

__device__ __forceinline__ void head_stats(const float4 (&v)[HJ], int J, int C, int c0, int mode, float &s0, float &s1)
{
    float s = 0.f, m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + 32 * j + k < C) { s = fmaf(e[k], e[k], s); m = fmaxf(m, e[k]); }
        }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
    if (mode == 0) { s0 = fmaxf(sqrtf(s), 1e-12f); s1 = 0.f; return; }
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + 32 * j + k < C) z += expf(e[k] - m);
        }
    z += __shfl_xor(z, 1); z += __shfl_xor(z, 2); z += __shfl_xor(z, 4);
    s0 = m; s1 = z;
}

__global__ __launch_bounds__(256) void head_fast_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                        float *__restrict__ out)
{
    __shared__ float tile[512 * HP];
    const int64_t p0 = (int64_t)blockIdx.x * HP;
    const int tid = threadIdx.x, px = tid >> 3, c0 = (tid & 7) * 4, J = ld >> 5;
    const int64_t p = min(p0 + px, P - 1);
    float4 v[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        v[j] = j < J ? *reinterpret_cast<const float4 *>(x + p * ld + c0 + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float s0, s1;
    head_stats(v, J, C, c0, mode, s0, s1);
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 32 * j + k;
                if (c < C) tile[c * HP + (px ^ (c & 31))] = mode == 0 ? e[k] / s0 : expf(e[k] - s0) / s1;
            }
        }
    __syncthreads();
    const int lp = tid & 31;
    if (p0 + lp < P)
        for (int c = tid >> 5; c < C; c += 8) out[(size_t)c * P + p0 + lp] = tile[c * HP + (lp ^ (c & 31))];
}

// Weight gradient of one layer: dW[n, k] += sum_p dz[p, n] * (a1[p, k] + a2[p, k]),  db[n] += sum_p dz[p, n].
// The contraction runs over PIXELS, the slow index of both operands, so the 32-pixel tiles are transposed on their
// way into LDS (8-byte stores of four pixels of one column) and both MFMA fragments become 16-byte rows again.
// Workgroup = (128 n) x (128 k) x (a chunk of pixels); chunks are combined with float atomics.
constexpr int WP = 32;          // pixels per step (8 groups of 4 x 32 groups of 4 columns = 256 threads)
constexpr int LDP2 = WP + 8;    // LDS row pitch (bf16)
constexpr int WCHUNK = 16384;   // pixels per workgroup

__global__ __launch_bounds__(256) void wgrad_bf16_kernel(int64_t P, int N, int K, const unsigned short *__restrict__ dz,
                                                         const unsigned short *__restrict__ a1,
                                                         const unsigned short *__restrict__ a2, float *__restrict__ dW,
                                                         float *__restrict__ db)
{
    __shared__ __attribute__((aligned(16))) unsigned short At[128][LDP2];  // [n][p]
    __shared__ __attribute__((aligned(16))) unsigned short Bt[128][LDP2];  // [k][p]
    __shared__ float bsh[8][128];  // bias partial sums per pixel-row group: summed in a fixed order (no atomics)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
    const int64_t pa = (int64_t)blockIdx.x * WCHUNK, pb = min(pa + WCHUNK, P);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging: thread = 4 consecutive pixels x 4 consecutive columns; a column's four pixels go to LDS as one 8-byte
    // store (the transposition), 8 stores per thread and step instead of 32 two-byte ones
    const int sp = (tid >> 5) * 4, sc = (tid & 31) * 4;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = db != nullptr && blockIdx.z == 0;
    for (int64_t p0 = pa; p0 < pb; p0 += WP) {
        uint2 zr[4], xr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t p = p0 + sp + r;
            zr[r] = xr[r] = make_uint2(0, 0);
            if (p < pb) {
                if (n0 + sc < N) zr[r] = *reinterpret_cast<const uint2 *>(dz + p * N + n0 + sc);
                if (k0 + sc < K) {
                    xr[r] = *reinterpret_cast<const uint2 *>(a1 + p * K + k0 + sc);
                    if (a2) {
                        const uint2 y = *reinterpret_cast<const uint2 *>(a2 + p * K + k0 + sc);
                        xr[r] = make_uint2(add_bf16x2(xr[r].x, y.x), add_bf16x2(xr[r].y, y.y));
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // column c of the thread's 4: its four pixels, packed
            auto col = [&](const uint2 (&v)[4]) {
                unsigned short e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned wv = (c < 2) ? v[r].x : v[r].y;
                    e[r] = (unsigned short)((c & 1) ? (wv >> 16) : (wv & 0xffff));
                }
                return make_uint2((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16));
            };
            const uint2 zc = col(zr), xc = col(xr);
            *reinterpret_cast<uint2 *>(&At[sc + c][sp]) = zc;
            *reinterpret_cast<uint2 *>(&Bt[sc + c][sp]) = xc;
            if (do_bias)
                bsum[c] += (bf2f((unsigned short)(zc.x & 0xffff)) + bf2f((unsigned short)(zc.x >> 16))) +
                           (bf2f((unsigned short)(zc.y & 0xffff)) + bf2f((unsigned short)(zc.y >> 16)));
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < WP; ks += 16) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const bf16x8 *>(&At[wy * 64 + i * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j] = *reinterpret_cast<const bf16x8 *>(&Bt[wx * 64 + j * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = h16_mfma(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }
    // accumulator: column = lane & 31 -> k, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wx * 64 + j * 32 + (lane & 31);
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wy * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < N) dW[((size_t)blockIdx.x * N + n) * K + k] = acc[i][j][r];  // this pixel chunk's partial matrix
            }
        }
    if (db != nullptr && blockIdx.z == 0) {  // (uniform over the workgroup)
#pragma unroll
        for (int c = 0; c < 4; ++c) bsh[tid >> 5][sc + c] = bsum[c];
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += bsh[q][tid];
            db[(size_t)blockIdx.x * N + n0 + tid] = t;
        }
    }
}

// Weight gradient for the 256-input layers (8 of CNN_decoder's 9): workgroup = 128 outputs (n) x all 256 inputs (k)
// x a chunk of pixels.  Both operands stay PIXEL-major in LDS exactly as they sit in memory (16-byte loads, 16-byte
// LDS writes, no shuffling in registers) and the MFMA fragments, which want eight consecutive PIXELS of one channel
// per lane, come out of gfx950's transposing LDS read: ds_read_b64_tr_b16 hands lane l of a 16-lane group the four
// rows of column l of a [4 pixels][16 channels] block.  Row pitches of 64 B mod 256 B put the four pixel rows of a
// read on four different quarters of the 64 banks.  The LDS image is double-buffered (one barrier per 32-pixel
// step) and the operands of the step after next are in flight in registers meanwhile; two workgroups per CU.
struct GagsTrue { static constexpr bool value = true; };
struct GagsFalse { static constexpr bool value = false; };
constexpr int W2P = 32;                       // pixels per step
constexpr int W2ZP = 128 * 2 + 64;            // bytes per pixel row, dz image (128 channels)
constexpr int W2XP = 256 * 2 + 64;            // bytes per pixel row, activation image (256 channels)
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s *lds_v4s;

template <bool TWO>
__global__ __launch_bounds__(256, 2) void wgrad256_kernel(int64_t P, int N, const unsigned short *__restrict__ dz,
                                                          const unsigned short *__restrict__ a1,
                                                          const unsigned short *__restrict__ a2, float *__restrict__ dW,
                                                          float *__restrict__ db, int64_t chunk, int n_tiles, unsigned n_chunks)
{
    constexpr int K = 256;
    __shared__ __attribute__((aligned(16))) unsigned char Zi[2][W2P * W2ZP];
    __shared__ __attribute__((aligned(16))) unsigned char Xi[2][W2P * W2XP];
    __shared__ float bsh[16][128];  // bias partial sums per pixel-row group: summed in a fixed order (no atomics)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;  // 2 x 2 waves: 64 n x 128 k each
    // the n tiles of one pixel chunk run back to back on the same XCD (they share the activation rows through its L2)
    unsigned ck = blockIdx.x;
    int nt = 0;
    if (n_tiles > 1) {
        const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        ck = (slot / n_tiles) * 8 + xcd;
        nt = slot % n_tiles;
        if (ck >= n_chunks) return;
    }
    const int n0 = nt * 128;
    const int64_t pa = (int64_t)ck * chunk, pb = min(pa + chunk, P);
    if (pa >= pb) return;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // loads: dz piece tid & 15 (8 channels) of rows tid / 16 + 16 q; activation piece tid & 31 of rows tid / 32 + 8 q
    const int zc = (tid & 15) * 8, zr = tid >> 4, xc = (tid & 31) * 8, xr = tid >> 5;
    // a step that lies inside the chunk (all but the last one of the last chunk) is addressed by ONE uniform base per stream and
    // the lane's six fixed 32-bit byte offsets (round 6: the per-row form cost ~100 VALU instructions per step, quarter-rate
    // 64-bit multiplies among them, on the SIMD whose matrix pipe waits meanwhile), without clamps and without row masks
    uint4 rz[2], rx[4], rx2[TWO ? 4 : 1];
    unsigned zo[2], xo[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) zo[q] = (unsigned)(((zr + 16 * q) * N + zc) * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) xo[q] = (unsigned)(((xr + 8 * q) * K + xc) * 2);
    auto fetch = [&](auto whole, int64_t p0) __attribute__((always_inline)) {
        if constexpr (decltype(whole)::value) {
            const char *zb = reinterpret_cast<const char *>(dz + p0 * N + n0), *xb = reinterpret_cast<const char *>(a1 + p0 * K);
            const char *xb2 = TWO ? reinterpret_cast<const char *>(a2 + p0 * K) : nullptr;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned o = zo[q];
                asm volatile("" : "+v"(o));  // (opaque: keeps the scalar-base + lane-offset form of the load)
                rz[q] = *reinterpret_cast<const uint4 *>(zb + o);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned o = xo[q];
                asm volatile("" : "+v"(o));
                rx[q] = *reinterpret_cast<const uint4 *>(xb + o);
                if constexpr (TWO) rx2[q] = *reinterpret_cast<const uint4 *>(xb2 + o);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) rz[q] = *reinterpret_cast<const uint4 *>(dz + min(p0 + zr + 16 * q, pb - 1) * N + n0 + zc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t o = min(p0 + xr + 8 * q, pb - 1) * K + xc;
                rx[q] = *reinterpret_cast<const uint4 *>(a1 + o);
                if constexpr (TWO) rx2[q] = *reinterpret_cast<const uint4 *>(a2 + o);
            }
        }
    };
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // (summed whether or not db is wanted: 24 instructions, no branch)
    auto commit = [&](auto whole, int buf, int64_t p0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint4 v = rz[q];
            if constexpr (!decltype(whole)::value) {
                const unsigned keep = p0 + zr + 16 * q < pb ? 0xffffffffu : 0u;  // rows beyond the chunk contribute zero
                v = make_uint4(v.x & keep, v.y & keep, v.z & keep, v.w & keep);
            }
            *reinterpret_cast<uint4 *>(&Zi[buf][(zr + 16 * q) * W2ZP + zc * 2]) = v;
            bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
            bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 v = rx[q];
            if constexpr (TWO)
                v = make_uint4(add_bf16x2(v.x, rx2[q].x), add_bf16x2(v.y, rx2[q].y), add_bf16x2(v.z, rx2[q].z), add_bf16x2(v.w, rx2[q].w));
            *reinterpret_cast<uint4 *>(&Xi[buf][(xr + 8 * q) * W2XP + xc * 2]) = v;  // (dz rows are zero there: no mask needed)
        }
    };
    constexpr GagsTrue yes{};
    constexpr GagsFalse no{};
    // fragment addresses of this lane inside an image (bytes): pixel row 8 (l >> 5) + ((l & 15) >> 2), channel
    // 16 ((l >> 4) & 1) + 4 (l & 3) of the 32-channel block; + 4 rows for the second half, + 16 rows for the second k step
    const int frow = 8 * (lane >> 5) + ((lane & 15) >> 2), fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int zoff = frow * W2ZP + (wy * 64 + fcol) * 2, xoff = frow * W2XP + (wx * 128 + fcol) * 2;
    auto frag = [&](const unsigned char *base) __attribute__((always_inline)) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)base);
        return lo;
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char *b = &Zi[buf][zoff + ks * 16 * W2ZP + i * 64];
                const v4s lo = frag(b), hi = frag(b + 4 * W2ZP);
                af[i] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned char *b = &Xi[buf][xoff + ks * 16 * W2XP + j * 64];
                const v4s lo = frag(b), hi = frag(b + 4 * W2XP);
                bf[j] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = h16_mfma(af[i], bf[j], acc[i][j]);
        }
    };
    const int64_t steps = (pb - pa + W2P - 1) / W2P, n_whole = (pb - pa) / W2P;
    if (n_whole > 0) fetch(yes, pa); else fetch(no, pa);
    if (n_whole > 0) commit(yes, 0, pa); else commit(no, 0, pa);
    if (n_whole > 1) fetch(yes, pa + W2P); else if (steps > 1) fetch(no, pa + W2P);
    __syncthreads();
    // step s computes on image s & 1 while the registers (step s + 1) go to the other image and are refilled for s + 2
    int64_t s = 0;
    for (; s + 2 < n_whole; ++s) {
        const int buf = (int)(s & 1);
        commit(yes, buf ^ 1, pa + (s + 1) * W2P);
        fetch(yes, pa + (s + 2) * W2P);
        compute(buf);
        gags_lds_barrier();  // (not __syncthreads(): that would land the fetch above before every step)
    }
    for (; s < steps; ++s) {  // the last two steps, and the chunk's ragged one
        const int buf = (int)(s & 1);
        if (s + 1 < steps) commit(no, buf ^ 1, pa + (s + 1) * W2P);
        if (s + 2 < steps) fetch(no, pa + (s + 2) * W2P);
        compute(buf);
        gags_lds_barrier();
    }
    // accumulator: column = lane & 31 -> k, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> n; 32 lanes = 128 contiguous bytes
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = wx * 128 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wy * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dW[((size_t)ck * N + n) * K + k] = acc[i][j][r];  // this pixel chunk's partial matrix
            }
        }
    if (db != nullptr) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bsh[zr][zc + q] = bsum[q];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += bsh[q][tid];
            db[(size_t)ck * N + n0 + tid] = t;
        }
    }
}

// Weight gradient of the narrow layers (N = 32 NB outputs, K = 32 KB inputs, NB KB <= 8: the scale decoder and the
// first layer of CNN_decoder): the whole N x K gradient fits one wave's accumulators, so every WAVE runs its own
// pipeline over its own range of pixels -- 16 pixels per step, pixel-major rows into a wave-private LDS patch, fragments
// through the transposing read -- with no workgroup barrier anywhere (LDS executes a wave's accesses in order); the
// four waves of a workgroup are summed in LDS before the global atomics.
template <int NB, int KB, bool TWO>
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(int64_t P, const unsigned short *__restrict__ dz,
                                                           const unsigned short *__restrict__ a1,
                                                           const unsigned short *__restrict__ a2, float *__restrict__ dW,
                                                           float *__restrict__ db, int64_t chunk)
{
    constexpr int N = 32 * NB, K = 32 * KB;
    constexpr int ZP = (N * 2 / 64 | 1) * 64, XP = (K * 2 / 64 | 1) * 64;  // row pitches: odd multiples of 64 B
    constexpr int PATCH = 16 * (ZP + XP);
    constexpr int RED = N * K * 4 + N * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(4 * PATCH > RED ? 4 * PATCH : RED)];
    __shared__ float bsh[4][64][8];  // bias partial sums per wave and lane: summed in a fixed order (no atomics)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char *Zi = smem + wave * PATCH, *Xi = Zi + 16 * ZP;
    const int64_t pa = ((int64_t)blockIdx.x * 4 + wave) * chunk, pb = min(pa + chunk, P);
    f32x16 acc[NB][KB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // 16-byte pieces lane + 64 q of the 16 x (N / 8) resp. 16 x (K / 8) pieces of a step
    constexpr int ZQ = N / 32, XQ = K / 32, ZR = N / 8, XR = K / 8;
    uint4 rz[ZQ], rx[XQ], rx2[TWO ? XQ : 1];
    auto fetch = [&](int64_t p0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < ZQ; ++q) {
            const int pc = lane + 64 * q;
            rz[q] = *reinterpret_cast<const uint4 *>(dz + min(p0 + pc / ZR, P - 1) * N + (pc % ZR) * 8);
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int pc = lane + 64 * q;
            const int64_t o = min(p0 + pc / XR, P - 1) * K + (pc % XR) * 8;
            rx[q] = *reinterpret_cast<const uint4 *>(a1 + o);
            if constexpr (TWO) rx2[q] = *reinterpret_cast<const uint4 *>(a2 + o);
        }
    };
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // channels (lane % ZR) * 8 .. + 7 (the same for every q)
    const bool do_bias = db != nullptr;
    auto commit = [&](int64_t p0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < ZQ; ++q) {
            const int pc = lane + 64 * q;
            const unsigned keep = p0 + pc / ZR < pb ? 0xffffffffu : 0u;  // rows beyond the range contribute zero
            const uint4 v = make_uint4(rz[q].x & keep, rz[q].y & keep, rz[q].z & keep, rz[q].w & keep);
            *reinterpret_cast<uint4 *>(Zi + (pc / ZR) * ZP + (pc % ZR) * 16) = v;
            if (do_bias) {
                bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
                bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
            }
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int pc = lane + 64 * q;
            uint4 v = rx[q];
            if constexpr (TWO)
                v = make_uint4(add_bf16x2(v.x, rx2[q].x), add_bf16x2(v.y, rx2[q].y), add_bf16x2(v.z, rx2[q].z), add_bf16x2(v.w, rx2[q].w));
            *reinterpret_cast<uint4 *>(Xi + (pc / XR) * XP + (pc % XR) * 16) = v;
        }
    };
    const int frow = 8 * (lane >> 5) + ((lane & 15) >> 2), fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    if (pa < pb) {
        fetch(pa);
        for (int64_t p0 = pa; p0 < pb; p0 += 16) {
            commit(p0);
            if (p0 + 16 < pb) fetch(p0 + 16);
            bf16x8 af[NB], bf[KB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const unsigned char *b = Zi + frow * ZP + (i * 32 + fcol) * 2;
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)b), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(b + 4 * ZP));
                af[i] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const unsigned char *b = Xi + frow * XP + (j * 32 + fcol) * 2;
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)b), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(b + 4 * XP));
                bf[j] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < KB; ++j) acc[i][j] = h16_mfma(af[i], bf[j], acc[i][j]);
        }
    }
    // sum the four waves in LDS (the patches are done with) -- one wave at a time, in wave order, plain read-modify-
    // write: a fixed order, hence reproducible bits -- then store the workgroup's partial matrix
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);
    for (int e = tid; e < N * K; e += 256) red[e] = 0.f;
    if (do_bias) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bsh[wave][lane][q] = bsum[q];
    }
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < KB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * K + j * 32 + (lane & 31)] += acc[i][j][r];
        }
        __syncthreads();
    }
    for (int e = tid; e < N * K; e += 256) dW[(size_t)blockIdx.x * N * K + e] = red[e];
    if (do_bias)
        for (int c = tid; c < N; c += 256) {  // column c: lanes c / 8 + ZR m of every wave, element c % 8
            float t = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int l = c >> 3; l < 64; l += ZR) t += bsh[wv][l][c & 7];
            db[(size_t)blockIdx.x * N + c] = t;
        }
}

inline int64_t narrow_chunk(int64_t n_pix)
{
    const int64_t waves = 4 * 768;  // three workgroups per CU
    return ((n_pix + waves - 1) / waves + 15) / 16 * 16;
}
inline unsigned narrow_parts(int64_t n_pix) { const int64_t c = narrow_chunk(n_pix); return (unsigned)((n_pix + 4 * c - 1) / (4 * c)); }

template <int NB, int KB>
void launch_wgrad_narrow(int64_t n_pix, const void *dz, const void *a1, const void *a2, float *d_w, float *d_b, hipStream_t stream)
{
    const int64_t chunk = narrow_chunk(n_pix);
    const unsigned grid = narrow_parts(n_pix);
    if (a2)
        hipLaunchKernelGGL((wgrad_narrow_kernel<NB, KB, true>), dim3(grid), dim3(256), 0, stream, n_pix, (const unsigned short *)dz,
                           (const unsigned short *)a1, (const unsigned short *)a2, d_w, d_b, chunk);
    else
        hipLaunchKernelGGL((wgrad_narrow_kernel<NB, KB, false>), dim3(grid), dim3(256), 0, stream, n_pix, (const unsigned short *)dz,
                           (const unsigned short *)a1, (const unsigned short *)a2, d_w, d_b, chunk);
}

// out[e] = sum of the partial results over the pixel chunks, in a FIXED order (=> reproducible): sixteen groups of a
// workgroup take the chunks g, g + 16, ... of 64 consecutive elements (up to 16 loads in flight each), their sums are
// added in group order.  (One thread per element walking all 768 chunks of a narrow layer was a chain of 48 round trips:
// 17-28 us for a 4 KB matrix, 0.5 ms per iteration over the thirty sums.)
// A weight matrix AND its bias row in one launch, written in the parameter's own shape (round 6; one launch per tensor
// before): the partial matrices are [n, k] (padded to multiples of 32), the outputs out_w[co, ci] and out_b[co]; every sum is
// multiplied by scale[0] when given (the f16 tier's power of two: exact).  Blocks [0, wb) sum the matrix, the rest the bias.
__global__ __launch_bounds__(1024) void sum_wparts_out_kernel(int n_parts, int n, int k, int co, int ci, unsigned wb,
                                                              const float *__restrict__ part_w, const float *__restrict__ part_b,
                                                              float *__restrict__ out_w, float *__restrict__ out_b,
                                                              const float *__restrict__ scale)
{
    __shared__ float sm[16][64];
    const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool bias = blockIdx.x >= wb;
    const int64_t n_el = bias ? co : (int64_t)co * ci, stride = bias ? n : (int64_t)n * k;
    const float *part = bias ? part_b : part_w;
    const int64_t e = (int64_t)(bias ? blockIdx.x - wb : blockIdx.x) * 64 + l;
    const int64_t ec = min(e, n_el - 1);
    const int64_t src = bias ? ec : (ec / ci) * k + ec % ci;
    float t = 0.f;
    int c = g;
    for (; c + 16 * 15 < n_parts; c += 16 * 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = part[(size_t)(c + 16 * i) * stride + src];
#pragma unroll
        for (int i = 0; i < 16; ++i) t += v[i];
    }
    for (; c < n_parts; c += 16) t += part[(size_t)c * stride + src];
    sm[g][l] = t;
    __syncthreads();
    if (g == 0 && e < n_el) {
        float r = sm[0][l];
#pragma unroll
        for (int i = 1; i < 16; ++i) r += sm[i][l];
        if (scale) r *= scale[0];
        (bias ? out_b : out_w)[e] = r;
    }
}

// backward of the output heads: channel-major cotangent G[C, P] and the saved pixel-major logits x[P, ld] ->
// pixel-major bf16 dz[P, ld] (columns >= C zero).
//   mode 0 (y = x / max(||x||, eps)):  dz = (g - y <y, g>) / max(||x||, eps)
//   mode 1 (y = softmax(x)):           dz = y * (g - <y, g>)
__global__ __launch_bounds__(256) void head_bwd_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                       const float *__restrict__ G, unsigned short *__restrict__ dz)
{
    __shared__ float tile[64][65];
    __shared__ float stat[64][3];
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int tid = threadIdx.x;
    // pass 1a: norm / (max, Z) per pixel from the logits
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
        if (q == 0) { stat[px][0] = mode == 0 ? fmaxf(sqrtf(s), 1e-12f) : m; stat[px][1] = z; stat[px][2] = 0.f; }
    }
    __syncthreads();
    // pass 1b: <y, g> per pixel (G is channel-major: transpose block-wise through LDS)
    float part = 0.f;  // thread = pixel tid & 63, channels (tid >> 6) + 4 j of each block
    for (int cb = 0; cb < C; cb += 64) {
        for (int e = tid; e < 64 * 64; e += 256) {
            const int c = e >> 6, px = e & 63;
            tile[px][c] = (cb + c < C && p0 + px < P) ? G[(size_t)(cb + c) * P + p0 + px] : 0.f;
        }
        __syncthreads();
        {
            const int px = tid & 63;
            const int64_t p = min(p0 + px, P - 1);
            for (int c = tid >> 6; c < 64 && cb + c < C; c += 4) {
                const float v = x[p * ld + cb + c];
                const float y = mode == 0 ? v / stat[px][0] : expf(v - stat[px][0]) / stat[px][1];
                part = fmaf(y, tile[px][c], part);
            }
        }
        __syncthreads();
    }
    atomicAdd(&stat[tid & 63][2], part);
    __syncthreads();
    // pass 2: dz, pixel-major bf16
    for (int cb = 0; cb < ld; cb += 64) {
        for (int e = tid; e < 64 * 64; e += 256) {
            const int c = e >> 6, px = e & 63;
            tile[px][c] = (cb + c < C && p0 + px < P) ? G[(size_t)(cb + c) * P + p0 + px] : 0.f;
        }
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += 256) {
            const int px = e >> 6, c = e & 63;
            if (p0 + px >= P || cb + c >= ld) continue;
            float d = 0.f;
            if (cb + c < C) {
                const float v = x[(p0 + px) * ld + cb + c], g = tile[px][c], dot = stat[px][2];
                if (mode == 0) {
                    const float nrm = stat[px][0];
                    d = (g - (v / nrm) * dot) / nrm;
                } else {
                    const float y = expf(v - stat[px][0]) / stat[px][1];
                    d = y * (g - dot);
                }
            }
            dz[(p0 + px) * ld + cb + c] = f2bf(d);
        }
        __syncthreads();
    }
}

// backward heads for ld <= 512, ld % 32 == 0: the same 32-pixel tiling as head_fast_kernel; the cotangent tile
// sits in LDS, the logits in registers, so G and x are each read once.
__global__ __launch_bounds__(256) void head_bwd_fast_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                            const float *__restrict__ G, unsigned short *__restrict__ dz)
{
    __shared__ float tile[512 * HP];
    const int64_t p0 = (int64_t)blockIdx.x * HP;
    const int tid = threadIdx.x, px = tid >> 3, c0 = (tid & 7) * 4, J = ld >> 5;
    const int64_t p = min(p0 + px, P - 1);
    float4 v[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        v[j] = j < J ? *reinterpret_cast<const float4 *>(x + p * ld + c0 + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    {
        // the cotangent rows in two batches of 32 loads in flight (a rolled loop would wait for every single row)
        const int lp = tid & 31, cw = tid >> 5;
        const int64_t pg = min(p0 + lp, P - 1);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float g[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int c = cw + 8 * (32 * b + i);
                g[i] = c < C ? G[(size_t)c * P + pg] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int c = cw + 8 * (32 * b + i);
                if (c < C) tile[c * HP + (lp ^ (c & 31))] = g[i];
            }
        }
    }
    float s0, s1;
    head_stats(v, J, C, c0, mode, s0, s1);
    __syncthreads();
    // reciprocals once per pixel: the per-element work is one or two FMAs (the result is rounded to bf16 anyway)
    const float inv0 = 1.f / s0, inv1 = mode == 1 ? 1.f / s1 : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 32 * j + k;
                if (c < C) {
                    const float y = mode == 0 ? e[k] : expf(e[k] - s0) * inv1;   // mode 0: the 1 / norm is applied to the sum
                    dot = fmaf(y, tile[c * HP + (px ^ (c & 31))], dot);
                }
            }
        }
    dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
    if (mode == 0) dot *= inv0;          // <y, g>
    const float k1 = dot * inv0 * inv0;  // mode 0: dz = g / n - x <y, g> / n^2
    if (p0 + px >= P) return;
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            unsigned short o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 32 * j + k;
                float d = 0.f;
                if (c < C) {
                    const float g = tile[c * HP + (px ^ (c & 31))];
                    if (mode == 0) d = fmaf(g, inv0, -e[k] * k1);
                    else d = (expf(e[k] - s0) * inv1) * (g - dot);
                }
                o[k] = f2bf(d);
            }
            uint2 w;
            w.x = (unsigned)o[0] | ((unsigned)o[1] << 16);
            w.y = (unsigned)o[2] | ((unsigned)o[3] << 16);
            *reinterpret_cast<uint2 *>(dz + p * ld + c0 + 32 * j) = w;
        }
}

// Pixel-major heads (layout 1): y[P][C] right next to the logits' own layout -- no transpose at all.  The [C,H,W]
// tensor the caller sees is then a permuted view of [H,W,C] memory, like the rasterizer's own output, and what the
// reference does with it next (compute_relvancy.py:266, evaluate_iou_loc.py:283: .permute(1,2,0)) is free.
__global__ __launch_bounds__(256) void head_pm_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                      float *__restrict__ out)
{
    const int tid = threadIdx.x, c0 = (tid & 7) * 4, J = ld >> 5;
    const int64_t pr = (int64_t)blockIdx.x * HP + (tid >> 3), p = min(pr, P - 1);
    float4 v[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        v[j] = j < J ? *reinterpret_cast<const float4 *>(x + p * ld + c0 + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float s0, s1;
    head_stats(v, J, C, c0, mode, s0, s1);
    if (pr >= P) return;
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J && c0 + 32 * j < C) {  // C % 4 == 0: the whole float4 is inside
            const float4 e = v[j];
            *reinterpret_cast<float4 *>(out + p * C + c0 + 32 * j) =
                mode == 0 ? make_float4(e.x / s0, e.y / s0, e.z / s0, e.w / s0)
                          : make_float4(expf(e.x - s0) / s1, expf(e.y - s0) / s1, expf(e.z - s0) / s1, expf(e.w - s0) / s1);
        }
}

__global__ __launch_bounds__(256) void head_bwd_pm_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                          const float *__restrict__ G, unsigned short *__restrict__ dz)
{
    const int tid = threadIdx.x, c0 = (tid & 7) * 4, J = ld >> 5;
    const int64_t pr = (int64_t)blockIdx.x * HP + (tid >> 3), p = min(pr, P - 1);
    float4 v[HJ], g[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        v[j] = j < J ? *reinterpret_cast<const float4 *>(x + p * ld + c0 + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        g[j] = (j < J && c0 + 32 * j < C) ? *reinterpret_cast<const float4 *>(G + p * C + c0 + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s0, s1;
    head_stats(v, J, C, c0, mode, s0, s1);
    const float inv0 = 1.f / s0, inv1 = mode == 1 ? 1.f / s1 : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J && c0 + 32 * j < C) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w}, gg[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) dot = fmaf(mode == 0 ? e[k] : expf(e[k] - s0) * inv1, gg[k], dot);
        }
    dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
    if (mode == 0) dot *= inv0;
    const float k1 = dot * inv0 * inv0;
    if (pr >= P) return;
#pragma unroll
    for (int j = 0; j < HJ; ++j)
        if (j < J) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w}, gg[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = 0.f;
                if (c0 + 32 * j + k < C) d[k] = mode == 0 ? fmaf(gg[k], inv0, -e[k] * k1) : (expf(e[k] - s0) * inv1) * (gg[k] - dot);
            }
            *reinterpret_cast<uint2 *>(dz + p * ld + c0 + 32 * j) = make_uint2(pack_bf16(d[0], d[1]), pack_bf16(d[2], d[3]));
        }
}

// Heads with a handful of channels (CNN_scale_decoder: softmax over 3): one thread per pixel, the whole row in
// registers; channel-major output / cotangent planes are coalesced along the pixels.  (The tiled kernels above issue 64
// mostly predicated-off loads per thread for such a head: 0.64 ms instead of 0.1.)
constexpr int HS_MAXC = 4;

__global__ __launch_bounds__(256) void head_small_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                         float *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float4 xv = *reinterpret_cast<const float4 *>(x + p * ld);
    const float e[4] = {xv.x, xv.y, xv.z, xv.w};
    float s = 0.f, m = -3.0e38f;
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) { s = fmaf(e[c], e[c], s); m = fmaxf(m, e[c]); }
    float z = 0.f;
    if (mode == 1) {
#pragma unroll
        for (int c = 0; c < HS_MAXC; ++c)
            if (c < C) z += expf(e[c] - m);
    }
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) out[(size_t)c * P + p] = mode == 0 ? e[c] / nrm : expf(e[c] - m) / z;
}

__global__ __launch_bounds__(256) void head_bwd_small_kernel(int64_t P, int C, int ld, int mode, const float *__restrict__ x,
                                                             const float *__restrict__ G, unsigned short *__restrict__ dz)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float4 xv = *reinterpret_cast<const float4 *>(x + p * ld);
    const float e[4] = {xv.x, xv.y, xv.z, xv.w};
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    float s = 0.f, m = -3.0e38f;
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) { g[c] = G[(size_t)c * P + p]; s = fmaf(e[c], e[c], s); m = fmaxf(m, e[c]); }
    float z = 0.f, y[4] = {0.f, 0.f, 0.f, 0.f};
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) {
            if (mode == 1) { y[c] = expf(e[c] - m); z += y[c]; } else y[c] = e[c] / nrm;
        }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) { if (mode == 1) y[c] /= z; dot = fmaf(y[c], g[c], dot); }
    float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) d[c] = mode == 0 ? (g[c] - y[c] * dot) / nrm : y[c] * (g[c] - dot);
    // the whole padded row: columns >= C are zero
    uint4 *dst = reinterpret_cast<uint4 *>(dz + p * ld);
    dst[0] = make_uint4(pack_bf16(d[0], d[1]), pack_bf16(d[2], d[3]), 0u, 0u);
    for (int q = 1; q < ld / 8; ++q) dst[q] = make_uint4(0u, 0u, 0u, 0u);
}

// backward of the softmax head from its OUTPUT y [C, P] (what the fused scale-decoder forward leaves: no logits are kept):
// dz = y (g - <y, g>), the same operations in the same order as head_bwd_small_kernel from its y onwards
__global__ __launch_bounds__(256) void softmax_bwd_small_y_kernel(int64_t P, int C, int ld, const float *__restrict__ Y,
                                                                  const float *__restrict__ G, unsigned short *__restrict__ dz)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float g[4] = {0.f, 0.f, 0.f, 0.f}, y[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) { g[c] = G[(size_t)c * P + p]; y[c] = Y[(size_t)c * P + p]; }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) dot = fmaf(y[c], g[c], dot);
    float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HS_MAXC; ++c)
        if (c < C) d[c] = y[c] * (g[c] - dot);
    uint4 *dst = reinterpret_cast<uint4 *>(dz + p * ld);
    dst[0] = make_uint4(pack_bf16(d[0], d[1]), pack_bf16(d[2], d[3]), 0u, 0u);
    for (int q = 1; q < ld / 8; ++q) dst[q] = make_uint4(0u, 0u, 0u, 0u);
}

// bf16 [P, ld] -> fp32 [P, C] (first C columns): the input gradient in the rasterizer's [H, W, D] layout
__global__ __launch_bounds__(256) void unpack_f32_kernel(int64_t P, int C, int ld, const unsigned short *__restrict__ x,
                                                         float *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * C) return;
    const int64_t p = i / C;
    y[i] = bf2f(x[p * ld + (i - p * C)]);
}

}  // namespace

namespace {
// one layer's parameters in every form the kernels read, in ONE launch (torch: zeros + slice copy + cast + transpose +
// two permuted copies + the bias pair = ~10 tiny kernels per layer, ~150 per iteration for the two decoders)
__global__ __launch_bounds__(256) void pack_layer_kernel(int co, int ci, int np, int kp, const float *__restrict__ w,
                                                         const float *__restrict__ b, unsigned short *__restrict__ wr,
                                                         unsigned short *__restrict__ wtr, unsigned short *__restrict__ wf,
                                                         unsigned short *__restrict__ wtf, float *__restrict__ bp)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < np) bp[e] = e < co ? b[e] : 0.f;
    if (e >= np * kp) return;
    const int n = e / kp, k = e - n * kp;
    const float v = (n < co && k < ci) ? w[(size_t)n * ci + k] : 0.f;
    const unsigned short h = gags_h16::h16_from(v);  // RN-even, as torch
    wr[(size_t)n * kp + k] = h;
    wtr[(size_t)k * np + n] = h;
    // fragment order of a [R, C] matrix: [R / 32][C / 16][2][32][8] (gags_amd/decoders.py: _frag_layout)
    wf[((((size_t)(n >> 5) * (kp >> 4) + (k >> 4)) * 2 + ((k >> 3) & 1)) * 32 + (n & 31)) * 8 + (k & 7)] = h;
    wtf[((((size_t)(k >> 5) * (np >> 4) + (n >> 4)) * 2 + ((n >> 3) & 1)) * 32 + (k & 31)) * 8 + (n & 7)] = h;
}
}  // namespace

extern "C" int GAGS_DEC(gags_decoder_pack_layer)(int co, int ci, const float *w, const float *b, void *w_bf16, void *wt_bf16, void *w_frag,
                                       void *wt_frag, float *bias_pad, void *stream)
{
    GAGS_CLEAR_ERR();
    if (co <= 0 || ci <= 0 || !w || !b || !w_bf16 || !wt_bf16 || !w_frag || !wt_frag || !bias_pad) return GAGS_EINVAL;
    const int np = (co + 31) / 32 * 32, kp = (ci + 31) / 32 * 32;
    hipLaunchKernelGGL(pack_layer_kernel, dim3((unsigned)((np * kp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, co, ci, np, kp, w, b,
                       (unsigned short *)w_bf16, (unsigned short *)wt_bf16, (unsigned short *)w_frag, (unsigned short *)wt_frag, bias_pad);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// every layer of a decoder in ONE launch (round 6: nine + six launches of ~5 us per iteration before): blockIdx.y = layer
namespace {
constexpr int PACK_MAX_LAYERS = 12;
struct PackArgs {
    int co[PACK_MAX_LAYERS], ci[PACK_MAX_LAYERS];
    const float *w[PACK_MAX_LAYERS], *b[PACK_MAX_LAYERS];
    unsigned short *wr[PACK_MAX_LAYERS], *wtr[PACK_MAX_LAYERS], *wf[PACK_MAX_LAYERS], *wtf[PACK_MAX_LAYERS];
    float *bp[PACK_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void pack_layers_kernel(PackArgs a)
{
    const int L = blockIdx.y;
    const int co = a.co[L], ci = a.ci[L], np = (co + 31) / 32 * 32, kp = (ci + 31) / 32 * 32;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < np) a.bp[L][e] = e < co ? a.b[L][e] : 0.f;
    if (e >= np * kp) return;
    const int n = e / kp, k = e - n * kp;
    const float v = (n < co && k < ci) ? a.w[L][(size_t)n * ci + k] : 0.f;
    const unsigned short h = gags_h16::h16_from(v);
    a.wr[L][(size_t)n * kp + k] = h;
    a.wtr[L][(size_t)k * np + n] = h;
    a.wf[L][((((size_t)(n >> 5) * (kp >> 4) + (k >> 4)) * 2 + ((k >> 3) & 1)) * 32 + (n & 31)) * 8 + (k & 7)] = h;
    a.wtf[L][((((size_t)(k >> 5) * (np >> 4) + (n >> 4)) * 2 + ((n >> 3) & 1)) * 32 + (k & 31)) * 8 + (n & 7)] = h;
}
}  // namespace

extern "C" int GAGS_DEC(gags_decoder_pack_layers)(int n_layers, const int *co, const int *ci, const float *const *w, const float *const *b,
                                        void *const *w_bf16, void *const *wt_bf16, void *const *w_frag, void *const *wt_frag,
                                        float *const *bias_pad, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_layers <= 0 || n_layers > PACK_MAX_LAYERS || !co || !ci || !w || !b || !w_bf16 || !wt_bf16 || !w_frag || !wt_frag || !bias_pad)
        return GAGS_EINVAL;
    PackArgs a;
    int most = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (co[i] <= 0 || ci[i] <= 0 || !w[i] || !b[i] || !w_bf16[i] || !wt_bf16[i] || !w_frag[i] || !wt_frag[i] || !bias_pad[i])
            return GAGS_EINVAL;
        a.co[i] = co[i]; a.ci[i] = ci[i]; a.w[i] = w[i]; a.b[i] = b[i];
        a.wr[i] = (unsigned short *)w_bf16[i]; a.wtr[i] = (unsigned short *)wt_bf16[i];
        a.wf[i] = (unsigned short *)w_frag[i]; a.wtf[i] = (unsigned short *)wt_frag[i]; a.bp[i] = bias_pad[i];
        const int np = (co[i] + 31) / 32 * 32, kp = (ci[i] + 31) / 32 * 32;
        most = np * kp > most ? np * kp : most;
    }
    hipLaunchKernelGGL(pack_layers_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n_layers), dim3(256), 0, (hipStream_t)stream, a);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int GAGS_DEC(gags_decoder_pack_input)(int64_t n_pix, int c, int c_pad, const float *x, void *y_bf16, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || c_pad < c || c_pad % 32 != 0 || (n_pix > 0 && (!x || !y_bf16))) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(to_bf16_pad_kernel, dim3((unsigned)((n_pix * c_pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_pix, c, c_pad, x, (unsigned short *)y_bf16);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int GAGS_DEC(gags_decoder_layer)(int64_t n_pix, int n_out, int k_in, const void *a1, const void *a2, const void *w,
                                  const float *bias, int relu, const void *mask_src, const void *residual, void *y_bf16,
                                  void *y_premask_bf16, float *y_f32, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || n_out <= 0 || k_in <= 0 || k_in % 32 != 0 || n_out % 8 != 0 || n_pix * k_in >= ((int64_t)1 << 31) || !a1 || !w ||
        (!y_bf16 && !y_f32 && !y_premask_bf16))
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    GemmArgs g;
    g.A1 = (const unsigned short *)a1; g.A2 = (const unsigned short *)a2; g.W = (const unsigned short *)w; g.bias = bias;
    g.mask_src = (const unsigned short *)mask_src; g.E = (const unsigned short *)residual;
    g.Y = (unsigned short *)y_bf16; g.Ypre = (unsigned short *)y_premask_bf16; g.Yf = y_f32; g.P = n_pix; g.N = n_out; g.K = k_in; g.relu = relu;
    const unsigned p_tiles = (unsigned)((n_pix + TM - 1) / TM);
    if (n_out > 128) {
        const int n_tiles = (n_out + 255) / 256;
        const unsigned grid = n_tiles == 1 ? p_tiles : (p_tiles + 7) / 8 * 8 * n_tiles;
        if (a2) hipLaunchKernelGGL((gemm_bf16_kernel<4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n_tiles, p_tiles);
        else hipLaunchKernelGGL((gemm_bf16_kernel<4, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n_tiles, p_tiles);
    } else {
        if (a2) hipLaunchKernelGGL((gemm_bf16_kernel<2, true>), dim3(p_tiles), dim3(256), 0, (hipStream_t)stream, g, 1, p_tiles);
        else hipLaunchKernelGGL((gemm_bf16_kernel<2, false>), dim3(p_tiles), dim3(256), 0, (hipStream_t)stream, g, 1, p_tiles);
    }
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int GAGS_DEC(gags_decoder_head)(int64_t n_pix, int c, int ld, int mode, const float *x, float *out, int layout, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || ld < c || (mode != 0 && mode != 1) || (n_pix > 0 && (!x || !out))) return GAGS_EINVAL;
    if (layout != 0 && !(layout == 1 && c % 4 == 0 && ld <= 512 && ld % 32 == 0)) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    if (layout == 1) {
        hipLaunchKernelGGL(head_pm_kernel, dim3((unsigned)((n_pix + HP - 1) / HP)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld, mode,
                           x, out);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (c <= HS_MAXC && ld % 8 == 0 && ld >= 8)
        hipLaunchKernelGGL(head_small_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld, mode,
                           x, out);
    else if (ld <= 512 && ld % 32 == 0)
        hipLaunchKernelGGL(head_fast_kernel, dim3((unsigned)((n_pix + HP - 1) / HP)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld,
                           mode, x, out);
    else
        hipLaunchKernelGGL(head_kernel, dim3((unsigned)((n_pix + 63) / 64)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld, mode, x, out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

namespace {
// which kernel serves a shape, and how many partial matrices (pixel chunks) it leaves
enum { WG_NARROW = 0, WG_256 = 1, WG_GENERAL = 2 };
inline bool narrow_shape(int n_out, int k_in)
{
    if (n_out % 32 || k_in % 32) return false;
    const int nb = n_out / 32, kb = k_in / 32;
    return (nb == 1 && (kb == 1 || kb == 2 || kb == 4 || kb == 8)) || (nb == 2 && (kb == 1 || kb == 2 || kb == 4)) ||
           (nb == 4 && (kb == 1 || kb == 2)) || (nb == 8 && kb == 1);
}
inline int64_t w256_chunk(int64_t n_pix) { return ((n_pix + 255) / 256 + W2P - 1) / W2P * W2P; }
inline int wgrad_plan(int64_t n_pix, int n_out, int k_in, int64_t &parts)
{
    if (narrow_shape(n_out, k_in)) { parts = narrow_parts(n_pix); return WG_NARROW; }
    if (k_in == 256 && n_out % 128 == 0) { const int64_t c = w256_chunk(n_pix); parts = (n_pix + c - 1) / c; return WG_256; }
    parts = (n_pix + WCHUNK - 1) / WCHUNK;
    return WG_GENERAL;
}
}  // namespace

extern "C" int64_t GAGS_DEC(gags_decoder_wgrad_scratch_bytes)(int64_t n_pix, int n_out, int k_in)
{
    if (n_pix <= 0 || n_out <= 0 || k_in <= 0) return 0;
    int64_t parts;
    wgrad_plan(n_pix, n_out, k_in, parts);
    return (parts * ((int64_t)n_out * k_in + n_out) * 4 + 255) / 256 * 256;
}

namespace {
int wgrad_impl(int64_t n_pix, int n_out, int k_in, const void *dz, const void *a1, const void *a2, float *d_w, float *d_b, int co,
               int ci, const float *out_scale, void *scratch, int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || n_out <= 0 || k_in <= 0 || n_out % 16 != 0 || k_in % 16 != 0 || !dz || !a1 || !d_w || co <= 0 || ci <= 0 ||
        co > n_out || ci > k_in)
        return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_pix == 0) {
        if (hipMemsetAsync(d_w, 0, sizeof(float) * (size_t)co * ci, st) != hipSuccess) return GAGS_ELAUNCH;
        if (d_b && hipMemsetAsync(d_b, 0, sizeof(float) * (size_t)co, st) != hipSuccess) return GAGS_ELAUNCH;
        return GAGS_OK;
    }
    if (!scratch || scratch_bytes < gags_decoder_wgrad_scratch_bytes(n_pix, n_out, k_in)) return GAGS_ESCRATCH;
    int64_t parts;
    const int plan = wgrad_plan(n_pix, n_out, k_in, parts);
    // every pixel chunk leaves a partial matrix (and bias row) in scratch; they are summed in chunk order: no atomics
    float *pw = (float *)scratch, *pb = d_b ? pw + (size_t)parts * n_out * k_in : nullptr;
    if (plan == WG_NARROW) {
        const int nb = n_out / 32, kb = k_in / 32;
        if (false) {}
#define GAGS_NARROW(NB_, KB_) else if (nb == NB_ && kb == KB_) launch_wgrad_narrow<NB_, KB_>(n_pix, dz, a1, a2, pw, pb, st);
        GAGS_NARROW(1, 1) GAGS_NARROW(2, 1) GAGS_NARROW(1, 2) GAGS_NARROW(2, 2) GAGS_NARROW(4, 1) GAGS_NARROW(1, 4)
        GAGS_NARROW(4, 2) GAGS_NARROW(2, 4) GAGS_NARROW(8, 1) GAGS_NARROW(1, 8)
#undef GAGS_NARROW
    } else if (plan == WG_256) {
        // one workgroup per n tile and pixel chunk, two per CU: 512 chunks of whole 32-pixel steps
        const int n_tiles = n_out / 128;
        const int64_t chunk = w256_chunk(n_pix);
        const unsigned n_chunks = (unsigned)parts;
        const unsigned grid = n_tiles == 1 ? n_chunks : (n_chunks + 7) / 8 * 8 * n_tiles;
        if (a2)
            hipLaunchKernelGGL(wgrad256_kernel<true>, dim3(grid), dim3(256), 0, st, n_pix, n_out, (const unsigned short *)dz,
                               (const unsigned short *)a1, (const unsigned short *)a2, pw, pb, chunk, n_tiles, n_chunks);
        else
            hipLaunchKernelGGL(wgrad256_kernel<false>, dim3(grid), dim3(256), 0, st, n_pix, n_out, (const unsigned short *)dz,
                               (const unsigned short *)a1, (const unsigned short *)a2, pw, pb, chunk, n_tiles, n_chunks);
    } else {
        hipLaunchKernelGGL(wgrad_bf16_kernel, dim3((unsigned)parts, (unsigned)((n_out + 127) / 128), (unsigned)((k_in + 127) / 128)),
                           dim3(256), 0, st, n_pix, n_out, k_in, (const unsigned short *)dz, (const unsigned short *)a1,
                           (const unsigned short *)a2, pw, pb);
    }
    // one launch sums the matrix and the bias row, in the caller's [co, ci] shape
    const unsigned wb = (unsigned)(((int64_t)co * ci + 63) / 64), bb = d_b ? (unsigned)((co + 63) / 64) : 0u;
    hipLaunchKernelGGL(sum_wparts_out_kernel, dim3(wb + bb), dim3(1024), 0, st, (int)parts, n_out, k_in, co, ci, wb, pw, pb, d_w, d_b,
                       out_scale);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
}  // namespace

extern "C" int GAGS_DEC(gags_decoder_wgrad)(int64_t n_pix, int n_out, int k_in, const void *dz, const void *a1, const void *a2,
                                  float *d_w, float *d_b, void *scratch, int64_t scratch_bytes, void *stream)
{
    return wgrad_impl(n_pix, n_out, k_in, dz, a1, a2, d_w, d_b, n_out, k_in, nullptr, scratch, scratch_bytes, stream);
}

extern "C" int GAGS_DEC(gags_decoder_wgrad_out)(int64_t n_pix, int n_out, int k_in, const void *dz, const void *a1, const void *a2,
                                      float *d_w, float *d_b, int co, int ci, const float *out_scale, void *scratch,
                                      int64_t scratch_bytes, void *stream)
{
    return wgrad_impl(n_pix, n_out, k_in, dz, a1, a2, d_w, d_b, co, ci, out_scale, scratch, scratch_bytes, stream);
}

extern "C" int GAGS_DEC(gags_decoder_head_bwd)(int64_t n_pix, int c, int ld, int mode, const float *x, const float *g, void *dz_bf16,
                                     int layout, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || ld < c || (mode != 0 && mode != 1) || (n_pix > 0 && (!x || !g || !dz_bf16))) return GAGS_EINVAL;
    if (layout != 0 && !(layout == 1 && c % 4 == 0 && ld <= 512 && ld % 32 == 0)) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    if (layout == 1) {
        hipLaunchKernelGGL(head_bwd_pm_kernel, dim3((unsigned)((n_pix + HP - 1) / HP)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld,
                           mode, x, g, (unsigned short *)dz_bf16);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (c <= HS_MAXC && ld % 8 == 0 && ld >= 8)
        hipLaunchKernelGGL(head_bwd_small_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld,
                           mode, x, g, (unsigned short *)dz_bf16);
    else if (ld <= 512 && ld % 32 == 0)
        hipLaunchKernelGGL(head_bwd_fast_kernel, dim3((unsigned)((n_pix + HP - 1) / HP)), dim3(256), 0, (hipStream_t)stream, n_pix, c,
                           ld, mode, x, g, (unsigned short *)dz_bf16);
    else
        hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((n_pix + 63) / 64)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld, mode,
                           x, g, (unsigned short *)dz_bf16);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int GAGS_DEC(gags_softmax_head_bwd_y)(int64_t n_pix, int c, int ld, const float *y, const float *g, void *dz_bf16, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || c > HS_MAXC || ld < 8 || ld % 8 != 0 || (n_pix > 0 && (!y || !g || !dz_bf16))) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(softmax_bwd_small_y_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ld,
                       y, g, (unsigned short *)dz_bf16);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int GAGS_DEC(gags_decoder_unpack_grad)(int64_t n_pix, int c, int ld, const void *x_bf16, float *y, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || ld < c || (n_pix > 0 && (!x_bf16 || !y))) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(unpack_f32_kernel, dim3((unsigned)((n_pix * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_pix, c,
                       ld, (const unsigned short *)x_bf16, y);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
