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
#include <stdlib.h>
#include "common.h"
#include "half16.h"
#include "gags_next.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float ff32x2_t __attribute__((ext_vector_type(2)));
using gags_h16::h16_mfma;
__device__ __forceinline__ unsigned fpack(float lo, float hi) { return gags_h16::h16_pack_sat(lo, hi); }  // (both kernels set the mode)
__device__ __forceinline__ float flo(unsigned u) { return gags_h16::h16_lo(u); }
__device__ __forceinline__ float fhi(unsigned u) { return gags_h16::h16_hi(u); }

#ifndef GAGS_ABL
#define GAGS_ABL 0  // timing-only ablations of the fused kernels (tools/chain_bench.py; results are WRONG with any of them): 1 no HBM
                    // stores (64 masks / 128 tiles / 256 logits alone), 2 no weight refills, 4 no MFMAs, 16 no B reads, 32 plain stores, 512 tile stores into 2 MB per tensor
#endif
constexpr int FT = 64;        // pixels per tile and group of four waves; a workgroup holds PH such groups (tile = 64 PH pixels)
constexpr int FH = 256;       // hidden width
constexpr int FLD = FH + 8;   // LDS row pitch in bf16 (528 B: 16-byte aligned rows, consecutive rows 4 banks apart)
constexpr int FPD = 8;        // weight fragments are requested this many K-steps ahead

typedef unsigned short (*Tile)[FLD];

// the tiles' HBM stores are non-temporal (`nt`): 10.6 GB per 1080p forward that nothing reads before the kernel ends stream past
// the L2 the weight fragments live in (round 6: -2 %; GAGS_ABL & 32 restores plain stores)
typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(uint4 *dst, uint4 v)
{
    if (!(GAGS_ABL & 32)) __builtin_nontemporal_store(nt_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4 *>(dst));
    else *dst = v;
}
__device__ __forceinline__ void stream_store(float4 *dst, float4 v)
{
    if (!(GAGS_ABL & 32)) __builtin_nontemporal_store(nt_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f32x4 *>(dst));
    else *dst = v;
}

// Weights arrive in MFMA-FRAGMENT order (gags_amd/decoders.py: _frag_layout): Wf[n_tile][k_step][lane][8] with lane =
// 32 kh + n and the eight values k = 16 k_step + 8 kh + 0..7 of row 32 n_tile + n -- the A operand of one MFMA is one
// contiguous, fully coalesced kilobyte, and every weight byte travels from L2 exactly once per tile.  (Read from the
// row-major matrix, a fragment is 32 row pieces of 32 bytes, one per 128-byte line: the eight waves of a CU keep 64 KB of
// such lines in flight, the L1 thrashes, and every line is fetched four times -- measured 5.8 ms instead of 2.)
// acc[i][j] (i: 32 channels of n-tile nt0 + i, j: 32 pixels 32 j..) = sum over k-steps ks0 .. ks0 + ksteps - 1 of W[n][k] in[p][k]
struct WFrag {
    bf16x8 a0[FPD], a1[FPD];
};

// the first FPD K-steps' weight fragments of a layer: issued BEFORE the epilogue of the layer above (and before its tile
// stores), so that their L2 latency hides behind that epilogue instead of opening every layer
__device__ __forceinline__ void wprefetch(WFrag &f, const unsigned short *__restrict__ Wf, int ksteps_total, int nt0, int ks0, int ksteps,
                                          int lane)
{
    const unsigned short *w0 = Wf + ((size_t)nt0 * ksteps_total + ks0) * 512 + lane * 8;
    const unsigned short *w1 = w0 + (size_t)ksteps_total * 512;
#pragma unroll
    for (int q = 0; q < FPD; ++q) {
        const int ks = min(q, ksteps - 1);
        f.a0[q] = *reinterpret_cast<const bf16x8 *>(w0 + 512 * ks);
        f.a1[q] = *reinterpret_cast<const bf16x8 *>(w1 + 512 * ks);
    }
}

__device__ __forceinline__ void layer_main(f32x16 (&acc)[2][2], WFrag &f, const unsigned short *__restrict__ Wf, int ksteps_total, int nt0,
                                           int ks0, int ksteps, Tile in, int lane, bool zero, int poff)
{
    if (zero) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const unsigned short *w0 = Wf + ((size_t)nt0 * ksteps_total + ks0) * 512 + lane * 8;
    const unsigned short *w1 = w0 + (size_t)ksteps_total * 512;
    for (int k0 = 0; k0 < ksteps; k0 += FPD) {
#pragma unroll
        for (int q = 0; q < FPD; ++q) {
            const int ks = k0 + q;
            const bf16x8 c0 = f.a0[q], c1 = f.a1[q];
            const int kn = min(ks + FPD, ksteps - 1);  // (past the end: a harmless re-read)
            if (!(GAGS_ABL & 2)) {
            f.a0[q] = *reinterpret_cast<const bf16x8 *>(w0 + 512 * kn);
            f.a1[q] = *reinterpret_cast<const bf16x8 *>(w1 + 512 * kn);
            }
            if (ks < ksteps) {  // (uniform)
                bf16x8 b0 = c0, b1 = c1;
                if (!(GAGS_ABL & 16)) {
                b0 = *reinterpret_cast<const bf16x8 *>(&in[poff + (lane & 31)][16 * ks + 8 * (lane >> 5)]);
                b1 = *reinterpret_cast<const bf16x8 *>(&in[poff + 32 + (lane & 31)][16 * ks + 8 * (lane >> 5)]);
                }
                if (!(GAGS_ABL & 4)) {
                acc[0][0] = h16_mfma(c0, b0, acc[0][0]);
                acc[0][1] = h16_mfma(c0, b1, acc[0][1]);
                acc[1][0] = h16_mfma(c1, b0, acc[1][0]);
                acc[1][1] = h16_mfma(c1, b1, acc[1][1]);
                } else { acc[0][0][0] += (float)b0[0] + (float)c0[0]; acc[1][1][0] += (float)b1[0] + (float)c1[0]; }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch FPD steps ahead (the scheduler sinks loads to their use)
        }
    }
}

// accumulator element (i, j, 4 g + e) of lane (p = lane & 31, h = lane >> 5): pixel 32 j + p, channel n_base + 32 i + 8 g + 4 h + e
// hidden-layer epilogue: out[p][n] = bf16(relu(acc + bias[n])); and, for the backward, the ReLU decisions as BITS:
// word n / 32 of a pixel holds [out > 0] of 32 channels (channel 8 g + 4 h + e of the word at bit 8 h + 2 g + (e >> 1) + 16 (e & 1));
// words are stored word-major inside 64-pixel groups, mask[(group * 8 + word) * 64 + pixel % 64]: both private to these two kernels -- 32 bytes per pixel and layer instead of the 512-byte activation row
// the input-gradient chain would otherwise re-read just for its sign.
__device__ __forceinline__ void epilogue_hidden(const f32x16 (&acc)[2][2], const float *__restrict__ bias, int n_base, Tile out,
                                                int lane, unsigned *__restrict__ mask, int64_t p0, int64_t P, int poff)
{
    // 64 values per lane and layer: this epilogue costs more issue slots than the layer's MFMAs (timestamps, round 3), so
    // it runs on packed instructions: v_pk_add_f32 (bias), v_cvt_pk_bf16_f32, then the ReLU on the PAIR -- v_pk_max_i16
    // with 0 (a negative bf16 is a negative int16; rounding first and clamping second gives the same bits) -- and the
    // ReLU decisions from v_pk_min_u16(pair, 1).  (Written as inline asm: the builtins' IEEE semantics expand to compares.)
    const int p = lane & 31, h = lane >> 5;
    unsigned bits[2][2] = {{0u, 0u}, {0u, 0u}};
    const unsigned one2 = 0x00010001u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n_base + 32 * i + 8 * g + 4 * h;
            const float4 b = *reinterpret_cast<const float4 *>(bias + n);
            const ff32x2_t b01 = {b.x, b.y}, b23 = {b.z, b.w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const ff32x2_t a01 = {acc[i][j][4 * g], acc[i][j][4 * g + 1]}, a23 = {acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                const ff32x2_t v01 = a01 + b01, v23 = a23 + b23;
                unsigned u0 = gags_h16::h16_pack_sat(v01[0], v01[1]);
                unsigned u1 = gags_h16::h16_pack_sat(v23[0], v23[1]);
                unsigned m0, m1;
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(u0) : "v"(u0));
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(u1) : "v"(u1));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(m0) : "v"(u0), "v"(one2));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(m1) : "v"(u1), "v"(one2));
                *reinterpret_cast<uint2 *>(&out[poff + 32 * j + p][n]) = make_uint2(u0, u1);
                // m0 / m1 carry the decisions of elements (0, 1) / (2, 3) at bits 0 and 16: shifted in as they are, two
                // v_lshl_or per group (round 6; assembling a nibble per group first cost eight) -- the word's layout is private to
                // the two fused kernels: element e of group g of half h at bit 8 h + 2 g + (e >> 1) + 16 (e & 1)
                bits[i][j] |= m0 << (2 * g);
                bits[i][j] |= m1 << (2 * g + 1);
            }
        }
    if (mask) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned mine = bits[i][j] << (8 * h);  // this half-wave's bits sit at 8 h + 0..7 and 8 h + 16..23
                const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                const unsigned word = sw[0] | sw[1];
                const int64_t pg = p0 + poff + 32 * j + p;
                // word-major inside a 64-pixel group (round 6): the 32 lanes of a store write 128 contiguous bytes (pixel-major,
                // 8 words per pixel, they wrote 4 bytes every 32: the masks were 5 % of the kernel's bytes and 5 % of its time)
                if (h == 0 && pg < P && !(GAGS_ABL & (1 | 64)))
                    mask[(((p0 + poff) >> 6) * 8 + (n_base >> 5) + i) * 64 + 32 * j + p] = word;
            }
    }
}

// the workgroup copies a [64][256] bf16 tile from LDS to its rows of a pixel-major tensor (16 bytes per lane, whole rows)
template <int NT>
__device__ __forceinline__ void store_tile(unsigned short *__restrict__ dst, int64_t p0, int64_t P, Tile src, int tid)
{
    if (!dst || (GAGS_ABL & (1 | 128))) return;
    if (GAGS_ABL & 512) p0 = (p0 / 64 % 64) * 64;  // (every tile's stores land in the same 2 MB of its tensor: they stay in L2)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + NT * q, row = id >> 5, c = (id & 31) * 8;
        if (p0 + row < P) stream_store(reinterpret_cast<uint4 *>(dst + (size_t)(p0 + row) * FH + c), *reinterpret_cast<const uint4 *>(&src[row][c]));
    }
}

// dst[p][:] = bf16(dst + add) element-wise (fp32 add, one rounding: what gags_decoder_layer does with two sources)
// dst += add in LDS, and the sum's tile to its place in a pixel-major [P, 256] tensor: the residual sums x1 + x2 and
// x3 + x4 are what layers 3 and 6 read AND what their weight gradients contract -- kept instead of x2 / x4 (whose ReLU
// decisions travel as bit masks), those weight gradients read one tensor, not two
template <int NT>
__device__ __forceinline__ void add_store_tile(Tile dst, Tile add, unsigned short *__restrict__ keep, int64_t p0, int64_t P, int tid)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + NT * q, row = id >> 5, c = (id & 31) * 8;
        const uint4 x = *reinterpret_cast<const uint4 *>(&dst[row][c]), y = *reinterpret_cast<const uint4 *>(&add[row][c]);
        const uint4 sum = make_uint4(fpack(flo(x.x) + flo(y.x), fhi(x.x) + fhi(y.x)), fpack(flo(x.y) + flo(y.y), fhi(x.y) + fhi(y.y)),
                                     fpack(flo(x.z) + flo(y.z), fhi(x.z) + fhi(y.z)), fpack(flo(x.w) + flo(y.w), fhi(x.w) + fhi(y.w)));
        *reinterpret_cast<uint4 *>(&dst[row][c]) = sum;
        if (keep && p0 + row < P && !(GAGS_ABL & (1 | 128))) stream_store(reinterpret_cast<uint4 *>(keep + (size_t)(p0 + row) * FH + c), sum);
    }
}
template <int NT>
__device__ __forceinline__ void add_tile(Tile dst, Tile add, int tid)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + NT * q, row = id >> 5, c = (id & 31) * 8;
        const uint4 x = *reinterpret_cast<const uint4 *>(&dst[row][c]), y = *reinterpret_cast<const uint4 *>(&add[row][c]);
        *reinterpret_cast<uint4 *>(&dst[row][c]) =
            make_uint4(fpack(flo(x.x) + flo(y.x), fhi(x.x) + fhi(y.x)), fpack(flo(x.y) + flo(y.y), fhi(x.y) + fhi(y.y)),
                       fpack(flo(x.z) + flo(y.z), fhi(x.z) + fhi(y.z)), fpack(flo(x.w) + flo(y.w), fhi(x.w) + fhi(y.w)));
    }
}

struct FwdArgs {
    const float *x;              // [P, c_in] fp32 pixel-major (the rasterizer's own output), c_in <= 32
    const unsigned short *W[9];  // bf16 in MFMA-fragment order (see layer_mma) of the padded [256, 32], 7 x [256, 256], [n_last, 256]
    const float *b[9];
    unsigned short *act[9];      // a0 [P, 32], then x1, t1, x2, x3, t4, x4, t6, t7 [P, 256]: null = not kept (inference)
    unsigned *mask[9];           // [1..8]: ReLU bit masks [P, 8] of the same activations (null = not kept)
    float *logits;               // [P, n_last] fp32
    int64_t P;
    int c_in, n_last;            // n_last % 256 == 0
};

template <int PH>
__global__ __launch_bounds__(256 * PH, 2 / PH) void decoder_fwd_fused_kernel(FwdArgs a)
{
    gags_h16::h16_saturate_mode();  // (f16 tier: conversions saturate in hardware; half16.h)
    __shared__ __attribute__((aligned(16))) unsigned short bufA[FT * PH][FLD];
    __shared__ __attribute__((aligned(16))) unsigned short bufB[FT * PH][FLD];
    constexpr int TP = FT * PH, NT = 256 * PH;  // pixels per tile, threads: PH groups of four waves, 64 pixels each
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, poff = FT * (tid >> 8);
    const int64_t p0 = (int64_t)blockIdx.x * TP;
    const int n_base = 64 * wave;
    // the eight hidden layers' biases, staged once: read from global memory inside an epilogue they queue up BEHIND the
    // next layer's prefetched weight fragments (loads return in order) and the epilogue stands still for that long
    __shared__ __attribute__((aligned(16))) float bias_s[8][FH];
    {
        float bv[8 * FH / NT];
#pragma unroll
        for (int q = 0; q < 8 * FH / NT; ++q) { const int e = tid + NT * q; bv[q] = a.b[e >> 8][e & (FH - 1)]; }
#pragma unroll
        for (int q = 0; q < 8 * FH / NT; ++q) { const int e = tid + NT * q; bias_s[e >> 8][e & (FH - 1)] = bv[q]; }
    }
    f32x16 acc[2][2];
    WFrag wf;
    wprefetch(wf, a.W[0], 2, 2 * wave, 0, 2, lane);

    // input tile -> bufB[p][0..31] bf16 (zero-padded), also kept as a0 for the first layer's weight gradient
    {   // (all eight requests first, addresses clamped: a load inside the bounds check is waited for on the spot, and the
        // eight round trips of a thread queue up behind one another)
        float xv[TP * 32 / NT];
#pragma unroll
        for (int q = 0; q < TP * 32 / NT; ++q) {
            const int e = tid + NT * q, row = e >> 5, c = e & 31;
            xv[q] = a.x[min(p0 + row, a.P - 1) * a.c_in + min(c, a.c_in - 1)];
        }
#pragma unroll
        for (int q = 0; q < TP * 32 / NT; ++q) {
            const int e = tid + NT * q, row = e >> 5, c = e & 31;
            const int64_t p = p0 + row;
            const float v = (p < a.P && c < a.c_in) ? xv[q] : 0.f;
            const unsigned short hv = (unsigned short)(fpack(v, 0.f) & 0xffffu);
            bufB[row][c] = hv;
            if (a.act[0] && p < a.P) a.act[0][p * 32 + c] = hv;
        }
    }
    __syncthreads();
    // L0: a0 (B) -> x1 (A)
    layer_main(acc, wf, a.W[0], 2, 2 * wave, 0, 2, bufB, lane, true, poff);
    wprefetch(wf, a.W[1], 16, 2 * wave, 0, 16, lane);
    epilogue_hidden(acc, bias_s[0], n_base, bufA, lane, a.mask[1], p0, a.P, poff);
    __syncthreads();
    store_tile<NT>(a.act[1], p0, a.P, bufA, tid);
    // L1: x1 (A) -> t1 (B)
    layer_main(acc, wf, a.W[1], 16, 2 * wave, 0, 16, bufA, lane, true, poff);
    wprefetch(wf, a.W[2], 16, 2 * wave, 0, 16, lane);
    epilogue_hidden(acc, bias_s[1], n_base, bufB, lane, a.mask[2], p0, a.P, poff);
    __syncthreads();
    store_tile<NT>(a.act[2], p0, a.P, bufB, tid);
    // L2: t1 (B) -> x2, written over t1 once every wave is done reading it; then A = x1 + x2
    layer_main(acc, wf, a.W[2], 16, 2 * wave, 0, 16, bufB, lane, true, poff);
    wprefetch(wf, a.W[3], 16, 2 * wave, 0, 16, lane);
    __syncthreads();
    epilogue_hidden(acc, bias_s[2], n_base, bufB, lane, a.mask[3], p0, a.P, poff);
    __syncthreads();
    add_store_tile<NT>(bufA, bufB, a.act[3], p0, a.P, tid);  // kept: x1 + x2, the input of layer 3 (what its weight gradient contracts)
    __syncthreads();
    // L3: x1 + x2 (A) -> x3 (B)
    layer_main(acc, wf, a.W[3], 16, 2 * wave, 0, 16, bufA, lane, true, poff);
    wprefetch(wf, a.W[4], 16, 2 * wave, 0, 16, lane);
    epilogue_hidden(acc, bias_s[3], n_base, bufB, lane, a.mask[4], p0, a.P, poff);
    __syncthreads();
    store_tile<NT>(a.act[4], p0, a.P, bufB, tid);
    // L4: x3 (B) -> t4 (A)
    layer_main(acc, wf, a.W[4], 16, 2 * wave, 0, 16, bufB, lane, true, poff);
    wprefetch(wf, a.W[5], 16, 2 * wave, 0, 16, lane);
    epilogue_hidden(acc, bias_s[4], n_base, bufA, lane, a.mask[5], p0, a.P, poff);
    __syncthreads();
    store_tile<NT>(a.act[5], p0, a.P, bufA, tid);
    // L5: t4 (A) -> x4 over t4; then B = x3 + x4
    layer_main(acc, wf, a.W[5], 16, 2 * wave, 0, 16, bufA, lane, true, poff);
    wprefetch(wf, a.W[6], 16, 2 * wave, 0, 16, lane);
    __syncthreads();
    epilogue_hidden(acc, bias_s[5], n_base, bufA, lane, a.mask[6], p0, a.P, poff);
    __syncthreads();
    add_store_tile<NT>(bufB, bufA, a.act[6], p0, a.P, tid);  // kept: x3 + x4, the input of layer 6
    __syncthreads();
    // L6: x3 + x4 (B) -> t6 (A)
    layer_main(acc, wf, a.W[6], 16, 2 * wave, 0, 16, bufB, lane, true, poff);
    wprefetch(wf, a.W[7], 16, 2 * wave, 0, 16, lane);
    epilogue_hidden(acc, bias_s[6], n_base, bufA, lane, a.mask[7], p0, a.P, poff);
    __syncthreads();
    store_tile<NT>(a.act[7], p0, a.P, bufA, tid);
    // L7: t6 (A) -> t7 (B)
    layer_main(acc, wf, a.W[7], 16, 2 * wave, 0, 16, bufA, lane, true, poff);
    epilogue_hidden(acc, bias_s[7], n_base, bufB, lane, a.mask[8], p0, a.P, poff);
    __syncthreads();
    store_tile<NT>(a.act[8], p0, a.P, bufB, tid);
    // L8: t7 (B) -> fp32 logits [P, n_last], 256 channels per pass.  Stored straight from the accumulators every
    // instruction would scatter 32-byte pieces 4 n_last bytes apart and HBM sees partial lines (csrc/decoder.hip measured
    // 2.4 x the bytes for that pattern); instead the two halves of a pass go through bufA (free by now: 64 pixels x 128
    // fp32 channels) and leave as whole 512-byte rows, 16 bytes per lane.
    float (*patch)[FLD / 2] = reinterpret_cast<float (*)[FLD / 2]>(&bufA[0][0]);  // [TP][132] floats, same 528-byte pitch
    const int p = lane & 31, h = lane >> 5;
    for (int nb = 0; nb < a.n_last; nb += FH) {
        wprefetch(wf, a.W[8], 16, nb / 32 + 2 * wave, 0, 16, lane);
        layer_main(acc, wf, a.W[8], 16, nb / 32 + 2 * wave, 0, 16, bufB, lane, true, poff);
        for (int half = 0; half < 2; ++half) {
            if ((wave >> 1) == half) {  // waves 2 half, 2 half + 1 hold channels 128 half .. + 127 of the pass
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = (n_base & 127) + 32 * i + 8 * g + 4 * h;
                        const float4 b = *reinterpret_cast<const float4 *>(a.b[8] + nb + 128 * half + nl);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            *reinterpret_cast<float4 *>(&patch[poff + 32 * j + p][nl]) =
                                make_float4(acc[i][j][4 * g] + b.x, acc[i][j][4 * g + 1] + b.y, acc[i][j][4 * g + 2] + b.z,
                                            acc[i][j][4 * g + 3] + b.w);
                    }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int id = tid + NT * q, row = id >> 5, c = (id & 31) * 4;
                if (p0 + row < a.P && !(GAGS_ABL & (1 | 256)))
                    stream_store(reinterpret_cast<float4 *>(a.logits + (size_t)(p0 + row) * a.n_last + nb + 128 * half + c),
                                 *reinterpret_cast<const float4 *>(&patch[row][c]));
            }
            __syncthreads();
        }
    }
}

// ---- backward: the nine input-gradient GEMMs of CNN_decoder in one kernel --------------------------------------------
//   dz7 = (W8^T dz8) * [t7 > 0]      dz6 = (W7^T dz7) * [t6 > 0]      g36 = W6^T dz6,  dz5 = g36 * [x4 > 0]
//   dz4 = (W5^T dz5) * [t4 > 0]      dz3 = (W4^T dz4 + g36) * [x3 > 0]   g13 = W3^T dz3,  dz2 = g13 * [x2 > 0]
//   dz1 = (W2^T dz2) * [t1 > 0]      dz0 = (W1^T dz1 + g13) * [x1 > 0]   d x = W0^T dz0
// Per 64-pixel tile the dz of the layer above sits in one LDS buffer and the result goes to the other; the ReLU decisions
// come as the forward's bit masks (four words per lane and layer, requested before the K loop), and the two skip
// gradients never leave the registers of the lanes that produced them (the accumulator -> (pixel, channel) mapping is the
// same in every layer: 32 VGPRs of packed bf16).  Every dz leaves once, for the weight gradients.  Arithmetic and
// roundings as gags_decoder_layer with mask_src / residual / y_premask: bit-identical to the layer-by-layer backward.
struct BwdArgs {
    const unsigned short *dz8;    // [P, n_last] bf16
    const unsigned short *Wt[9];  // W_i^T, bf16 in MFMA-fragment order: [32, 256], 7 x [256, 256], [256, n_last]
    const unsigned *mask[9];      // [1..8]: ReLU bit masks [P, 8] of x1, t1, x2, x3, t4, x4, t6, t7
    unsigned short *dz[8];        // dz0 .. dz7 [P, 256] out
    float *gin;                   // [P, c_in] fp32 out, or null
    const float *gin_scale;       // optional device scalar gin is multiplied by (the f16 tier's power of two: exact)
    int64_t P;
    int c_in, n_last;
};

template <int NT>
__device__ __forceinline__ void fetch_tile(uint4 (&r)[8], const unsigned short *__restrict__ src, int ld, int col0, int64_t p0,
                                           int64_t P, int tid)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + NT * q, row = id >> 5, c = (id & 31) * 8;
        const int64_t pg = min(p0 + row, P - 1);  // clamped: unconditional loads; rows past the image are never stored
        r[q] = *reinterpret_cast<const uint4 *>(src + (size_t)pg * ld + col0 + c);
    }
}
template <int NT>
__device__ __forceinline__ void commit_tile(Tile dst, const uint4 (&r)[8], int tid)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int id = tid + NT * q, row = id >> 5, c = (id & 31) * 8;
        *reinterpret_cast<uint4 *>(&dst[row][c]) = r[q];
    }
}

// the four mask words of this lane: pixels p0 + 32 j + p, channels 32 (2 wave + i) .. + 31
__device__ __forceinline__ void fetch_mask(unsigned (&mw)[2][2], const unsigned *__restrict__ mask, int64_t p0, int64_t P, int wave, int lane, int poff)
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mw[i][j] = mask[(((p0 + poff) >> 6) * 8 + 2 * wave + i) * 64 + 32 * j + (lane & 31)];  // (padded to whole groups)
}

// out[p][n] = bf16(acc (+ res)) * [mask bit];  KEEP: the value before the mask stays in `keep` (packed bf16: a skip gradient);
// ADD: `res` (a kept skip gradient) is added before the rounding.
template <bool KEEP, bool ADD>
__device__ __forceinline__ void epilogue_dgrad(const f32x16 (&acc)[2][2], int n_base, const unsigned (&mw)[2][2], Tile out,
                                               uint2 (&keep)[2][4][2], int lane, int poff)
{
    const int p = lane & 31, h = lane >> 5;
    unsigned wh[2][2];  // this half-wave's bits of the four words, at 2 g + (e >> 1) + 16 (e & 1) (epilogue_hidden's layout)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) wh[i][j] = mw[i][j] >> (8 * h);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n_base + 32 * i + 8 * g + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                if constexpr (ADD) {
                    const uint2 e = keep[i][g][j];
                    v0 += flo(e.x); v1 += fhi(e.x); v2 += flo(e.y); v3 += fhi(e.y);
                }
                const unsigned u0 = fpack(v0, v1), u1 = fpack(v2, v3);
                if constexpr (KEEP) keep[i][g][j] = make_uint2(u0, u1);
                // the ReLU decisions applied to the PACKED pairs: the pair's two bits sit 16 apart in the word -> the halves'
                // multipliers {0, 1} by one shift + one and -> v_pk_mul_lo_u16 (a bf16 times 1 or 0 as integers is itself or +0)
                const unsigned k0 = (wh[i][j] >> (2 * g)) & 0x00010001u, k1 = (wh[i][j] >> (2 * g + 1)) & 0x00010001u;
                unsigned w0, w1;
                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(w0) : "v"(u0), "v"(k0));
                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(w1) : "v"(u1), "v"(k1));
                *reinterpret_cast<uint2 *>(&out[poff + 32 * j + p][n]) = make_uint2(w0, w1);
            }
        }
}

template <int PH>
__global__ __launch_bounds__(256 * PH, 2 / PH) void decoder_bwd_fused_kernel(BwdArgs a)
{
    gags_h16::h16_saturate_mode();  // (f16 tier: conversions saturate in hardware; half16.h)
    __shared__ __attribute__((aligned(16))) unsigned short bufA[FT * PH][FLD];
    __shared__ __attribute__((aligned(16))) unsigned short bufB[FT * PH][FLD];
    Tile X = bufA, Y = bufB;
    constexpr int TP = FT * PH, NT = 256 * PH;  // pixels per tile, threads: PH groups of four waves, 64 pixels each
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, poff = FT * (tid >> 8);
    const int64_t p0 = (int64_t)blockIdx.x * TP;
    const int n_base = 64 * wave;
    f32x16 acc[2][2];
    uint2 skip[2][4][2];  // the skip gradient in flight (g36, later g13): this lane's own 32 values, packed bf16
    unsigned mw[2][2], mw2[2][2];  // ReLU masks of the layer in hand and of the next one down (requested one layer early:
                                   // a load issued right before a layer's weight stream holds up every refill behind it)
    uint4 slab[8];
    WFrag wf;

    // L8: dz8 [64, n_last] through X in slabs of 256 columns -> dz7 (Y)
    fetch_mask(mw, a.mask[8], p0, a.P, wave, lane, poff);
    for (int kb = 0; kb < a.n_last; kb += FH) {
        fetch_tile<NT>(slab, a.dz8, a.n_last, kb, p0, a.P, tid);
        if (kb) __syncthreads();  // the previous slab has been multiplied
        commit_tile<NT>(X, slab, tid);
        __syncthreads();
        wprefetch(wf, a.Wt[8], a.n_last / 16, 2 * wave, kb / 16, 16, lane);
        layer_main(acc, wf, a.Wt[8], a.n_last / 16, 2 * wave, kb / 16, 16, X, lane, kb == 0, poff);
    }
    wprefetch(wf, a.Wt[7], 16, 2 * wave, 0, 16, lane);
    fetch_mask(mw2, a.mask[7], p0, a.P, wave, lane, poff);
    epilogue_dgrad<false, false>(acc, n_base, mw, Y, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[7], p0, a.P, Y, tid);
    // L7: dz7 (Y) -> dz6 (X)
    layer_main(acc, wf, a.Wt[7], 16, 2 * wave, 0, 16, Y, lane, true, poff);
    wprefetch(wf, a.Wt[6], 16, 2 * wave, 0, 16, lane);
    fetch_mask(mw, a.mask[6], p0, a.P, wave, lane, poff);
    epilogue_dgrad<false, false>(acc, n_base, mw2, X, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[6], p0, a.P, X, tid);
    // L6: dz6 (X) -> g36 (kept), dz5 = g36 * [x4 > 0] (Y)
    layer_main(acc, wf, a.Wt[6], 16, 2 * wave, 0, 16, X, lane, true, poff);
    wprefetch(wf, a.Wt[5], 16, 2 * wave, 0, 16, lane);
    fetch_mask(mw2, a.mask[5], p0, a.P, wave, lane, poff);
    epilogue_dgrad<true, false>(acc, n_base, mw, Y, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[5], p0, a.P, Y, tid);
    // L5: dz5 (Y) -> dz4 (X)
    layer_main(acc, wf, a.Wt[5], 16, 2 * wave, 0, 16, Y, lane, true, poff);
    wprefetch(wf, a.Wt[4], 16, 2 * wave, 0, 16, lane);
    fetch_mask(mw, a.mask[4], p0, a.P, wave, lane, poff);
    epilogue_dgrad<false, false>(acc, n_base, mw2, X, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[4], p0, a.P, X, tid);
    // L4: dz4 (X) + g36 -> dz3 (Y)
    layer_main(acc, wf, a.Wt[4], 16, 2 * wave, 0, 16, X, lane, true, poff);
    wprefetch(wf, a.Wt[3], 16, 2 * wave, 0, 16, lane);
    fetch_mask(mw2, a.mask[3], p0, a.P, wave, lane, poff);
    epilogue_dgrad<false, true>(acc, n_base, mw, Y, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[3], p0, a.P, Y, tid);
    // L3: dz3 (Y) -> g13 (kept), dz2 = g13 * [x2 > 0] (X)
    layer_main(acc, wf, a.Wt[3], 16, 2 * wave, 0, 16, Y, lane, true, poff);
    wprefetch(wf, a.Wt[2], 16, 2 * wave, 0, 16, lane);
    fetch_mask(mw, a.mask[2], p0, a.P, wave, lane, poff);
    epilogue_dgrad<true, false>(acc, n_base, mw2, X, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[2], p0, a.P, X, tid);
    // L2: dz2 (X) -> dz1 (Y)
    layer_main(acc, wf, a.Wt[2], 16, 2 * wave, 0, 16, X, lane, true, poff);
    wprefetch(wf, a.Wt[1], 16, 2 * wave, 0, 16, lane);
    fetch_mask(mw2, a.mask[1], p0, a.P, wave, lane, poff);
    epilogue_dgrad<false, false>(acc, n_base, mw, Y, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[1], p0, a.P, Y, tid);
    // L1: dz1 (Y) + g13 -> dz0 (X)
    layer_main(acc, wf, a.Wt[1], 16, 2 * wave, 0, 16, Y, lane, true, poff);
    epilogue_dgrad<false, true>(acc, n_base, mw2, X, skip, lane, poff);
    __syncthreads();
    store_tile<NT>(a.dz[0], p0, a.P, X, tid);
    // L0: d x[p][c] = sum_n dz0[p][n] W0[n][c]: 32 (padded) channels x 64 pixels = two accumulator tiles, waves 0 and 1;
    // rounded to bf16 and widened, as the layer-by-layer path hands it over (gags_decoder_unpack_grad)
    if (a.gin && wave < 2) {  // (of every group of four)
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        const unsigned short *w0 = a.Wt[0] + lane * 8;  // fragment order: [n_tile 0][k_step][lane][8]
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8 *>(w0 + 512 * ks);
            const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(&X[poff + 32 * wave + (lane & 31)][16 * ks + 8 * (lane >> 5)]);
            c = h16_mfma(wf, bf, c);
        }
        const int64_t pg = p0 + poff + 32 * wave + (lane & 31);
        const float gs = a.gin_scale ? a.gin_scale[0] : 1.0f;
        if (pg < a.P) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ch = 8 * g + 4 * (lane >> 5) + e;
                    if (ch < a.c_in) a.gin[pg * a.c_in + ch] = flo(fpack(c[4 * g + e], 0.f)) * gs;
                }
        }
    }
}

// tile size experiment switch (GAGS_FUSED_PH=1|2: 64- or 128-pixel tiles), read once
inline int fused_ph()
{
    static const int ph = [] { const char *e = getenv("GAGS_FUSED_PH"); return (e && e[0] == '2') ? 2 : 1; }();
    return ph;
}

}  // namespace

extern "C" int GAGS_DEC(gags_decoder_fwd_fused)(int64_t n_pix, int c_in, int n_last, const float *x, const void *const *w_bf16,
                                      const float *const *bias, void *const *acts_bf16, void *masks, float *logits,
                                      void *stream)
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
        a.mask[i] = (masks && i > 0) ? (unsigned *)masks + (size_t)(i - 1) * ((n_pix + 63) / 64 * 64) * 8 : nullptr;
    }
    if (fused_ph() == 2)
        hipLaunchKernelGGL(decoder_fwd_fused_kernel<2>, dim3((unsigned)((n_pix + 2 * FT - 1) / (2 * FT))), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(decoder_fwd_fused_kernel<1>, dim3((unsigned)((n_pix + FT - 1) / FT)), dim3(256), 0, (hipStream_t)stream, a);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int GAGS_DEC(gags_decoder_bwd_fused_scaled)(int64_t n_pix, int c_in, int n_last, const void *dz_last_bf16,
                                             const void *const *wt_bf16, const void *masks, void *const *dz_bf16, float *gin,
                                             const float *gin_scale, void *stream);

extern "C" int GAGS_DEC(gags_decoder_bwd_fused)(int64_t n_pix, int c_in, int n_last, const void *dz_last_bf16, const void *const *wt_bf16,
                                      const void *masks, void *const *dz_bf16, float *gin, void *stream)
{
    return GAGS_DEC(gags_decoder_bwd_fused_scaled)(n_pix, c_in, n_last, dz_last_bf16, wt_bf16, masks, dz_bf16, gin, nullptr, stream);
}

extern "C" int GAGS_DEC(gags_decoder_bwd_fused_scaled)(int64_t n_pix, int c_in, int n_last, const void *dz_last_bf16,
                                             const void *const *wt_bf16, const void *masks, void *const *dz_bf16, float *gin,
                                             const float *gin_scale, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c_in <= 0 || c_in > 32 || n_last <= 0 || n_last % FH != 0 || !dz_last_bf16 || !wt_bf16 || !masks || !dz_bf16)
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    BwdArgs a;
    a.dz8 = (const unsigned short *)dz_last_bf16;
    a.gin = gin; a.gin_scale = gin_scale; a.P = n_pix; a.c_in = c_in; a.n_last = n_last;
    for (int i = 0; i < 9; ++i) {
        if (!wt_bf16[i]) return GAGS_EINVAL;
        a.Wt[i] = (const unsigned short *)wt_bf16[i];
        a.mask[i] = i > 0 ? (const unsigned *)masks + (size_t)(i - 1) * ((n_pix + 63) / 64 * 64) * 8 : nullptr;
    }
    for (int i = 0; i < 8; ++i) {
        if (!dz_bf16[i]) return GAGS_EINVAL;
        a.dz[i] = (unsigned short *)dz_bf16[i];
    }
    if (fused_ph() == 2)
        hipLaunchKernelGGL(decoder_bwd_fused_kernel<2>, dim3((unsigned)((n_pix + 2 * FT - 1) / (2 * FT))), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(decoder_bwd_fused_kernel<1>, dim3((unsigned)((n_pix + FT - 1) / FT)), dim3(256), 0, (hipStream_t)stream, a);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
