// N1 at the reference's precision (VERDICT r2 "a decoder mode at the reference's precision").
// models/networks.py:109-248 are fp32 Conv2d stacks; gfx950 has no TF32 and its fp32 matrix rate (157 TFLOP/s) would cost
// >= 50 ms per iteration.  Here every fp32 operand is written as THREE bfloat16 terms,
//     a = h + m + l,   h = bf16(a),  m = bf16(a - h),  l = bf16(a - h - m)        (exact: 3 x 8 significand bits >= 24)
// and a product as the six MFMA terms of order <= 2,
//     a b ~ h h' + h m' + m h' + h l' + m m' + l h'                               (dropped: <= 2^-24 |a b|, below fp32 rounding)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: the arithmetic of an fp32 GEMM at 2.6x the fp32 matrix rate.
// Activations, gradients and weights are fp32 in memory; the split happens on the way into LDS.
// One tile kernel serves
//   layers / input gradients   Y[p, n] = (act(sum_k (A1 + A2)[p, k] W[n, k] + bias[n]) + E[p, n]) * (mask_src[p, n] > 0)
//   weight gradients           dW[n, k] = sum_p dZ[p, n] (A1 + A2)[p, k]   (contraction over the SLOW index of both operands:
//                              the loader transposes; pixel chunks -> partial matrices -> summed in chunk order: NO atomics,
//                              bit-reproducible)
// plus the bias gradient (column sums, same two deterministic stages) and the output heads' backward in fp32.
#include "common.h"
#include "gags_next.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int XM = 128, XN = 128, XK = 32;  // workgroup tile: 128 x 128 outputs, 32 contraction elements per step
constexpr int XLD = XK + 8;                 // LDS row pitch in bf16 (80 B: 16-byte aligned rows, banks spread)
constexpr int NTERM = 3;

__device__ __forceinline__ unsigned short xf2bf(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (operands are finite)
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float xbf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void split3(float a, unsigned short (&t)[NTERM])
{
    t[0] = xf2bf(a);
    float r = a - xbf2f(t[0]);
    t[1] = xf2bf(r);
    r = r - xbf2f(t[1]);
    t[2] = xf2bf(r);
}

struct XArgs {
    const float *A1, *A2;   // NT: [M, K] rows contiguous in K.   TN: [P, M] (the contraction index p is the slow one)
    const float *B;         // NT: [N, K].                        TN: [P, N]
    const float *B2;        // TN only: second activation source, summed with B
    const float *bias, *mask_src, *E;
    float *Y, *Ypre;
    int64_t M;              // NT: rows of Y (pixels).  TN: unused
    int N, K;               // NT: columns of Y, contraction length.  TN: output is [Mo, No]
    int lda, ldb, ldy, relu;
    int Mo, No;             // TN: output rows (n_out) / columns (k_in)
    int64_t P, chunk;       // TN: contraction length and pixels per workgroup
    float *part;            // TN: partial outputs [n_chunks][Mo * No]
};

// rows [r0, r0 + 128) x contraction [k0, k0 + 32) of a row-major fp32 matrix (+ optional second summand) -> three bf16 planes
template <bool TWO>
__device__ __forceinline__ void stage_rows(unsigned short (*S)[XM][XLD], const float *__restrict__ a1, const float *__restrict__ a2,
                                           int ld, int64_t r0, int64_t rows, int k0, int K, bool vec, int tid)
{
    const int kc = (tid & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (tid >> 3) + 32 * q;
        const int64_t rg = r0 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (rg < rows) {
            const float *p1 = a1 + rg * ld + k0 + kc;
            if (vec && k0 + kc + 3 < K) {
                const float4 u = *reinterpret_cast<const float4 *>(p1);
                v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
                if constexpr (TWO) {
                    const float4 w = *reinterpret_cast<const float4 *>(a2 + rg * ld + k0 + kc);
                    v[0] += w.x; v[1] += w.y; v[2] += w.z; v[3] += w.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k0 + kc + e < K) {
                        v[e] = p1[e];
                        if constexpr (TWO) v[e] += a2[rg * ld + k0 + kc + e];
                    }
            }
        }
        unsigned short t[4][NTERM];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(v[e], t[e]);
#pragma unroll
        for (int s = 0; s < NTERM; ++s)
            *reinterpret_cast<uint2 *>(&S[s][r][kc]) =
                make_uint2((unsigned)t[0][s] | ((unsigned)t[1][s] << 16), (unsigned)t[2][s] | ((unsigned)t[3][s] << 16));
    }
}

// TN: pixels [p0, p0 + 32) x columns [c0, c0 + 128) of src [P, C] -> S[term][column][pixel] (transposed on the way in)
template <bool TWO>
__device__ __forceinline__ void stage_cols(unsigned short (*S)[XM][XLD], const float *__restrict__ s1, const float *__restrict__ s2,
                                           int ld, int64_t p0, int64_t p_end, int c0, int C, bool vec, int tid)
{
    const int cc = (tid & 31) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pr = (tid >> 5) + 8 * q;
        const int64_t pg = p0 + pr;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (pg < p_end) {
            const float *src = s1 + pg * ld + c0 + cc;
            if (vec && c0 + cc + 3 < C) {
                const float4 u = *reinterpret_cast<const float4 *>(src);
                v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
                if constexpr (TWO) {
                    const float4 w = *reinterpret_cast<const float4 *>(s2 + pg * ld + c0 + cc);
                    v[0] += w.x; v[1] += w.y; v[2] += w.z; v[3] += w.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + cc + e < C) {
                        v[e] = src[e];
                        if constexpr (TWO) v[e] += s2[pg * ld + c0 + cc + e];
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned short t[NTERM];
            split3(v[e], t);
#pragma unroll
            for (int s = 0; s < NTERM; ++s) S[s][cc + e][pr] = t[s];
        }
    }
}

// one 32-wide contraction step from LDS: each wave owns 64 x 64 outputs (2 x 2 MFMA tiles)
__device__ __forceinline__ void tile_step(f32x16 (&acc)[2][2], unsigned short (*Xs)[XM][XLD], unsigned short (*Ys)[XN][XLD], int wy,
                                          int wx, int lane)
{
#pragma unroll
    for (int ks = 0; ks < XK; ks += 16) {
        bf16x8 a[2][NTERM], b[2][NTERM];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < NTERM; ++s) {
                a[i][s] = *reinterpret_cast<const bf16x8 *>(&Xs[s][wy * 64 + i * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
                b[i][s] = *reinterpret_cast<const bf16x8 *>(&Ys[s][wx * 64 + i * 32 + (lane & 31)][ks + 8 * (lane >> 5)]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            }
    }
}

template <bool TWO>
__global__ __launch_bounds__(256, 2) void gemm_x3_nt_kernel(XArgs g)
{
    __shared__ __attribute__((aligned(16))) unsigned short Xs[NTERM][XM][XLD];
    __shared__ __attribute__((aligned(16))) unsigned short Ys[NTERM][XN][XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * XM;
    const int n0 = blockIdx.y * XN;
    const bool veca = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A1) & 15) == 0) &&
                      (!TWO || (reinterpret_cast<uintptr_t>(g.A2) & 15) == 0);
    const bool vecb = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += XK) {
        stage_rows<TWO>(Xs, g.A1, g.A2, g.lda, m0, g.M, k0, g.K, veca, tid);
        stage_rows<false>(Ys, g.B, nullptr, g.ldb, n0, g.N, k0, g.K, vecb, tid);
        __syncthreads();
        tile_step(acc, Xs, Ys, wy, wx, lane);
        __syncthreads();
    }
    // accumulator of tile (i, j): column = lane & 31 -> n, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> pixel
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wx * 64 + j * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t p = m0 + wy * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (p >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.relu) v = fmaxf(v, 0.f);
                const size_t o = (size_t)p * g.ldy + n;
                if (g.E) v += g.E[o];
                if (g.Ypre) g.Ypre[o] = v;
                if (g.mask_src) v = g.mask_src[o] > 0.f ? v : 0.f;
                if (g.Y) g.Y[o] = v;
            }
        }
}

template <bool TWO>
__global__ __launch_bounds__(256, 2) void gemm_x3_tn_kernel(XArgs g)
{
    __shared__ __attribute__((aligned(16))) unsigned short Xs[NTERM][XM][XLD];
    __shared__ __attribute__((aligned(16))) unsigned short Ys[NTERM][XN][XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    const int64_t pa = (int64_t)blockIdx.x * g.chunk, pb = min(pa + g.chunk, g.P);
    const int m0 = blockIdx.y * XM, n0 = blockIdx.z * XN;
    const bool veca = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A1) & 15) == 0);
    const bool vecb = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) &&
                      (!TWO || (reinterpret_cast<uintptr_t>(g.B2) & 15) == 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int64_t p0 = pa; p0 < pb; p0 += XK) {
        stage_cols<false>(Xs, g.A1, nullptr, g.lda, p0, pb, m0, g.Mo, veca, tid);
        stage_cols<TWO>(Ys, g.B, g.B2, g.ldb, p0, pb, n0, g.No, vecb, tid);
        __syncthreads();
        tile_step(acc, Xs, Ys, wy, wx, lane);
        __syncthreads();
    }
    float *out = g.part + (size_t)blockIdx.x * g.Mo * g.No;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = n0 + wx * 64 + j * 32 + (lane & 31);
            if (k >= g.No) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = m0 + wy * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < g.Mo) out[(size_t)n * g.No + k] = acc[i][j][r];
            }
        }
}

// out[e] = sum over the chunks, in chunk order (fixed => reproducible)
__global__ __launch_bounds__(256) void sum_parts_kernel(int n_chunks, int64_t elems, const float *__restrict__ part,
                                                        float *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    float s = 0.f;
    for (int c = 0; c < n_chunks; ++c) s += part[(size_t)c * elems + e];
    out[e] = s;
}

// column sums of dz[P, C] over a chunk of pixels (the bias gradient), one thread per column, rows in order
__global__ __launch_bounds__(256) void colsum_kernel(int64_t P, int C, int ld, int64_t chunk, const float *__restrict__ dz,
                                                     float *__restrict__ part)
{
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t pa = (int64_t)blockIdx.x * chunk, pb = min(pa + chunk, P);
    float s = 0.f;
    for (int64_t p = pa; p < pb; ++p) s += dz[p * ld + c];
    part[(size_t)blockIdx.x * C + c] = s;
}

// backward of the output heads in fp32 (see decoder.hip head_bwd_kernel): one wave per pixel
//   mode 0 (y = x / max(||x||, eps)):  dz = (g - y <y, g>) / max(||x||, eps);   mode 1 (softmax):  dz = y (g - <y, g>)
__global__ __launch_bounds__(256) void head_bwd_exact_kernel(int64_t P, int C, int ldx, int mode, const float *__restrict__ x,
                                                             const float *__restrict__ G, int layout, float *__restrict__ dz,
                                                             int lddz)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    float s = 0.f, m = -3.0e38f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[p * ldx + c];
        s = fmaf(v, v, s);
        m = fmaxf(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); m = fmaxf(m, __shfl_xor(m, off, 64)); }
    float z = 0.f;
    if (mode == 1) {
        for (int c = lane; c < C; c += 64) z += expf(x[p * ldx + c] - m);
        for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
    }
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[p * ldx + c];
        const float y = mode == 0 ? v / nrm : expf(v - m) / z;
        const float gv = layout == 1 ? G[p * C + c] : G[(size_t)c * P + p];
        dot = fmaf(y, gv, dot);
    }
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    for (int c = lane; c < lddz; c += 64) {
        float d = 0.f;
        if (c < C) {
            const float v = x[p * ldx + c];
            const float gv = layout == 1 ? G[p * C + c] : G[(size_t)c * P + p];
            if (mode == 0) d = (gv - (v / nrm) * dot) / nrm;
            else d = (expf(v - m) / z) * (gv - dot);
        }
        dz[p * lddz + c] = d;
    }
}

inline int64_t al256(int64_t v) { return (v + 255) / 256 * 256; }
inline int64_t wgrad_chunk(int64_t n_pix)
{
    // ~512 pixel chunks (two workgroups per CU), whole 32-pixel steps
    return ((n_pix + 511) / 512 + XK - 1) / XK * XK;
}

}  // namespace

extern "C" int gags_decoder_layer_exact(int64_t n_pix, int n_out, int k_in, const float *a1, const float *a2, int lda,
                                        const float *w, const float *bias, int relu, const float *mask_src,
                                        const float *residual, float *y, float *y_premask, int ldy, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || n_out <= 0 || k_in <= 0 || lda < k_in || ldy < n_out || !a1 || !w || (!y && !y_premask)) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    XArgs g = {};
    g.A1 = a1; g.A2 = a2; g.B = w; g.bias = bias; g.mask_src = mask_src; g.E = residual; g.Y = y; g.Ypre = y_premask;
    g.M = n_pix; g.N = n_out; g.K = k_in; g.lda = lda; g.ldb = k_in; g.ldy = ldy; g.relu = relu;
    const dim3 grid((unsigned)((n_pix + XM - 1) / XM), (unsigned)((n_out + XN - 1) / XN));
    if (a2) hipLaunchKernelGGL(gemm_x3_nt_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(gemm_x3_nt_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, g);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int64_t gags_decoder_wgrad_exact_scratch_bytes(int64_t n_pix, int n_out, int k_in)
{
    if (n_pix <= 0 || n_out <= 0 || k_in <= 0) return 0;
    const int64_t chunk = wgrad_chunk(n_pix), n_chunks = (n_pix + chunk - 1) / chunk;
    return al256(n_chunks * ((int64_t)n_out * k_in + n_out) * 4);
}

extern "C" int gags_decoder_wgrad_exact(int64_t n_pix, int n_out, int k_in, const float *dz, int lddz, const float *a1,
                                        const float *a2, int lda, float *d_w, float *d_b, void *scratch,
                                        int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || n_out <= 0 || k_in <= 0 || lddz < n_out || lda < k_in || !dz || !a1 || !d_w) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_pix == 0) {
        if (hipMemsetAsync(d_w, 0, sizeof(float) * (size_t)n_out * k_in, st) != hipSuccess) return GAGS_ELAUNCH;
        if (d_b && hipMemsetAsync(d_b, 0, sizeof(float) * (size_t)n_out, st) != hipSuccess) return GAGS_ELAUNCH;
        return GAGS_OK;
    }
    if (!scratch || scratch_bytes < gags_decoder_wgrad_exact_scratch_bytes(n_pix, n_out, k_in)) return GAGS_ESCRATCH;
    const int64_t chunk = wgrad_chunk(n_pix);
    const int n_chunks = (int)((n_pix + chunk - 1) / chunk);
    float *part = (float *)scratch, *part_b = part + (size_t)n_chunks * n_out * k_in;
    XArgs g = {};
    g.A1 = dz; g.lda = lddz; g.B = a1; g.B2 = a2; g.ldb = lda; g.Mo = n_out; g.No = k_in; g.P = n_pix; g.chunk = chunk; g.part = part;
    const dim3 grid((unsigned)n_chunks, (unsigned)((n_out + XM - 1) / XM), (unsigned)((k_in + XN - 1) / XN));
    if (a2) hipLaunchKernelGGL(gemm_x3_tn_kernel<true>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_x3_tn_kernel<false>, grid, dim3(256), 0, st, g);
    const int64_t elems = (int64_t)n_out * k_in;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, n_chunks, elems, part, d_w);
    if (d_b) {
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)n_chunks, (unsigned)((n_out + 255) / 256)), dim3(256), 0, st, n_pix, n_out,
                           lddz, chunk, dz, part_b);
        hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, n_chunks, (int64_t)n_out, part_b,
                           d_b);
    }
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_decoder_head_bwd_exact(int64_t n_pix, int c, int ldx, int mode, const float *x, const float *g, int layout,
                                           float *dz, int lddz, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || ldx < c || lddz < c || (mode != 0 && mode != 1) || (layout != 0 && layout != 1) ||
        (n_pix > 0 && (!x || !g || !dz)))
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(head_bwd_exact_kernel, dim3((unsigned)((n_pix + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n_pix, c, ldx, mode,
                       x, g, layout, dz, lddz);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
