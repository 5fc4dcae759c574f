// N2 / N4: the image-space kernels either side of the rasterizer in the reference's training and query
// loops -- ground-truth feature assembly, masked L1 map, segment-wise losses, LERF relevancy.  All HBM-bound byte /
// gather work over [C, H, W] maps: coalesced along pixels, small tables (segment embeddings, segment statistics) left
// to L2; nothing is reshaped into a GEMM.  Numerics follow the reference's torch ops (fp32; sums that torch does as
// one big reduction are accumulated in double here).
#include "common.h"
#include "gags_next.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// get_trained_seg (utils/loss_utils.py:138-154)
// A workgroup = 64 x 4 pixels; the three planes' (64 + 4) x (4 + 4) patches go through LDS once (zero outside the image) and
// every pixel adds its 25 taps in the order of the direct form (rows, then columns; a zero tap adds exactly nothing): the same
// bits as one bounds-checked global load per tap, 94 -> ~20 us at 1080p (round 6).
constexpr int TSW = 64, TSH = 4;
__global__ __launch_bounds__(256) void trained_seg_kernel(int h, int w, const float *__restrict__ seg_map,
                                                          const float *__restrict__ scale_map, float *__restrict__ out)
{
    __shared__ float patch[3][TSH + 4][TSW + 4];
    const int x0 = blockIdx.x * TSW, y0 = blockIdx.y * TSH;
    for (int i = threadIdx.x; i < 3 * (TSH + 4) * (TSW + 4); i += 256) {
        const int ch = i / ((TSH + 4) * (TSW + 4)), r = i - ch * ((TSH + 4) * (TSW + 4));
        const int py = r / (TSW + 4), px = r - py * (TSW + 4);
        const int yy = y0 + py - 2, xx = x0 + px - 2;
        patch[ch][py][px] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? scale_map[((size_t)ch * h + yy) * w + xx] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & (TSW - 1), ly = threadIdx.x / TSW;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y >= h) return;
    float best = 0.f;
    int arg = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float s = 0.f;  // conv2d with a 5x5 kernel of 1/25, zero padding 2
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) s = fmaf(patch[ch][ly + dy][lx + dx], 0.04f, s);
        if (ch == 0 || s > best) { best = s; arg = ch; }  // first maximum wins, as torch.argmax
    }
    out[(size_t)y * w + x] = seg_map[((size_t)(1 + arg) * h + y) * w + x];
}

// ---------------------------------------------------------------------------------------------------------------
// scale_regulation_loss (utils/loss_utils.py:59-66)
__global__ __launch_bounds__(256) void entropy_fwd_kernel(int64_t n, const float *__restrict__ s, double *__restrict__ acc)
{
    __shared__ double sm[4];
    double a = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = s[i];
        a += (double)(-v * logf(v + 1e-6f));
    }
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, (sm[0] + sm[1]) + (sm[2] + sm[3]));
}

__global__ __launch_bounds__(256) void entropy_bwd_kernel(int64_t n, const float *__restrict__ s, float v, float *__restrict__ vs)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = s[i];
    vs[i] = -(logf(x + 1e-6f) + x / (x + 1e-6f)) * v;
}

// (the cotangent as a device scalar: no host readback in the middle of a backward pass; the factor is formed as the host
// formed it -- the quotient in double, rounded to float once)
__global__ __launch_bounds__(256) void entropy_bwd_dev_kernel(int64_t n, const float *__restrict__ s, const float *__restrict__ v,
                                                              float *__restrict__ vs)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f = (float)((double)v[0] / (double)n);
    const float x = s[i];
    vs[i] = -(logf(x + 1e-6f) + x / (x + 1e-6f)) * f;
}

// ---------------------------------------------------------------------------------------------------------------
// per-segment moments.  One wave = 64 consecutive pixels.
// wave64 sum on the VALU (DPP: a pairwise tree, total in lane 63), returned wave-uniform
__device__ __forceinline__ float seg_wave_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// FOUR consecutive pixels per lane (one 16-byte load per channel): on real segment maps (large regions) a wave of 256
// pixels still meets one or two ids, so rounds, wave sums and double atomics are a quarter as many per pixel.  (The
// synthetic map of tools/decoder_bench.py draws a random id per 8x8 block -- 32 ids per wave -- and is bound by the
// rounds' wave sums either way: 0.5 ms per call at 1080p, c = 16.)
__global__ __launch_bounds__(256) void segment_stats_kernel(int64_t n_pix, int c, const float *__restrict__ x,
                                                            const float *__restrict__ seg, int n_seg,
                                                            double *__restrict__ s1, double *__restrict__ s2,
                                                            int32_t *__restrict__ cnt, int vec, int copies, int pm)
{   // pm: x is PIXEL-major [n_pix, c] (the rasterizer's own layout: the [C,H,W] map a loss receives is a permuted view of
    // it, and `.contiguous()` on that view is a 132 MB copy per iteration at 1080p, c = 16); else channel-major [c, n_pix]
    // `copies` private sets of accumulators, picked by workgroup: the double atomics execute at the memory side and
    // serialize per ADDRESS (~0.5 us each) -- with a few hundred segments in the image every address takes ~900 of them
    {
        const size_t cp = blockIdx.x % (unsigned)copies;
        s1 += cp * (size_t)n_seg * c; s2 += cp * (size_t)n_seg * c; cnt += cp * (size_t)n_seg;
    }
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int lane = threadIdx.x & 63;
    int id[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        id[q] = -1;
        if (p0 + q < n_pix) {
            const float f = seg[p0 + q];
            id[q] = (f >= 0.f && f < (float)n_seg) ? (int)f : -1;
        }
    }
    // one round per distinct segment id in the wave: the group's values of a channel are summed in fp32 on the VALU (in
    // the lane, then pairwise over the wave; the sums over many waves are the ones that need doubles) and the leader
    // issues ONE double atomic per moment.  Channels in groups of 8 held in registers: read once.
    for (int cb = 0; cb < c; cb += 8) {
        float xv[8][4];
        if (pm && (c & 7) == 0) {  // (uniform) pixel-major rows of whole 8-channel groups: two 16-byte loads per pixel
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float *px = x + (size_t)min(p0 + q, n_pix - 1) * c + cb;
                const float4 t0 = *reinterpret_cast<const float4 *>(px), t1 = *reinterpret_cast<const float4 *>(px + 4);
                xv[0][q] = t0.x; xv[1][q] = t0.y; xv[2][q] = t0.z; xv[3][q] = t0.w;
                xv[4][q] = t1.x; xv[5][q] = t1.y; xv[6][q] = t1.z; xv[7][q] = t1.w;
            }
        } else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = min(cb + j, c - 1);
            const float *row = x + (size_t)ch * n_pix;
            if (pm) {  // (a lane walks its pixels' rows channel by channel: every 64-byte line is used up over j)
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[j][q] = x[(size_t)min(p0 + q, n_pix - 1) * c + ch];
            } else if (vec) {  // n_pix % 4 == 0 and a 16-byte aligned base: every row is aligned (uniform)
                const float4 t = *reinterpret_cast<const float4 *>(row + min(p0, n_pix - 4));
                xv[j][0] = t.x; xv[j][1] = t.y; xv[j][2] = t.z; xv[j][3] = t.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[j][q] = row[min(p0 + q, n_pix - 1)];
            }
        }
        unsigned pend = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) pend |= (id[q] >= 0 ? 1u : 0u) << q;
        unsigned long long todo = __ballot(pend != 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int fq = __ffs((int)pend) - 1;  // this lane's first pending pixel (-1: none)
            const int first_id = fq == 0 ? id[0] : fq == 1 ? id[1] : fq == 2 ? id[2] : id[3];
            const int cur = __builtin_amdgcn_readlane(first_id, leader);
            const int lq = __builtin_amdgcn_readlane(fq, leader);
            unsigned in = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) in |= ((pend >> q & 1u) && id[q] == cur ? 1u : 0u) << q;
            int n_in = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) n_in += (int)__popcll(__ballot((in >> q & 1u) != 0));
            if (cb == 0 && lane == leader) atomicAdd(&cnt[cur], n_in);
            const double ng = (double)n_in;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // moments of the group about one of its OWN values (the leader's first pixel): the fp32 sums then carry the
                // spread, not the mean, and the shift goes back in double -- sum x^2 - (sum x)^2 / n used to cancel in fp32
                // rounding once a region's variance fell below ~1e-7 mean^2, which is where this loss drives it
                const float mine = lq == 0 ? xv[j][0] : lq == 1 ? xv[j][1] : lq == 2 ? xv[j][2] : xv[j][3];
                const float sft = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), leader));
                float v = 0.f, vv = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = (in >> q & 1u) ? xv[j][q] - sft : 0.f;
                    v += t; vv = fmaf(t, t, vv);
                }
                const float a = seg_wave_sum(v), b = seg_wave_sum(vv);
                if (lane == leader && cb + j < c) {
                    const double sd = (double)sft, ad = (double)a;
                    atomicAdd(&s1[(size_t)cur * c + cb + j], ad + ng * sd);
                    atomicAdd(&s2[(size_t)cur * c + cb + j], (double)b + 2.0 * sd * ad + ng * sd * sd);
                }
            }
            pend &= ~in;
            todo = __ballot(pend != 0);
        }
    }
}

// The same moments by RUNS (round 6): every stream of consecutive pixels keeps the sums of its current run of equal ids in
// registers -- about the run's first value, as above -- and adds a finished run to a table of doubles in LDS; the workgroup's
// table leaves as ONE private copy (plain stores; the copies are summed by the caller like the private accumulator sets of
// the kernel above).  No wave sums and no global atomics at all: the work per pixel does not depend on how finely the map is
// cut (the rounds above cost ~200 VALU instructions per distinct id in a wave: 0.48 ms at 1080p, c = 16, on a map of 8 x 8
// blocks), and large regions flush almost nothing.  CL lanes per pixel: 16 = the sixteen channels of a pixel-major row,
// 1 = a single plane.
template <int CL>
__global__ __launch_bounds__(256) void segment_stats_runs_kernel(int64_t n_pix, const float *__restrict__ x,
                                                                 const float *__restrict__ seg, int n_seg, double *__restrict__ s1,
                                                                 double *__restrict__ s2, int32_t *__restrict__ cnt)
{
    extern __shared__ double seg_tab[];  // s1 [n_seg][CL], s2 [n_seg][CL], counts [n_seg]
    constexpr int NS = 256 / CL, RUN = CL == 16 ? 128 : 32;  // streams per workgroup, pixels per stream and block
    const int tid = threadIdx.x, n_tab = n_seg * CL;
    double *t1 = seg_tab, *t2 = seg_tab + n_tab;
    int *tc = reinterpret_cast<int *>(seg_tab + 2 * (size_t)n_tab);
    for (int i = tid; i < 2 * n_tab; i += 256) seg_tab[i] = 0.0;
    for (int i = tid; i < n_seg; i += 256) tc[i] = 0;
    __syncthreads();
    const int j = tid % CL, stream = tid / CL;
    int cur = -1, n = 0;
    float sft = 0.f, v = 0.f, vv = 0.f;
    auto flush = [&]() {
        if (cur >= 0 && n > 0) {
            const double sd = (double)sft, ad = (double)v, ng = (double)n;
            atomicAdd(&t1[cur * CL + j], ad + ng * sd);
            atomicAdd(&t2[cur * CL + j], (double)vv + 2.0 * sd * ad + ng * sd * sd);
            if (j == 0) atomicAdd(&tc[cur], n);
        }
    };
    auto take = [&](float f, float xv) {
        const int id = (f >= 0.f && f < (float)n_seg) ? (int)f : -1;
        if (id != cur) {
            flush();
            cur = id; n = 0; sft = xv; v = 0.f; vv = 0.f;
        }
        const float t = xv - sft;
        v += t; vv = fmaf(t, t, vv); ++n;
    };
    for (int64_t base = (int64_t)blockIdx.x * (NS * RUN); base < n_pix; base += (int64_t)gridDim.x * (NS * RUN)) {
        int64_t p = base + (int64_t)stream * RUN;
        const int64_t pe = min(p + RUN, n_pix);
        for (; p + 4 <= pe; p += 4) {  // four pixels' loads in flight
            float f[4], xv[4];
            if constexpr (CL == 1) {
                if (((reinterpret_cast<uintptr_t>(x + p) | reinterpret_cast<uintptr_t>(seg + p)) & 15) == 0) {
                    const float4 a = *reinterpret_cast<const float4 *>(x + p), b = *reinterpret_cast<const float4 *>(seg + p);
                    xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; f[0] = b.x; f[1] = b.y; f[2] = b.z; f[3] = b.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { xv[q] = x[p + q]; f[q] = seg[p + q]; }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { xv[q] = x[(p + q) * CL + j]; f[q] = seg[p + q]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) take(f[q], xv[q]);
        }
        for (; p < pe; ++p) take(seg[p], x[p * CL + j]);
        flush();
        cur = -1; n = 0;
    }
    __syncthreads();
    s1 += (size_t)blockIdx.x * n_tab; s2 += (size_t)blockIdx.x * n_tab; cnt += (size_t)blockIdx.x * n_seg;
    for (int i = tid; i < n_tab; i += 256) { s1[i] = t1[i]; s2[i] = t2[i]; }
    for (int i = tid; i < n_seg; i += 256) cnt[i] = tc[i];
}

// The two segment losses from the moments, in two launches instead of the ~20 element-wise / reduce launches the same
// arithmetic took as torch expressions on [n_seg] tensors (round 6: ~5 us of GPU time each, back to back on the critical path):
// (1) the private copies summed in copy order: sixteen groups of a workgroup take the copies g, g + 16, ... of 64 consecutive
// elements, their sums are added in group order (reproducible);
__global__ __launch_bounds__(1024) void seg_sum_copies_kernel(int k, int n1, int n_seg, unsigned b1, const double *__restrict__ s1c,
                                                              const double *__restrict__ s2c, const int32_t *__restrict__ cntc,
                                                              double *__restrict__ s1, double *__restrict__ s2,
                                                              int32_t *__restrict__ cnt)
{
    __shared__ double sm[2][16][64];
    const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (blockIdx.x >= b1) {  // counts
        const int e = (int)(blockIdx.x - b1) * 64 + l, ec = min(e, n_seg - 1);
        int t = 0;
        for (int cp = g; cp < k; cp += 16) t += cntc[(size_t)cp * n_seg + ec];
        sm[0][g][l] = (double)t;  // (exact: counts are below 2^31)
        __syncthreads();
        if (g == 0 && e < n_seg) {
            double r = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) r += sm[0][i][l];
            cnt[e] = (int32_t)r;
        }
        return;
    }
    const int e = (int)blockIdx.x * 64 + l, ec = min(e, n1 - 1);
    double a = 0.0, b = 0.0;
    for (int cp = g; cp < k; cp += 16) { a += s1c[(size_t)cp * n1 + ec]; b += s2c[(size_t)cp * n1 + ec]; }
    sm[0][g][l] = a; sm[1][g][l] = b;
    __syncthreads();
    if (g == 0 && e < n1) {
        double ra = 0.0, rb = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { ra += sm[0][i][l]; rb += sm[1][i][l]; }
        s1[e] = ra; s2[e] = rb;
    }
}

__device__ __forceinline__ double block_sum_1024(double v, double *sm16)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += sm16[i];
    return r;
}

// (2) one workgroup: the loss and the per-segment tables its backward gathers from.
//   mode 0, Scale_balance_loss (utils/loss_utils.py:32-57, mix_seg): mean over the PRESENT segments of the segment's mean;
//           coef[i] = 1 / (n_i K) for the backward (K = present segments, at least 1)
//   mode 1, scale_region_regulation_loss (:103-136): sum over segments of >= 2 pixels of n_i mean_c var_c (unbiased) / (H W);
//           mean[i][c] and coef[i] = 2 n_i / ((n_i - 1) c H W) for the backward
__global__ __launch_bounds__(1024) void seg_loss_finalize_kernel(int mode, int n_seg, int c, double hw, const double *__restrict__ s1,
                                                                 const double *__restrict__ s2, const int32_t *__restrict__ cnt,
                                                                 float *__restrict__ loss, float *__restrict__ coef,
                                                                 float *__restrict__ mean)
{
    __shared__ double sm16[16];
    const int tid = threadIdx.x;
    if (mode == 0) {
        double present = 0.0, msum = 0.0;
        for (int i = tid; i < n_seg; i += 1024) {
            const int n = cnt[i];
            if (n > 0) { present += 1.0; msum += s1[i] / (double)n; }
        }
        const double K = fmax(block_sum_1024(present, sm16), 1.0);
        const double total = block_sum_1024(msum, sm16);
        for (int i = tid; i < n_seg; i += 1024) {
            const int n = cnt[i];
            coef[i] = n > 0 ? (float)(1.0 / ((double)n * K)) : 0.f;
        }
        if (tid == 0) loss[0] = (float)(total / K);
        return;
    }
    double acc = 0.0;
    for (int i = tid; i < n_seg; i += 1024) {
        const int n = cnt[i];
        const bool ok = n >= 2;  // segments of 0 or 1 pixels are skipped (loss_utils.py:124-125)
        const double nn = ok ? (double)n : 2.0;
        double vs = 0.0;
        for (int ch = 0; ch < c; ++ch) {
            const double m = s1[(size_t)i * c + ch] / nn;
            // unbiased, as torch.var; the moments arrive accurately summed in double, and a variance is never negative
            vs += fmax((s2[(size_t)i * c + ch] - nn * m * m) / (nn - 1.0), 0.0);
            mean[(size_t)i * c + ch] = (float)m;
        }
        if (ok) acc += nn * (vs / (double)c);
        coef[i] = ok ? (float)(2.0 * nn / ((nn - 1.0) * (double)c * hw)) : 0.f;
    }
    const double total = block_sum_1024(acc, sm16);
    if (tid == 0) loss[0] = (float)(total / hw);
}

// pixel-major [n_pix, c], c % 4 == 0: one lane per float4, consecutive lanes on consecutive 16 bytes of the tensor
__global__ __launch_bounds__(256) void region_var_bwd_pm_kernel(int64_t n_pix, int c, const float *__restrict__ x,
                                                                const float *__restrict__ seg, int n_seg,
                                                                const float *__restrict__ mean, const float *__restrict__ coef,
                                                                float *__restrict__ vx, const float *__restrict__ add)
{
    const int q4 = c >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_pix * q4) return;
    const int64_t p = t / q4;
    const int ch = (int)(t - p * q4) * 4;
    const float f = seg[p];
    const int id = (f >= 0.f && f < (float)n_seg) ? (int)f : -1;
    const float4 xv = *reinterpret_cast<const float4 *>(x + (size_t)p * c + ch);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id >= 0) {
        const float k = coef[id];
        const float4 mv = *reinterpret_cast<const float4 *>(mean + (size_t)id * c + ch);
        o = make_float4(k * (xv.x - mv.x), k * (xv.y - mv.y), k * (xv.z - mv.z), k * (xv.w - mv.w));
    }
    if (add) {  // another consumer's gradient of the same map, added here instead of by a separate pass over both tensors
        const float4 g = *reinterpret_cast<const float4 *>(add + (size_t)p * c + ch);
        o = make_float4(o.x + g.x, o.y + g.y, o.z + g.z, o.w + g.w);
    }
    *reinterpret_cast<float4 *>(vx + (size_t)p * c + ch) = o;
}

__global__ __launch_bounds__(256) void region_var_bwd_kernel(int64_t n_pix, int c, const float *__restrict__ x,
                                                             const float *__restrict__ seg, int n_seg,
                                                             const float *__restrict__ mean, const float *__restrict__ coef,
                                                             float *__restrict__ vx, int pm)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    const float f = seg[p];
    const int id = (f >= 0.f && f < (float)n_seg) ? (int)f : -1;
    const float k = id >= 0 ? coef[id] : 0.f;
    for (int ch = 0; ch < c; ++ch) {
        const size_t o = pm ? (size_t)p * c + ch : (size_t)ch * n_pix + p;  // pixel-major [n_pix, c] or channel-major
        vx[o] = id >= 0 ? k * (x[o] - mean[(size_t)id * c + ch]) : 0.f;
    }
}

__global__ __launch_bounds__(256) void gather_seg_coef_kernel(int64_t n_pix, const float *__restrict__ seg, int n_seg,
                                                              const float *__restrict__ coef, float *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    const float f = seg[p];
    out[p] = (f >= 0.f && f < (float)n_seg) ? coef[(int)f] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// read_sam_clip_feature and the fused distillation L1 (scene/dataset_readers.py:54-121, train.py:165-166).
//
// A workgroup owns 64 consecutive pixels of the [H, W] map and walks the channels in blocks of 64.  Phase A (thread =
// pixel x 16-channel quarter): the level features F_l[c] = bilinear blend of the embedding rows of the pixel's (up to
// four) source pixels, rows read 64 contiguous bytes per thread; they go to LDS.  Phase B (thread = channel x pixel,
// pixels fastest): everything that touches the channel-major maps, coalesced along pixels.
constexpr int TP = 64;       // pixels per workgroup
constexpr int CB = 64;       // channels per block
constexpr int LDP = CB + 1;  // LDS row pitch (floats): conflict-free in both phases

struct Taps {
    int id[3][4];   // embedding row of (level, tap)
    float wgt[4];   // bilinear weights of the four taps (same for every level)
    float mask;     // 1 if all three levels have a segment at the nearest source pixel
};

__device__ __forceinline__ Taps make_taps(int p, int H, int W, int h, int w, int n_emb, const float *__restrict__ seg_map)
{
    Taps t;
    const int y = p / W, x = p - y * W;
    // torch upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1)
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    t.wgt[0] = (1.f - ly) * (1.f - lx); t.wgt[1] = (1.f - ly) * lx; t.wgt[2] = ly * (1.f - lx); t.wgt[3] = ly * lx;
    const int sp[4] = {y0 * w + x0, y0 * w + x1, y1 * w + x0, y1 * w + x1};
    // nearest resize of the validity mask: src = floor(dst * in / out)
    const int ny = min((int)floorf((float)y * ((float)h / (float)H)), h - 1);
    const int nx = min((int)floorf((float)x * ((float)w / (float)W)), w - 1);
    bool ok = true;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const float *lev = seg_map + (size_t)(l + 1) * h * w;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = (int)lev[sp[k]];
            t.id[l][k] = id < 0 ? id + n_emb : id;  // img_embed[-1] is the LAST row in the reference (Python indexing)
        }
        ok = ok && lev[ny * w + nx] != -1.0f;
    }
    t.mask = ok ? 1.f : 0.f;
    return t;
}

// F_l[ch] for 16 channels starting at c0 of one pixel, as torch computes it:
// h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11); the products by the lambdas are folded into wgt[] here
__device__ __forceinline__ void level_feature16(const Taps &t, int l, const float *__restrict__ img_embed, int c, int c0,
                                                float (&f)[16])
{
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (t.wgt[k] == 0.f) continue;  // identity resize: a single tap
        const float4 *row = reinterpret_cast<const float4 *>(img_embed + (size_t)t.id[l][k] * c + c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = row[q];
            f[4 * q] = fmaf(t.wgt[k], v.x, f[4 * q]); f[4 * q + 1] = fmaf(t.wgt[k], v.y, f[4 * q + 1]);
            f[4 * q + 2] = fmaf(t.wgt[k], v.z, f[4 * q + 2]); f[4 * q + 3] = fmaf(t.wgt[k], v.w, f[4 * q + 3]);
        }
    }
}

// MODE 0: feature_map + mask; 1: v_scale from v_feature; 2: fused L1 map forward; 3: fused L1 map backward
template <int MODE>
__global__ __launch_bounds__(256) void sam_feature_kernel(int c, int H, int W, int h, int w, int n_emb,
                                                          const float *__restrict__ pred /* 2,3: pred; 1: v_feature */,
                                                          const float *__restrict__ img_embed,
                                                          const float *__restrict__ seg_map,
                                                          const float *__restrict__ scale_map,
                                                          const float *__restrict__ v_map, float *__restrict__ out0,
                                                          float *__restrict__ out1)
{
    __shared__ float F[3][TP][LDP];
    __shared__ float red[4][TP][4];
    const int HW = H * W;
    const int p0 = blockIdx.x * TP;
    // phase-A identity: pixel pa, channel quarter qa
    const int pa = threadIdx.x >> 2, qa = threadIdx.x & 3;
    const int pA = min(p0 + pa, HW - 1);
    const Taps tp = make_taps(pA, H, W, h, w, n_emb, seg_map);
    // phase-B identity: pixel pb (fastest), channel residue rb
    const int pb = threadIdx.x & 63, rb = threadIdx.x >> 6;
    const int pB = p0 + pb;
    const bool inB = pB < HW;
    const int pBc = min(pB, HW - 1);
    float sc[3] = {0.f, 0.f, 0.f};
    if (MODE != 1) {
#pragma unroll
        for (int l = 0; l < 3; ++l) sc[l] = scale_map[(size_t)l * HW + pBc];
    }
    // the mask of pixel pb: computed by the phase-A threads of that pixel; hand it over through LDS
    if (qa == 0) red[0][pa][0] = tp.mask;
    __syncthreads();
    const float maskB = red[0][pb][0];
    __syncthreads();
    const float vB = (MODE == 3) ? v_map[pBc] * (1.0f / (float)c) : 0.f;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;  // MODE 2: |diff| sum in acc0; MODE 1 / 3: v_scale partials

    for (int cb = 0; cb < c; cb += CB) {
        // this block's 16 values of the channel-major operand are requested FIRST, all at once, and arrive while the
        // embedding rows are gathered (inside the loop below they would be waited for four at a time)
        float pv[CB / 4];
        if (MODE != 0) {
#pragma unroll
            for (int k = 0; k < CB / 4; ++k) pv[k] = pred[(size_t)min(cb + rb + 4 * k, c - 1) * HW + pBc];
            __builtin_amdgcn_sched_barrier(0);  // keep the requests up here (the scheduler would sink them to their uses)
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            float f[16];
            const int c0 = cb + qa * 16;
            if (c0 < c) level_feature16(tp, l, img_embed, c, c0, f);
#pragma unroll
            for (int j = 0; j < 16; ++j) F[l][pa][qa * 16 + j] = (c0 < c) ? f[j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CB / 4; ++k) {
            const int cl = rb + 4 * k, ch = cb + cl;
            const bool live = ch < c && inB;
            const float f0 = F[0][pb][cl], f1 = F[1][pb][cl], f2 = F[2][pb][cl];
            const size_t o = (size_t)min(ch, c - 1) * HW + pBc;
            if (MODE == 0) {
                // feature_map_s * scale_map[0] + feature_map_m * scale_map[1] + feature_map_l * scale_map[2]
                if (live) out0[o] = (f0 * sc[0] + f1 * sc[1]) + f2 * sc[2];
            } else if (MODE == 1) {
                const float v = live ? pv[k] : 0.f;
                acc0 = fmaf(v, f0, acc0); acc1 = fmaf(v, f1, acc1); acc2 = fmaf(v, f2, acc2);
            } else {
                const float gt = (f0 * sc[0] + f1 * sc[1]) + f2 * sc[2];
                const float diff = ch < c ? pv[k] * maskB - gt * maskB : 0.f;
                if (MODE == 2) {
                    acc0 += fabsf(diff);
                } else {
                    const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                    const float g = sgn * vB * maskB;  // d |pred*m - gt*m| / d pred, times v / c
                    if (live) out0[o] = g;
                    acc0 = fmaf(-g, f0, acc0); acc1 = fmaf(-g, f1, acc1); acc2 = fmaf(-g, f2, acc2);
                }
            }
        }
        __syncthreads();
    }
    if (MODE == 0) {
        if (rb == 0 && inB) out1[pB] = maskB;
        return;
    }
    red[rb][pb][0] = acc0; red[rb][pb][1] = acc1; red[rb][pb][2] = acc2;
    __syncthreads();
    if (rb == 0 && inB) {
        if (MODE == 2) {
            out0[pB] = ((red[0][pb][0] + red[1][pb][0]) + (red[2][pb][0] + red[3][pb][0])) / (float)c;
            out1[pB] = maskB;
        } else {
            float *vs = out1;
#pragma unroll
            for (int l = 0; l < 3; ++l)
                vs[(size_t)l * HW + pB] = (red[0][pb][l] + red[1][pb][l]) + (red[2][pb][l] + red[3][pb][l]);
        }
    }
}

// The fused distillation L1 on a PIXEL-major prediction pred[P][c] (what gags_decoder_head writes with layout 1: the
// [C,H,W] tensor the caller sees is a permuted view of it).  Nothing is transposed: thread = (pixel, 4 channels), a
// pixel's 2 KB row is read as consecutive float4 by consecutive lanes, the embedding rows likewise (L2-resident), the
// taps of the tile's 32 pixels are computed once and shared through LDS.  MODE 2: forward, 3: backward.
constexpr int TPM = 32;  // pixels per workgroup
struct TapsLds {
    int id[3][4];
    float wgt[4], mask, sc[3], v;
};

__device__ __forceinline__ float pm_wave_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int MODE>
__global__ __launch_bounds__(256) void sam_l1_pm_kernel(int c, int H, int W, int h, int w, int n_emb,
                                                        const float *__restrict__ pred, const float *__restrict__ img_embed,
                                                        const float *__restrict__ seg_map, const float *__restrict__ scale_map,
                                                        const float *__restrict__ v_map, float *__restrict__ out0,
                                                        float *__restrict__ out1)
{
    __shared__ TapsLds tl[TPM];
    __shared__ float red[TPM][4];  // per pixel: |diff| sum (MODE 2) or the three v_scale sums (MODE 3)
    const int HW = H * W;
    const int p0 = blockIdx.x * TPM;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < TPM) {
        const int pc = min(p0 + tid, HW - 1);
        const Taps t = make_taps(pc, H, W, h, w, n_emb, seg_map);
        TapsLds q;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q.id[l][k] = t.id[l][k];
            q.sc[l] = scale_map[(size_t)l * HW + pc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) q.wgt[k] = t.wgt[k];
        q.mask = t.mask;
        q.v = MODE == 3 ? v_map[pc] * (1.0f / (float)c) : 0.f;
        tl[tid] = q;
        red[tid][0] = red[tid][1] = red[tid][2] = red[tid][3] = 0.f;
    }
    __syncthreads();
    const int q4 = c >> 2;                 // float4 per pixel
    const int total = min(TPM, HW - p0) * q4;
    for (int i = tid; i - lane < total; i += 256) {  // (whole waves stay in the loop: the sums below are wave-wide)
        const bool live = i < total;
        const int ic = live ? i : total - 1;
        const int px = ic / q4, c4 = ic - px * q4;
        const TapsLds &t = tl[px];
        const size_t o = ((size_t)(p0 + px) * c) + 4 * c4;
        const float4 pv = *reinterpret_cast<const float4 *>(pred + o);
        float f[3][4];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            f[l][0] = f[l][1] = f[l][2] = f[l][3] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (t.wgt[k] == 0.f) continue;  // identity resize: a single tap
                const float4 e = *reinterpret_cast<const float4 *>(img_embed + (size_t)t.id[l][k] * c + 4 * c4);
                f[l][0] = fmaf(t.wgt[k], e.x, f[l][0]); f[l][1] = fmaf(t.wgt[k], e.y, f[l][1]);
                f[l][2] = fmaf(t.wgt[k], e.z, f[l][2]); f[l][3] = fmaf(t.wgt[k], e.w, f[l][3]);
            }
        }
        const float pe[4] = {pv.x, pv.y, pv.z, pv.w};
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, gq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gt = (f[0][q] * t.sc[0] + f[1][q] * t.sc[1]) + f[2][q] * t.sc[2];
            const float diff = live ? pe[q] * t.mask - gt * t.mask : 0.f;
            if (MODE == 2) {
                a0 += fabsf(diff);
            } else {
                const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                const float g = sgn * t.v * t.mask;
                gq[q] = g;
                a0 = fmaf(-g, f[0][q], a0); a1 = fmaf(-g, f[1][q], a1); a2 = fmaf(-g, f[2][q], a2);
            }
        }
        if (MODE == 3 && live) *reinterpret_cast<float4 *>(out0 + o) = make_float4(gq[0], gq[1], gq[2], gq[3]);
        // a wave covers 64 consecutive float4 of ONE pixel when c >= 256 (q4 % 64 == 0); otherwise lanes add themselves
        if ((q4 & 63) == 0) {
            a0 = pm_wave_sum(a0);
            if (MODE == 3) { a1 = pm_wave_sum(a1); a2 = pm_wave_sum(a2); }
            if (lane == 0) {
                atomicAdd(&red[px][0], a0);
                if (MODE == 3) { atomicAdd(&red[px][1], a1); atomicAdd(&red[px][2], a2); }
            }
        } else if (live) {
            atomicAdd(&red[px][0], a0);
            if (MODE == 3) { atomicAdd(&red[px][1], a1); atomicAdd(&red[px][2], a2); }
        }
    }
    __syncthreads();
    if (tid < TPM && p0 + tid < HW) {
        if (MODE == 2) {
            out0[p0 + tid] = red[tid][0] / (float)c;
            out1[p0 + tid] = tl[tid].mask;
        } else {
#pragma unroll
            for (int l = 0; l < 3; ++l) out1[(size_t)l * HW + p0 + tid] = red[tid][l];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// CNN_decoder's output head FUSED with the distillation L1 (train.py:159-166 in one kernel each way): from the last
// layer's fp32 logits x[P][ld] straight to l1_map = mean_c |normalize(x) m - gt m| (and back: from d l1_map to the
// bf16 gradient of the logits), without writing the normalised [512,H,W] map, reading it back for the loss, writing
// the loss's [512,H,W] gradient and reading that back for the head's backward: 8.5 + 12.7 GB per iteration at 1080p.
// Workgroup = 32 pixels; eight lanes per pixel, each holding 64 of its channels (16 float4) in registers, as in
// the pixel-major heads; the taps of the 32 pixels are computed once and shared through LDS.
constexpr int FHJ = 16;  // float4 per lane (ld <= 512)

typedef __bf16 l_bf16x2 __attribute__((ext_vector_type(2)));
typedef float l_f32x2 __attribute__((ext_vector_type(2)));

// c = ld = 512 (CNN_decoder(16, 512), the reference's configuration): 16 float4 per lane, straight-line code.
// ONE_TAP: the segmentation map has the render's resolution (identity resize: every pixel has exactly one source
// pixel), the common case -- one gather per level, the gathers of step j + 1 in flight during step j.
// DZM (BWD): the logits' gradient leaves as 0 = bf16 (the bf16 mode), 1 = fp32 (the fp32-tensor decoder tiers), 2 = IEEE half
// multiplied by the power of two dz_scale[0] and saturated at +-65504 (the f16 tier: csrc/half16.h).
struct GagsLossTrue { static constexpr bool value = true; };
struct GagsLossFalse { static constexpr bool value = false; };
template <bool BWD, bool ONE_TAP, int DZM = 0>
__global__ __launch_bounds__(256, 3) void head_distill_kernel(int H, int W, int h, int w, int n_emb,
                                                              const float *__restrict__ x, const float *__restrict__ img_embed,
                                                              const float *__restrict__ seg_map, const float *__restrict__ scale_map,
                                                              const float *__restrict__ v_map, float *__restrict__ l1_map,
                                                              float *__restrict__ mask_out, unsigned short *__restrict__ dz,
                                                              float *__restrict__ v_scale, const float *__restrict__ dz_scale = nullptr)
{
    constexpr int c = 512;
    if constexpr (BWD && DZM == 2) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");  // half conversions saturate
    __shared__ TapsLds tl[TPM];
    const int HW = H * W;
    const int p0 = blockIdx.x * TPM;
    const int tid = threadIdx.x;
    if (tid < TPM) {
        const int pc = min(p0 + tid, HW - 1);
        const Taps t = make_taps(pc, H, W, h, w, n_emb, seg_map);
        TapsLds q;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q.id[l][k] = t.id[l][k];
            q.sc[l] = scale_map[(size_t)l * HW + pc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) q.wgt[k] = t.wgt[k];
        q.mask = t.mask;
        q.v = BWD ? v_map[pc] * (1.0f / (float)c) : 0.f;
        tl[tid] = q;
    }
    const int px = tid >> 3, c0 = (tid & 7) * 4;
    const int pr = p0 + px, p = min(pr, HW - 1);
    float4 v[FHJ];
#pragma unroll
    for (int j = 0; j < FHJ; ++j) v[j] = *reinterpret_cast<const float4 *>(x + (size_t)p * c + c0 + 32 * j);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < FHJ; ++j) ss = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, ss))));
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize(dim=0), models/networks.py:192
    const float inv = 1.0f / nrm;
    __syncthreads();
    const TapsLds &t = tl[px];
    const float m = t.mask, s0 = t.sc[0], s1 = t.sc[1], s2 = t.sc[2], vm = t.v * t.mask;
    const float w0 = t.wgt[0], w1 = t.wgt[1], w2 = t.wgt[2], w3 = t.wgt[3];
    const float *er[3][4];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int k = 0; k < (ONE_TAP ? 1 : 4); ++k) er[l][k] = img_embed + (size_t)t.id[l][k] * c + c0;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, dot = 0.f;
    l_f32x2 a02 = {0.f, 0.f}, a12 = {0.f, 0.f}, a22 = {0.f, 0.f}, dot2 = {0.f, 0.f};  // BWD: packed partial sums
    unsigned sgn_pos[2] = {0u, 0u}, sgn_neg[2] = {0u, 0u};  // BWD: sign of diff per element (64 per lane)
    // BWD, signs: the fast pass takes copysign(1, diff) and shifts the sign bits into two words (one v_alignbit per element;
    // three instructions per element and pass instead of sixteen: 1.57 -> 1.36 ms at 1080p, round 6) and keeps the smallest
    // |diff| it met; torch.sign(0) = 0 matters for about one element in 10^7 (an exact tie of two fp32 values), and a pixel that
    // met one runs the pass again in the exact form (EXACT: signs in {-1, 0, +1} as two bit sets, as before).
    float minabs = 3.0e38f;
    auto pass1 = [&](auto exact_tag) __attribute__((always_inline)) {
    constexpr bool EXACT = decltype(exact_tag)::value;
    float4 en[3];
    if (ONE_TAP) {
#pragma unroll
        for (int l = 0; l < 3; ++l) en[l] = *reinterpret_cast<const float4 *>(er[l][0]);
    }
#pragma unroll
    for (int j = 0; j < FHJ; ++j) {
        float f[3][4];
        if (ONE_TAP) {
#pragma unroll
            for (int l = 0; l < 3; ++l) { f[l][0] = en[l].x; f[l][1] = en[l].y; f[l][2] = en[l].z; f[l][3] = en[l].w; }
            if (j + 1 < FHJ) {
#pragma unroll
                for (int l = 0; l < 3; ++l) en[l] = *reinterpret_cast<const float4 *>(er[l][0] + 32 * (j + 1));
            }
        } else {
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float4 e0 = *reinterpret_cast<const float4 *>(er[l][0] + 32 * j), e1 = *reinterpret_cast<const float4 *>(er[l][1] + 32 * j);
                const float4 e2 = *reinterpret_cast<const float4 *>(er[l][2] + 32 * j), e3 = *reinterpret_cast<const float4 *>(er[l][3] + 32 * j);
                // the taps in order 0..3, as level_feature16 accumulates them (a zero weight adds exactly nothing)
                f[l][0] = fmaf(w3, e3.x, fmaf(w2, e2.x, fmaf(w1, e1.x, w0 * e0.x)));
                f[l][1] = fmaf(w3, e3.y, fmaf(w2, e2.y, fmaf(w1, e1.y, w0 * e0.y)));
                f[l][2] = fmaf(w3, e3.z, fmaf(w2, e2.z, fmaf(w1, e1.z, w0 * e0.z)));
                f[l][3] = fmaf(w3, e3.w, fmaf(w2, e2.w, fmaf(w1, e1.w, w0 * e0.w)));
            }
        }
        const float xe[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        // the element-wise part on PACKED fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, two elements each: this kernel is
        // bound by its VALU issue slots, ~100 per step and lane, not by its 4.25 GB of logits); same operations, same order
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
            const l_f32x2 x2 = {xe[q], xe[q + 1]}, f0 = {f[0][q], f[0][q + 1]}, f1 = {f[1][q], f[1][q + 1]}, f2 = {f[2][q], f[2][q + 1]};
            const l_f32x2 y2 = x2 * inv;  // (x * (1 / n): within an ulp of F.normalize's x / n)
            const l_f32x2 gt2 = (f0 * s0 + f1 * s1) + f2 * s2;
            // (m is 0 or 1: y m - gt m = (y - gt) m exactly; the factor rides on the pixel's sum / on v m instead of on every element)
            const l_f32x2 d2 = y2 - gt2;
            if (!BWD) {
                a0 += fabsf(d2[0]);
                a0 += fabsf(d2[1]);
            } else {
                // sign(diff) in {-1, 0, +1} by integer arithmetic on the bits (a float compare per element keeps a lane
                // mask in an SGPR pair alive until the bit sets are assembled: 128 pairs, spilled)
                l_f32x2 sg2;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const unsigned u = __float_as_uint(d2[e]);
                    const int bit = 4 * j + q + e;
                    if constexpr (!EXACT) {
                        // the sign bit shifted into the word (v_alignbit: (word << 1) | (u >> 31); element i of a word ends
                        // at bit 31 - i), copysign(1, d) as the factor
                        sgn_neg[bit >> 5] = __builtin_amdgcn_alignbit(sgn_neg[bit >> 5], u, 31);
                        sg2[e] = __uint_as_float((u & 0x80000000u) | 0x3f800000u);
                    } else {
                        const unsigned neg = u >> 31, mag = min(u & 0x7fffffffu, 1u);
                        sgn_neg[bit >> 5] |= (mag & neg) << (bit & 31);
                        sgn_pos[bit >> 5] |= (mag & (neg ^ 1u)) << (bit & 31);
                        sg2[e] = __uint_as_float((u & 0x80000000u) | 0x3f800000u) * (float)mag;  // sign(diff)
                    }
                }
                if constexpr (!EXACT) minabs = fminf(minabs, fminf(fabsf(d2[0]), fabsf(d2[1])));
                const l_f32x2 gg2 = sg2 * vm;  // d l1 / d y = sign(diff) v m
                dot2 = __builtin_elementwise_fma(x2, gg2, dot2);  // (even / odd elements in the two halves, added at the end)
                a02 = __builtin_elementwise_fma(-gg2, f0, a02); a12 = __builtin_elementwise_fma(-gg2, f1, a12);
                a22 = __builtin_elementwise_fma(-gg2, f2, a22);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // one step at a time (unfenced, every gather of the pixel is hoisted: 256 VGPRs)
    }
    };
    pass1(GagsLossFalse{});
    bool exact_signs = false;
    if (BWD) {
        minabs = fminf(minabs, __shfl_xor(minabs, 1)); minabs = fminf(minabs, __shfl_xor(minabs, 2));
        minabs = fminf(minabs, __shfl_xor(minabs, 4));  // (the eight lanes of a pixel decide together: their sums are shared)
        if (minabs == 0.f && vm != 0.f) {
            exact_signs = true;
            a02 = a12 = a22 = dot2 = l_f32x2{0.f, 0.f};
            sgn_neg[0] = sgn_neg[1] = 0u;
            pass1(GagsLossTrue{});
        }
        a0 = a02[0] + a02[1]; a1 = a12[0] + a12[1]; a2 = a22[0] + a22[1]; dot = dot2[0] + dot2[1];
    }
    a0 += __shfl_xor(a0, 1); a0 += __shfl_xor(a0, 2); a0 += __shfl_xor(a0, 4);
    if (!BWD) {
        if ((tid & 7) == 0 && pr < HW) {
            l1_map[pr] = (a0 * m) / (float)c;
            mask_out[pr] = m;
        }
        return;
    }
    a1 += __shfl_xor(a1, 1); a1 += __shfl_xor(a1, 2); a1 += __shfl_xor(a1, 4);
    a2 += __shfl_xor(a2, 1); a2 += __shfl_xor(a2, 4); a2 += __shfl_xor(a2, 2);
    dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
    if (pr >= HW) return;
    if ((tid & 7) == 0) {
        v_scale[pr] = a0; v_scale[(size_t)HW + pr] = a1; v_scale[2 * (size_t)HW + pr] = a2;
    }
    // y = x / n:  dz = (g - y <y, g>) / n = g / n - x <x, g> / n^3;  g = sign(diff) v m, the signs kept as two bit sets
    float k1 = dot * inv * inv * inv, gmag = vm * inv;
    if constexpr (DZM == 2) { const float sc = dz_scale[0]; k1 *= sc; gmag *= sc; }  // (a power of two: exact)
    auto pass2 = [&](auto exact_tag) __attribute__((always_inline)) {
    constexpr bool EXACT = decltype(exact_tag)::value;
#pragma unroll
    for (int j = 0; j < FHJ; ++j) {
        const float xe[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        unsigned pk[2];
        float df[4];
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
            l_f32x2 sg2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int bit = 4 * j + q + e;
                if constexpr (!EXACT)
                    sg2[e] = __uint_as_float(((sgn_neg[bit >> 5] << (bit & 31)) & 0x80000000u) | 0x3f800000u);
                else
                    sg2[e] = (float)((int)((sgn_pos[bit >> 5] >> (bit & 31)) & 1u) - (int)((sgn_neg[bit >> 5] >> (bit & 31)) & 1u));
            }
            const l_f32x2 x2 = {xe[q], xe[q + 1]};
            const l_f32x2 dq2 = __builtin_elementwise_fma(-x2, l_f32x2{k1, k1}, sg2 * gmag);
            if constexpr (DZM == 2) {
                typedef _Float16 l_f16x2 __attribute__((ext_vector_type(2)));
                // (saturating at +-65504: MODE.FP16_OVFL is set at the top of this instantiation -- csrc/half16.h has the story)
                pk[q >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(dq2, l_f16x2));
            } else
            pk[q >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(dq2, l_bf16x2));
            df[q] = dq2[0]; df[q + 1] = dq2[1];
        }
        // (streaming stores: 2.1 GB that the input-gradient chain and the last layer's weight gradient read from HBM later)
        typedef float nt_f4 __attribute__((ext_vector_type(4)));
        typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
        if constexpr (DZM == 1)
            __builtin_nontemporal_store(nt_f4{df[0], df[1], df[2], df[3]},
                                        reinterpret_cast<nt_f4 *>(reinterpret_cast<float *>(dz) + (size_t)pr * c + c0 + 32 * j));
        else
            __builtin_nontemporal_store(nt_u2{pk[0], pk[1]}, reinterpret_cast<nt_u2 *>(dz + (size_t)pr * c + c0 + 32 * j));
    }
    };
    if (exact_signs) pass2(GagsLossTrue{});  // (rare: the pixel met an exact tie)
    else pass2(GagsLossFalse{});
}

// ---------------------------------------------------------------------------------------------------------------
// LERF relevancy (eval/openclip_encoder.py:42-56): one wave per pixel embedding, phrases in LDS.
constexpr int REL_MAX_PHRASES = 32;

__global__ __launch_bounds__(256) void relevancy_kernel(int64_t n_pix, int c, int n_pos, int n_neg,
                                                        const float *__restrict__ embed, const float *__restrict__ pos,
                                                        const float *__restrict__ neg, float *__restrict__ probs)
{
    extern __shared__ float ph[];  // [n_pos + n_neg][c]
    const int np = n_pos + n_neg;
    for (int i = threadIdx.x; i < np * c; i += 256) ph[i] = i < n_pos * c ? pos[i] : neg[i - n_pos * c];
    __syncthreads();
    __shared__ float sims[4][REL_MAX_PHRASES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 4 + wv;
    if (p >= n_pix) return;
    float e[16];  // this lane's share of the embedding (c <= 1024)
#pragma unroll
    for (int q = 0; q < 16; ++q) e[q] = (lane + 64 * q < c) ? embed[(size_t)p * c + lane + 64 * q] : 0.f;
    for (int j = 0; j < np; ++j) {
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (lane + 64 * q < c) d = fmaf(e[q], ph[j * c + lane + 64 * q], d);
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if (lane == 0) sims[wv][j] = d;
    }
    if (lane != 0) return;
    for (int j = 0; j < n_pos; ++j) {
        float best0 = 0.f, best1 = 0.f;
        for (int k = 0; k < n_neg; ++k) {
            const float a = 10.f * sims[wv][j], b = 10.f * sims[wv][n_pos + k];
            const float m = fmaxf(a, b);
            const float ea = expf(a - m), eb = expf(b - m);
            const float p0 = ea / (ea + eb), p1 = eb / (ea + eb);
            if (k == 0 || p0 < best0) { best0 = p0; best1 = p1; }  // argmin keeps the first minimum
        }
        probs[((size_t)j * n_pix + p) * 2] = best0;
        probs[((size_t)j * n_pix + p) * 2 + 1] = best1;
    }
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int gags_trained_seg(int h, int w, const float *seg_map, const float *scale_map, float *out, void *stream)
{
    GAGS_CLEAR_ERR();
    if (h <= 0 || w <= 0 || !seg_map || !scale_map || !out) return GAGS_EINVAL;
    hipLaunchKernelGGL(trained_seg_kernel, dim3((unsigned)((w + TSW - 1) / TSW), (unsigned)((h + TSH - 1) / TSH)), dim3(256), 0,
                       (hipStream_t)stream, h, w, seg_map, scale_map, out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_entropy_fwd(int64_t n, const float *s, double *acc, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || !acc || (n > 0 && !s)) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    const unsigned grid = (unsigned)(nblk(n) < 2048u ? nblk(n) : 2048u);
    hipLaunchKernelGGL(entropy_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, s, acc);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_entropy_bwd(int64_t n, const float *s, float v_over_n, float *v_s, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || (n > 0 && (!s || !v_s))) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    hipLaunchKernelGGL(entropy_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, n, s, v_over_n, v_s);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_entropy_bwd_dev(int64_t n, const float *s, const float *v, float *v_s, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || (n > 0 && (!s || !v || !v_s))) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    hipLaunchKernelGGL(entropy_bwd_dev_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, n, s, v, v_s);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_segment_stats_multi(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, int copies,
                                        double *s1, double *s2, int32_t *cnt, int layout, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || n_seg <= 0 || copies <= 0 || !s1 || !s2 || !cnt || (n_pix > 0 && (!x || !seg)) ||
        (layout != 0 && layout != 1))
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    const int vec = (n_pix % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(segment_stats_kernel, dim3(nblk((n_pix + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n_pix, c, x, seg, n_seg,
                       s1, s2, cnt, vec, copies, layout);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

namespace {
constexpr int64_t RUNS_LDS_MAX = 150 * 1024;
inline int64_t runs_lds_bytes(int c, int n_seg) { return (int64_t)n_seg * c * 16 + (int64_t)n_seg * 4; }
inline bool runs_shape(int c, int n_seg, int layout)
{
    return ((c == 16 && layout == 1) || c == 1) && n_seg > 0 && runs_lds_bytes(c, n_seg) <= RUNS_LDS_MAX;
}
}  // namespace

extern "C" int gags_segment_stats_runs_copies(int64_t n_pix, int c, int n_seg, int layout)
{
    if (n_pix <= 0 || !runs_shape(c, n_seg, layout)) return 0;
    const int64_t per = c == 16 ? 16 * 128 : 256 * 32;
    const int64_t need = (n_pix + per - 1) / per;
    return (int)(need < 512 ? need : 512);
}

extern "C" int gags_segment_stats_runs(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, int copies, double *s1,
                                       double *s2, int32_t *cnt, int layout, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix <= 0 || !runs_shape(c, n_seg, layout) || !x || !seg || !s1 || !s2 || !cnt ||
        copies != gags_segment_stats_runs_copies(n_pix, c, n_seg, layout))
        return GAGS_EINVAL;
    const int64_t lds = runs_lds_bytes(c, n_seg);
    static bool raised = false;  // (dynamic LDS above 64 KB is an opt-in per kernel)
    if (!raised) {
        if (hipFuncSetAttribute((const void *)segment_stats_runs_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)RUNS_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void *)segment_stats_runs_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)RUNS_LDS_MAX) != hipSuccess)
            return GAGS_ELAUNCH;
        raised = true;
    }
    if (c == 16)
        hipLaunchKernelGGL(segment_stats_runs_kernel<16>, dim3((unsigned)copies), dim3(256), (size_t)lds, (hipStream_t)stream, n_pix, x,
                           seg, n_seg, s1, s2, cnt);
    else
        hipLaunchKernelGGL(segment_stats_runs_kernel<1>, dim3((unsigned)copies), dim3(256), (size_t)lds, (hipStream_t)stream, n_pix, x,
                           seg, n_seg, s1, s2, cnt);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_segment_loss(int mode, int n_seg, int c, int copies, int64_t n_pix, const double *s1c, const double *s2c,
                                 const int32_t *cntc, double *s1, double *s2, int32_t *cnt, float *loss, float *coef, float *mean,
                                 void *stream)
{
    GAGS_CLEAR_ERR();
    if ((mode != 0 && mode != 1) || n_seg <= 0 || c <= 0 || copies <= 0 || n_pix <= 0 || (mode == 0 && c != 1) || !s1c || !s2c ||
        !cntc || !s1 || !s2 || !cnt || !loss || !coef || (mode == 1 && !mean))
        return GAGS_EINVAL;
    const int64_t n1 = (int64_t)n_seg * c;
    if (n1 > (1 << 24)) return GAGS_EINVAL;
    const unsigned b1 = (unsigned)((n1 + 63) / 64), b2 = (unsigned)((n_seg + 63) / 64);
    hipLaunchKernelGGL(seg_sum_copies_kernel, dim3(b1 + b2), dim3(1024), 0, (hipStream_t)stream, copies, (int)n1, n_seg, b1, s1c, s2c,
                       cntc, s1, s2, cnt);
    hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mode, n_seg, c, (double)n_pix, s1, s2,
                       cnt, loss, coef, mean);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_segment_stats(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, double *s1, double *s2,
                                  int32_t *cnt, void *stream)
{
    return gags_segment_stats_multi(n_pix, c, x, seg, n_seg, 1, s1, s2, cnt, 0, stream);
}

extern "C" int gags_region_var_bwd_add(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, const float *mean,
                                       const float *coef, const float *add, float *v_x, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || (c & 3) != 0 || n_seg <= 0 || (n_pix > 0 && (!x || !seg || !mean || !coef || !add || !v_x)) ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v_x) | reinterpret_cast<uintptr_t>(mean) |
          reinterpret_cast<uintptr_t>(add)) & 15) != 0)
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(region_var_bwd_pm_kernel, dim3(nblk(n_pix * (c >> 2))), dim3(256), 0, (hipStream_t)stream, n_pix, c, x, seg, n_seg,
                       mean, coef, v_x, add);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_region_var_bwd_layout(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, const float *mean,
                                          const float *coef, float *v_x, int layout, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || n_seg <= 0 || (n_pix > 0 && (!x || !seg || !mean || !coef || !v_x)) || (layout != 0 && layout != 1))
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    if (layout == 1 && (c & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v_x) | reinterpret_cast<uintptr_t>(mean)) & 15) == 0)
        hipLaunchKernelGGL(region_var_bwd_pm_kernel, dim3(nblk(n_pix * (c >> 2))), dim3(256), 0, (hipStream_t)stream, n_pix, c, x, seg,
                           n_seg, mean, coef, v_x, (const float *)nullptr);
    else
        hipLaunchKernelGGL(region_var_bwd_kernel, dim3(nblk(n_pix)), dim3(256), 0, (hipStream_t)stream, n_pix, c, x, seg, n_seg,
                           mean, coef, v_x, layout);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_region_var_bwd(int64_t n_pix, int c, const float *x, const float *seg, int n_seg, const float *mean,
                                   const float *coef, float *v_x, void *stream)
{
    return gags_region_var_bwd_layout(n_pix, c, x, seg, n_seg, mean, coef, v_x, 0, stream);
}

extern "C" int gags_gather_seg_coef(int64_t n_pix, const float *seg, int n_seg, const float *coef, float *out, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || n_seg <= 0 || (n_pix > 0 && (!seg || !coef || !out))) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(gather_seg_coef_kernel, dim3(nblk(n_pix)), dim3(256), 0, (hipStream_t)stream, n_pix, seg, n_seg, coef,
                       out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

namespace {
inline bool sam_args_ok(int c, int H, int W, int h, int w, int n_emb)
{
    return c > 0 && c % 16 == 0 && H > 0 && W > 0 && h > 0 && w > 0 && n_emb > 0;
}
}  // namespace

extern "C" int gags_sam_clip_feature(int c, int H, int W, int h, int w, int n_emb, const float *img_embed,
                                     const float *seg_map, const float *scale_map, float *feature_map, float *mask,
                                     void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || !img_embed || !seg_map || !scale_map || !feature_map || !mask) return GAGS_EINVAL;
    hipLaunchKernelGGL(sam_feature_kernel<0>, dim3((H * W + TP - 1) / TP), dim3(256), 0, (hipStream_t)stream, c, H, W, h, w,
                       n_emb, (const float *)nullptr, img_embed, seg_map, scale_map, (const float *)nullptr, feature_map, mask);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_sam_clip_feature_bwd_scale(int c, int H, int W, int h, int w, int n_emb, const float *img_embed,
                                               const float *seg_map, const float *v_feature, float *v_scale, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || !img_embed || !seg_map || !v_feature || !v_scale) return GAGS_EINVAL;
    hipLaunchKernelGGL(sam_feature_kernel<1>, dim3((H * W + TP - 1) / TP), dim3(256), 0, (hipStream_t)stream, c, H, W, h, w,
                       n_emb, v_feature, img_embed, seg_map, (const float *)nullptr, (const float *)nullptr,
                       (float *)nullptr, v_scale);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_distill_l1_map_fwd(int c, int H, int W, int h, int w, int n_emb, const float *pred, const float *img_embed,
                                       const float *seg_map, const float *scale_map, float *l1_map, float *mask, int layout,
                                       void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || !pred || !img_embed || !seg_map || !scale_map || !l1_map || !mask ||
        (layout != 0 && layout != 1))
        return GAGS_EINVAL;
    if (layout == 1) {  // pred is [H, W, c]
        hipLaunchKernelGGL(sam_l1_pm_kernel<2>, dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, c, H, W, h, w,
                           n_emb, pred, img_embed, seg_map, scale_map, (const float *)nullptr, l1_map, mask);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    hipLaunchKernelGGL(sam_feature_kernel<2>, dim3((H * W + TP - 1) / TP), dim3(256), 0, (hipStream_t)stream, c, H, W, h, w,
                       n_emb, pred, img_embed, seg_map, scale_map, (const float *)nullptr, l1_map, mask);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_distill_l1_map_bwd(int c, int H, int W, int h, int w, int n_emb, const float *pred, const float *img_embed,
                                       const float *seg_map, const float *scale_map, const float *v_map, float *v_pred,
                                       float *v_scale, int layout, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || !pred || !img_embed || !seg_map || !scale_map || !v_map || !v_pred || !v_scale ||
        (layout != 0 && layout != 1))
        return GAGS_EINVAL;
    if (layout == 1) {  // pred and v_pred are [H, W, c]
        hipLaunchKernelGGL(sam_l1_pm_kernel<3>, dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, c, H, W, h, w,
                           n_emb, pred, img_embed, seg_map, scale_map, v_map, v_pred, v_scale);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    hipLaunchKernelGGL(sam_feature_kernel<3>, dim3((H * W + TP - 1) / TP), dim3(256), 0, (hipStream_t)stream, c, H, W, h, w,
                       n_emb, pred, img_embed, seg_map, scale_map, v_map, v_pred, v_scale);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_decoder_head_distill_fwd(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                             const float *img_embed, const float *seg_map, const float *scale_map,
                                             float *l1_map, float *mask, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || c != 512 || ld != 512 || !x || !img_embed || !seg_map || !scale_map || !l1_map || !mask)
        return GAGS_EINVAL;  // (the reference's CNN_decoder(16, 512); other widths take the two-step route)
    if (H == h && W == w)
        hipLaunchKernelGGL((head_distill_kernel<false, true>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H,
                           W, h, w, n_emb, x, img_embed, seg_map, scale_map, (const float *)nullptr, l1_map, mask,
                           (unsigned short *)nullptr, (float *)nullptr);
    else
        hipLaunchKernelGGL((head_distill_kernel<false, false>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H,
                           W, h, w, n_emb, x, img_embed, seg_map, scale_map, (const float *)nullptr, l1_map, mask,
                           (unsigned short *)nullptr, (float *)nullptr);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_decoder_head_distill_bwd(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                             const float *img_embed, const float *seg_map, const float *scale_map,
                                             const float *v_map, void *dz_bf16, float *v_scale, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || c != 512 || ld != 512 || !x || !img_embed || !seg_map || !scale_map || !v_map ||
        !dz_bf16 || !v_scale)
        return GAGS_EINVAL;
    if (H == h && W == w)
        hipLaunchKernelGGL((head_distill_kernel<true, true>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H, W,
                           h, w, n_emb, x, img_embed, seg_map, scale_map, v_map, (float *)nullptr, (float *)nullptr,
                           (unsigned short *)dz_bf16, v_scale);
    else
        hipLaunchKernelGGL((head_distill_kernel<true, false>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H, W,
                           h, w, n_emb, x, img_embed, seg_map, scale_map, v_map, (float *)nullptr, (float *)nullptr,
                           (unsigned short *)dz_bf16, v_scale);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_decoder_head_distill_bwd_h16(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                                 const float *img_embed, const float *seg_map, const float *scale_map,
                                                 const float *v_map, void *dz_f16, const float *dz_scale, float *v_scale,
                                                 void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || c != 512 || ld != 512 || !x || !img_embed || !seg_map || !scale_map || !v_map ||
        !dz_f16 || !dz_scale || !v_scale)
        return GAGS_EINVAL;
    if (H == h && W == w)
        hipLaunchKernelGGL((head_distill_kernel<true, true, 2>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H,
                           W, h, w, n_emb, x, img_embed, seg_map, scale_map, v_map, (float *)nullptr, (float *)nullptr,
                           (unsigned short *)dz_f16, v_scale, dz_scale);
    else
        hipLaunchKernelGGL((head_distill_kernel<true, false, 2>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H,
                           W, h, w, n_emb, x, img_embed, seg_map, scale_map, v_map, (float *)nullptr, (float *)nullptr,
                           (unsigned short *)dz_f16, v_scale, dz_scale);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_decoder_head_distill_bwd_f32(int c, int ld, int H, int W, int h, int w, int n_emb, const float *x,
                                                 const float *img_embed, const float *seg_map, const float *scale_map,
                                                 const float *v_map, float *dz, float *v_scale, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!sam_args_ok(c, H, W, h, w, n_emb) || c != 512 || ld != 512 || !x || !img_embed || !seg_map || !scale_map || !v_map ||
        !dz || !v_scale)
        return GAGS_EINVAL;
    if (H == h && W == w)
        hipLaunchKernelGGL((head_distill_kernel<true, true, 1>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H,
                           W, h, w, n_emb, x, img_embed, seg_map, scale_map, v_map, (float *)nullptr, (float *)nullptr,
                           (unsigned short *)dz, v_scale);
    else
        hipLaunchKernelGGL((head_distill_kernel<true, false, 1>), dim3((H * W + TPM - 1) / TPM), dim3(256), 0, (hipStream_t)stream, H,
                           W, h, w, n_emb, x, img_embed, seg_map, scale_map, v_map, (float *)nullptr, (float *)nullptr,
                           (unsigned short *)dz, v_scale);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_relevancy(int64_t n_pix, int c, int n_pos, int n_neg, const float *embed, const float *pos,
                              const float *neg, float *probs, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || c <= 0 || n_pos <= 0 || n_neg <= 0 || n_pos + n_neg > REL_MAX_PHRASES || c > 1024 ||
        (size_t)(n_pos + n_neg) * c * 4 > 60000 || !pos || !neg || (n_pix > 0 && (!embed || !probs)))
        return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    hipLaunchKernelGGL(relevancy_kernel, dim3((unsigned)((n_pix + 3) / 4)), dim3(256), (size_t)(n_pos + n_neg) * c * 4,
                       (hipStream_t)stream, n_pix, c, n_pos, n_neg, embed, pos, neg, probs);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
