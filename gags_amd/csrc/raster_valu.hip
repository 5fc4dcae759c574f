// K9 / K10, VALU flavour: one lane per pixel, CDIM accumulators per lane.
// This is the path for narrow feature widths (D = 3, 4, 16: RGB, RGB+ED and the reference's
// default 16-d semantic feature, train.py:68) and the fallback for widths the MFMA kernels do
// not take.  For D > CDIM the channel chunks run as the second grid dimension of ONE launch.
//
// Workgroup = one 16x16 tile, 4 waves; wave w owns the 8x8 pixel block (w&1, w>>1) so that a
// wave-level "nobody is hit" test skips as many Gaussians as possible.  Gaussians of the tile's
// depth-sorted range are staged through LDS in batches of 256 (xy, conic, opacity, id).
// Feature rows are wave-uniform: they are fetched with scalar loads (SGPR operands of v_fmac),
// not through LDS.  HBM/LDS-latency bound at these widths.
#include "common.h"

namespace {

constexpr int BATCH = 256;

struct GaussLds {
    float x, y, a, b, c, o;
};

__device__ __forceinline__ void pixel_of_thread(int tile, int tile_w, int &pi, int &pj)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int ty = tile / tile_w, tx = tile - ty * tile_w;
    pj = tx * GAGS_TILE + (w & 1) * 8 + (l & 7);
    pi = ty * GAGS_TILE + (w >> 1) * 8 + (l >> 3);
}

template <int CDIM>
__global__ __launch_bounds__(256) void raster_fwd_valu(
    int d, int width, int height, int tile_w, int n_tiles, const float *__restrict__ means2d,
    const float *__restrict__ conics, const float *__restrict__ opacities, const float *__restrict__ colors,
    const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int n_isects, float *__restrict__ render_colors,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids)
{
    __shared__ GaussLds gs[BATCH];
    __shared__ int32_t ids[BATCH];

    const int tile = gags_xcd_remap(blockIdx.x, n_tiles);
    const int ch0 = blockIdx.y * CDIM;
    const int nch = min(CDIM, d - ch0);
    int pi, pj;
    pixel_of_thread(tile, tile_w, pi, pj);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;

    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;

    float acc[CDIM];
#pragma unroll
    for (int k = 0; k < CDIM; ++k) acc[k] = 0.f;
    float T = 1.0f;
    int cur = 0;
    bool done = !inside;

    for (int b0 = start; b0 < end; b0 += BATCH) {
        if (__syncthreads_count(done) >= 256) break;
        const int idx = b0 + threadIdx.x;
        if (idx < end) {
            const int g = flatten_ids[idx];
            ids[threadIdx.x] = g;
            const float2 m = reinterpret_cast<const float2 *>(means2d)[g];
            GaussLds r;
            r.x = m.x; r.y = m.y;
            r.a = conics[3 * g]; r.b = conics[3 * g + 1]; r.c = conics[3 * g + 2];
            r.o = opacities[g];
            gs[threadIdx.x] = r;
        }
        __syncthreads();
        const int bs = min(BATCH, end - b0);
        for (int t = 0; t < bs; ++t) {
            const GaussLds r = gs[t];
            const float dx = r.x - px, dy = r.y - py;
            const float sigma = 0.5f * (r.a * dx * dx + r.c * dy * dy) + r.b * dx * dy;
            const float alpha = fminf(GAGS_ALPHA_MAX, r.o * gags_exp_neg(sigma));
            bool hit = !done && !(sigma < 0.f || alpha < GAGS_ALPHA_MIN);
            const float next_T = T * (1.0f - alpha);
            if (hit && next_T <= GAGS_T_STOP) {
                done = true;
                hit = false;
            }
            if (!__any(hit)) continue;
            const float vis = hit ? alpha * T : 0.f;
            const int g = __builtin_amdgcn_readfirstlane(ids[t]);
            const float *__restrict__ c = colors + (size_t)g * d + ch0;
#pragma unroll
            for (int k = 0; k < CDIM; ++k) acc[k] = __builtin_fmaf(c[k < nch ? k : 0], vis, acc[k]);
            if (hit) {
                cur = b0 + t;
                T = next_T;
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)pi * width + pj;
        if (blockIdx.y == 0) {
            render_alphas[pix] = 1.0f - T;
            last_ids[pix] = cur;
        }
        float *o = render_colors + pix * d + ch0;
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            if (k < nch) o[k] = backgrounds ? __builtin_fmaf(T, backgrounds[ch0 + k], acc[k]) : acc[k];
    }
}

// wave64 sum by DPP: quad swaps, half-row / row mirrors, then row broadcasts; total lands in
// lane 63 and is returned wave-uniform (SGPR) through readlane.
__device__ __forceinline__ float dpp_f(float v, int ctrl, int row_mask)
{
    int r;
    switch (ctrl) {  // ctrl must be a literal for the builtin
        case 0xB1: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false); break;
        case 0x4E: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false); break;
        case 0x141: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false); break;
        case 0x140: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false); break;
        case 0x142: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false); break;
        default: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false); break;
    }
    (void)row_mask;
    return __int_as_float(r);
}

__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f(v, 0xB1, 0xF);
    v += dpp_f(v, 0x4E, 0xF);
    v += dpp_f(v, 0x141, 0xF);
    v += dpp_f(v, 0x140, 0xF);
    v += dpp_f(v, 0x142, 0xA);
    v += dpp_f(v, 0x143, 0xC);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ void atomic_add_f32(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int CDIM, bool GEOM>
__global__ __launch_bounds__(256) void raster_bwd_valu(
    int d, int width, int height, int tile_w, int n_tiles, const float *__restrict__ means2d,
    const float *__restrict__ conics, const float *__restrict__ opacities, const float *__restrict__ colors,
    const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int n_isects, const float *__restrict__ render_alphas,
    const int32_t *__restrict__ last_ids, const float *__restrict__ v_render_colors,
    const float *__restrict__ v_render_alphas, float *__restrict__ v_colors, float *__restrict__ v_opacities,
    float *__restrict__ v_means2d, float *__restrict__ v_conics)
{
    __shared__ GaussLds gs[BATCH];
    __shared__ int32_t ids[BATCH];

    const int tile = gags_xcd_remap(blockIdx.x, n_tiles);
    const int ch0 = blockIdx.y * CDIM;
    const int nch = min(CDIM, d - ch0);
    const int lane = threadIdx.x & 63;
    int pi, pj;
    pixel_of_thread(tile, tile_w, pi, pj);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const size_t pix = inside ? (size_t)pi * width + pj : 0;

    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;

    const float T_final = inside ? 1.0f - render_alphas[pix] : 1.0f;
    float T = T_final;
    const int bin_final = inside ? last_ids[pix] : -1;
    float vc[CDIM], buf[CDIM];
    float bg_dot = 0.f;
#pragma unroll
    for (int k = 0; k < CDIM; ++k) {
        vc[k] = (inside && k < nch) ? v_render_colors[pix * d + ch0 + k] : 0.f;
        buf[k] = 0.f;
        if (GEOM && backgrounds && k < nch) bg_dot += backgrounds[ch0 + k] * vc[k];
    }
    const float va = (GEOM && inside && blockIdx.y == 0 && v_render_alphas) ? v_render_alphas[pix] : 0.f;

    // wave-level last contributing index: nothing beyond it needs loading by this wave
    int wave_last = bin_final;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, o, 64));

    // walk the range back to front in batches of 256; batch b covers [hi-255, hi]
    for (int hi = end - 1; hi >= start; hi -= BATCH) {
        __syncthreads();
        const int idx = hi - (int)threadIdx.x;
        if (idx >= start) {
            const int g = flatten_ids[idx];
            ids[threadIdx.x] = g;
            const float2 m = reinterpret_cast<const float2 *>(means2d)[g];
            GaussLds r;
            r.x = m.x; r.y = m.y;
            r.a = conics[3 * g]; r.b = conics[3 * g + 1]; r.c = conics[3 * g + 2];
            r.o = opacities[g];
            gs[threadIdx.x] = r;
        }
        __syncthreads();
        const int bs = min(BATCH, hi + 1 - start);
        for (int t = max(0, hi - wave_last); t < bs; ++t) {
            const int s = hi - t;  // sorted index
            const GaussLds r = gs[t];
            const float dx = r.x - px, dy = r.y - py;
            const float sigma = 0.5f * (r.a * dx * dx + r.c * dy * dy) + r.b * dx * dy;
            const float vis = gags_exp_neg(sigma);
            const float alpha = fminf(GAGS_ALPHA_MAX, r.o * vis);
            const bool valid = (s <= bin_final) && !(sigma < 0.f || alpha < GAGS_ALPHA_MIN);
            if (!__any(valid)) continue;
            const int g = __builtin_amdgcn_readfirstlane(ids[t]);
            const float *__restrict__ c = colors + (size_t)g * d + ch0;

            float fac = 0.f, ra = 1.f, v_alpha = 0.f;
            if (valid) {
                ra = 1.0f / (1.0f - alpha);
                T *= ra;
                fac = alpha * T;
            }
            float mine = 0.f;  // lane k ends up owning the wave total of channel k
#pragma unroll
            for (int k = 0; k < CDIM; ++k) {
                const float tot = wave_sum(fac * vc[k]);
                mine = (lane == k) ? tot : mine;
                if (GEOM) {
                    const float ck = c[k < nch ? k : 0];
                    if (valid) {
                        v_alpha += (ck * T - buf[k] * ra) * vc[k];
                        buf[k] += ck * fac;
                    }
                }
            }
            if (lane < nch) atomic_add_f32(v_colors + (size_t)g * d + ch0 + lane, mine);

            if (GEOM) {
                float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f, g5 = 0.f;
                if (valid) {
                    v_alpha += T_final * ra * va;
                    if (backgrounds) v_alpha += -T_final * ra * bg_dot;
                    if (r.o * vis <= GAGS_ALPHA_MAX) {
                        const float v_sigma = -r.o * vis * v_alpha;
                        g0 = 0.5f * v_sigma * dx * dx;
                        g1 = v_sigma * dx * dy;
                        g2 = 0.5f * v_sigma * dy * dy;
                        g3 = v_sigma * (r.a * dx + r.b * dy);
                        g4 = v_sigma * (r.b * dx + r.c * dy);
                        g5 = vis * v_alpha;
                    }
                }
                g0 = wave_sum(g0); g1 = wave_sum(g1); g2 = wave_sum(g2);
                g3 = wave_sum(g3); g4 = wave_sum(g4); g5 = wave_sum(g5);
                if (lane < 3) atomic_add_f32(v_conics + 3 * (size_t)g + lane, lane == 0 ? g0 : (lane == 1 ? g1 : g2));
                else if (lane < 5) atomic_add_f32(v_means2d + 2 * (size_t)g + (lane - 3), lane == 3 ? g3 : g4);
                else if (lane == 5) atomic_add_f32(v_opacities + g, g5);
            }
        }
    }
}


// Diagnostics for the roofline model: counts[0] += (pixel,Gaussian) pairs evaluated
// before each pixel's stop, counts[1] += pairs actually blended.  Same walk as the forward.
__global__ __launch_bounds__(256) void raster_stats_kernel(
    int width, int height, int tile_w, int n_tiles, const float *__restrict__ means2d,
    const float *__restrict__ conics, const float *__restrict__ opacities, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int n_isects, unsigned long long *__restrict__ counts)
{
    __shared__ GaussLds gs[BATCH];
    const int tile = blockIdx.x;
    int pi, pj;
    pixel_of_thread(tile, tile_w, pi, pj);
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    float T = 1.0f;
    bool done = !inside;
    unsigned n_eval = 0, n_blend = 0;
    for (int b0 = start; b0 < end; b0 += BATCH) {
        if (__syncthreads_count(done) >= 256) break;
        const int idx = b0 + threadIdx.x;
        if (idx < end) {
            const int g = flatten_ids[idx];
            const float2 m = reinterpret_cast<const float2 *>(means2d)[g];
            GaussLds r;
            r.x = m.x; r.y = m.y;
            r.a = conics[3 * g]; r.b = conics[3 * g + 1]; r.c = conics[3 * g + 2];
            r.o = opacities[g];
            gs[threadIdx.x] = r;
        }
        __syncthreads();
        const int bs = min(BATCH, end - b0);
        for (int t = 0; t < bs; ++t) {
            if (done) continue;
            const GaussLds r = gs[t];
            const float dx = r.x - px, dy = r.y - py;
            const float sigma = 0.5f * (r.a * dx * dx + r.c * dy * dy) + r.b * dx * dy;
            const float alpha = fminf(GAGS_ALPHA_MAX, r.o * gags_exp_neg(sigma));
            ++n_eval;
            if (sigma < 0.f || alpha < GAGS_ALPHA_MIN) continue;
            const float next_T = T * (1.0f - alpha);
            if (next_T <= GAGS_T_STOP) { done = true; continue; }
            ++n_blend;
            T = next_T;
        }
    }
    unsigned long long e = n_eval, b = n_blend;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_xor(e, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], e);
        atomicAdd(&counts[1], b);
    }
}

template <int CDIM>
int launch_fwd(int d, int width, int height, const float *means2d, const float *conics, const float *opacities,
               const float *colors, const float *backgrounds, const int32_t *offsets, const int32_t *flat,
               int n_isects, float *out, float *alphas, int32_t *last_ids, hipStream_t st)
{
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h;
    dim3 grid(n_tiles, (d + CDIM - 1) / CDIM);
    hipLaunchKernelGGL(raster_fwd_valu<CDIM>, grid, dim3(256), 0, st, d, width, height, tile_w, n_tiles, means2d,
                       conics, opacities, colors, backgrounds, offsets, flat, n_isects, out, alphas, last_ids);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

template <int CDIM, bool GEOM>
int launch_bwd(int d, int width, int height, const float *means2d, const float *conics, const float *opacities,
               const float *colors, const float *backgrounds, const int32_t *offsets, const int32_t *flat,
               int n_isects, const float *alphas, const int32_t *last_ids, const float *v_out, const float *v_alpha,
               float *v_colors, float *v_opac, float *v_m2d, float *v_con, hipStream_t st)
{
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h;
    dim3 grid(n_tiles, (d + CDIM - 1) / CDIM);
    hipLaunchKernelGGL((raster_bwd_valu<CDIM, GEOM>), grid, dim3(256), 0, st, d, width, height, tile_w, n_tiles,
                       means2d, conics, opacities, colors, backgrounds, offsets, flat, n_isects, alphas, last_ids,
                       v_out, v_alpha, v_colors, v_opac, v_m2d, v_con);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

}  // namespace

// internal entry points used by api.hip's dispatcher
int gags_raster_fwd_valu(int d, int width, int height, const float *means2d, const float *conics,
                         const float *opacities, const float *colors, const float *backgrounds,
                         const int32_t *offsets, const int32_t *flat, int n_isects, float *out, float *alphas,
                         int32_t *last_ids, hipStream_t st)
{
    GAGS_CLEAR_ERR();
#define ARGS d, width, height, means2d, conics, opacities, colors, backgrounds, offsets, flat, n_isects, out, alphas, last_ids, st
    if (d <= 4) return launch_fwd<4>(ARGS);
    if (d <= 16) return launch_fwd<16>(ARGS);
    return launch_fwd<32>(ARGS);
#undef ARGS
}

int gags_raster_bwd_valu(int d, int width, int height, const float *means2d, const float *conics,
                         const float *opacities, const float *colors, const float *backgrounds,
                         const int32_t *offsets, const int32_t *flat, int n_isects, const float *alphas,
                         const int32_t *last_ids, const float *v_out, const float *v_alpha, float *v_colors,
                         float *v_opac, float *v_m2d, float *v_con, bool geom, hipStream_t st)
{
    GAGS_CLEAR_ERR();
#define ARGS d, width, height, means2d, conics, opacities, colors, backgrounds, offsets, flat, n_isects, alphas, last_ids, v_out, v_alpha, v_colors, v_opac, v_m2d, v_con, st
    if (geom) {
        if (d <= 4) return launch_bwd<4, true>(ARGS);
        if (d <= 16) return launch_bwd<16, true>(ARGS);
        return launch_bwd<32, true>(ARGS);
    }
    if (d <= 4) return launch_bwd<4, false>(ARGS);
    if (d <= 16) return launch_bwd<16, false>(ARGS);
    return launch_bwd<32, false>(ARGS);
#undef ARGS
}

extern "C" int gags_raster_stats(int width, int height, const float *means2d, const float *conics,
                                 const float *opacities, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                 int64_t n_isects, int64_t *counts, void *stream)
{
    GAGS_CLEAR_ERR();
    if (width <= 0 || height <= 0 || n_isects < 0 || n_isects >= (1ll << 31) || !isect_offsets || !counts)
        return GAGS_EINVAL;
    if (n_isects == 0) return GAGS_OK;
    if (!means2d || !conics || !opacities || !flatten_ids) return GAGS_EINVAL;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    hipLaunchKernelGGL(raster_stats_kernel, dim3(tile_w * tile_h), dim3(256), 0, (hipStream_t)stream, width, height,
                       tile_w, tile_w * tile_h, means2d, conics, opacities, isect_offsets, flatten_ids, (int)n_isects,
                       (unsigned long long *)counts);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
