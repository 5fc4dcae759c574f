// K8b + the WEIGHTS pass of the matrix-core rasterizer.
//
// raster_weights: one wave per (tile, 8x8 pixel block), one PIXEL per lane, one hit at a time (round 6; before: a pixel pair per
// lane and two hits per step).  It does ALL the per-(pixel, Gaussian)
// scalar work of the view exactly once -- alpha, skip rule, transmittance chain, stop rule --
// independent of the feature width, at high occupancy (no accumulators: ~45 VGPRs), and leaves
// behind what the feature-width-proportional passes need as pure streams:
//   wt[slot][64]   : alpha*T of the block's 64 pixels for every K-step slot that blended anything, as 32
//                    (upper half, lower half) pairs: element 2p + h = pixel p of the 8x4 half h.  The forward
//                    reads one pair per lane and K-step (ONE 8-byte load: its two A operands); the backward
//                    reads rows, 32 consecutive floats per lane (its A operands: K-step t of half-wave k is
//                    pixel 16k + t/2 of half t%2);
//   gid[slot]      : Gaussian id of the slot (N for the zero slot that pads an odd count);
//   sidx[slot]     : sorted intersection index of the slot (-1 for the pad slot), and hit[sidx] = 1: the backward
//                    numbers the (tile, Gaussian) pairs that blended anything by a prefix sum over `hit` and merges
//                    the four blocks' partial gradient rows of such a pair into ONE row (raster_bwd_mfma.hip);
//   blk_rows[blk]  : number of slots of the block (even);
//   Tbuf / render_alphas / last_ids : per-pixel results of the chain.
// Slots of a block live in a fixed, sparse region of the slot space (no counting pre-pass):
//   region(tile, blk) = 4*offsets[tile] + 64*tile + blk * r16(L_tile), capacity r16(L_tile) = L_tile rounded up to 16
//   (gags_slot_base); the slots behind the block's count, up to the next multiple of 16, are zero slots.
// 8x8 rather than 8x4 blocks: a Gaussian then leaves ~36 % fewer (block, slot) rows, and those rows are the
// backward's HBM traffic; the price is ~30 % more zero weights inside the MFMA tiles.
#include "raster_mfma_common.h"

using namespace gags_mfma;

namespace {

__global__ __launch_bounds__(256) void pack_isects_kernel(int n_isects, const int32_t *__restrict__ flatten_ids,
                                                          const float *__restrict__ means2d,
                                                          const float *__restrict__ conics,
                                                          const float *__restrict__ opacities,
                                                          GRec *__restrict__ packed)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_isects) return;
    packed[s] = make_grec(means2d, conics, opacities, flatten_ids[s]);
}

// two-step packing: the record (incl. the log / sqrt of the extent) is built once per GAUSSIAN, then the
// per-intersection pass is a pure 32-byte gather (5x fewer transcendental evaluations at I/N ~ 5)
__global__ __launch_bounds__(256) void make_grec_kernel(int n, const float *__restrict__ means2d,
                                                        const float *__restrict__ conics,
                                                        const float *__restrict__ opacities,
                                                        const int32_t *__restrict__ radii, GRec *__restrict__ grec)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    if (radii && radii[g] <= 0) return;  // never referenced by an intersection
    grec[g] = make_grec(means2d, conics, opacities, g);
}

__global__ __launch_bounds__(256) void gather_grec_kernel(int n_isects, const int32_t *__restrict__ flatten_ids,
                                                          const GRec *__restrict__ grec, GRec *__restrict__ packed)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_isects) return;
    const float4 *src = reinterpret_cast<const float4 *>(grec + flatten_ids[s]);
    float4 *dst = reinterpret_cast<float4 *>(packed + s);
    const float4 u = src[0], v = src[1];
    dst[0] = u; dst[1] = v;
}

// COUNT (round 6, list trimming): the same walk with NO output but need[tile] = how many entries of the tile's sorted list
// any of its pixels reads before the tile is done (all pixels saturated, or the list exhausted).  Heavy views stop after a few
// hundred of a tile's tens of thousands of entries; run on the list cut to need[tile], every pass below does exactly what it
// does on the full list -- same hits, same pairs, same order -- while its scratch (1 KB per LIST ENTRY) shrinks by that factor.
// FD (round 6): the feature pass of a table of exactly FD = 16 channels -- the width the reference itself rasterizes
// (train.py:68) -- FUSED into this pass: with one pixel per lane the weight of a hit is in a register, the hit's feature row is
// the same for the whole wave (sixteen floats: one scalar load), and  acc[c] = fmaf(f[c], w, acc[c])  in list order is the
// oracle's own chain (a zero weight is an exact no-op): the render is bit-identical to raster_fwd_feat<1>'s, which streams
// the 1.1 GB of weight tiles a second time for sixteen channels' worth of arithmetic.  The tiles are still written (the
// backward reads them).
template <bool COUNT, int FD = 0>
__global__ __launch_bounds__(64, 8) void raster_weights_kernel(
    int width, int height, int tile_w, int n_tiles, int n_gauss, const GRec *__restrict__ packed,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int n_isects,
    float *__restrict__ wt, int32_t *__restrict__ gid_s, int32_t *__restrict__ sidx_s, int32_t *__restrict__ hit,
    int32_t *__restrict__ blk_rows, float *__restrict__ Tbuf, float *__restrict__ render_alphas,
    int32_t *__restrict__ last_ids, int by_gauss, int32_t *__restrict__ need,
    const float *__restrict__ colors = nullptr, const float *__restrict__ backgrounds = nullptr,
    float *__restrict__ render_colors = nullptr)
{
    static_assert(!(COUNT && FD > 0), "the count mode has no outputs");
    __shared__ __attribute__((aligned(16))) HRec ring[RING];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * GAGS_BLOCKS_PER_TILE);
    const int blk = logical & 3;
    const int tile = gags_tile_of_order(logical >> 2, tile_w, n_tiles / tile_w);
    const int lane = threadIdx.x;
    BlockGeom64 g;
    g.init(tile, blk, tile_w, width, height, lane);

    const int start = offsets[tile];
    const int end = offsets[tile + 1]  /* n_tiles + 1 entries: the last one is the intersection count */;
    const int sb = gags_slot_base(start, end, tile, blk);

    // One PIXEL per lane (round 6): lane e = 2 p + h owns pixel p of half h of the block -- element e of a weight row, so a
    // slot's 64 weights leave as one 256-byte store -- and the wave takes the block's hits ONE at a time: alpha of the hit at
    // the lane's pixel, then the transmittance step.  Until round 5 a lane owned a pixel PAIR and one of the two hits of a
    // step (the A-operand layout of the fused matrix-core forward this pass was cut out of): every lane then ran the
    // transmittance chain of both hits for both of its pixels -- each chain step twice per wave -- behind two half-wave
    // swaps: 127 VALU instructions per 128 (hit, pixel) pairs; now ~45 per 64.  Same arithmetic per (hit, pixel), same slot
    // order, same stop rule: weights, alphas and last_ids are bit-identical.
    const int e = lane, pp = e >> 1, hh = e & 1;
    const int pj = g.bx0 + (pp & 7), pi = g.by0 + (pp >> 3) + 4 * hh;
    const bool inside = (pi < height) && (pj < width);
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    float T = 1.0f;
    int cur = 0;
    bool done = !inside;
    float facc[FD > 0 ? FD : 1];
#pragma unroll
    for (int c = 0; c < (FD > 0 ? FD : 1); ++c) facc[c] = 0.f;

    HitStream hs;
    hs.by_gauss = by_gauss != 0;
    hs.init(ring, packed, flatten_ids, start, end, lane, g);

    int row = sb;
    int stop = start;  // COUNT: one past the last list entry this block consumed
    hs.refill(4);
    while (hs.rd < hs.nq && !__all(done)) {
        const HRec h = ring[hs.rd & (RING - 1)];  // (the same record for every lane: a broadcast read)
        hs.rd += 1;
        if ((hs.nq - hs.rd) < 4 && hs.pending) hs.refill(4);
        const float a = eval_alpha(h, px, py, true);
        // skip / stop / blend, exactly step_pair's order of operations
        const float t = T * (1.0f - a);
        const bool ok = !done && a > 0.f;
        const bool stp = ok && t <= GAGS_T_STOP;
        const bool bl = ok && !stp;
        const float w = bl ? a * T : 0.f;
        T = bl ? t : T;
        done = done || stp;
        cur = bl ? h.sidx : cur;
        if constexpr (COUNT) {
            stop = h.sidx + 1;
            continue;
        }
        // a hit that blends into none of the block's pixels leaves no slot: zero rows would be multiplied, stored, sorted and
        // summed like any other -- they were 20 % of all rows
        if (__ballot(w != 0.f) != 0ull) {
            if (FD == 0 || wt != nullptr) {  // (FD > 0 without tiles: a render nobody differentiates -- nothing but the image leaves)
                __builtin_nontemporal_store(w, &wt[(size_t)row * 64 + e]);
                if (e == 0) {
                    gid_s[row] = h.gid;
                    sidx_s[row] = h.sidx;
                    hit[h.sidx] = 1;  // up to four blocks store the same 1
                }
            }
            row += 1;
            if constexpr (FD > 0) {
                // (the hit's Gaussian is the same for every lane: its row arrives by scalar loads)
                const float *frow = colors + (size_t)__builtin_amdgcn_readfirstlane(h.gid) * FD;
#pragma unroll
                for (int c = 0; c < FD; ++c) facc[c] = __builtin_fmaf(frow[c], w, facc[c]);
            }
        }
    }
    if constexpr (COUNT) {
        if (lane == 0 && stop > start) atomicMax(&need[tile], stop - start);
        return;
    }
    const int used = row - sb;
    const int cnt = (used + 1) & ~1;  // consumers take slots in pairs: an odd count is padded with one zero slot that belongs to no Gaussian
    // ... and the region is filled with zero slots up to the next multiple of 16 (its capacity is one: gags_slot_base), so
    // that the 16-slot steps of the 16-bit matrix-core feature pass need neither a clamp nor a mask
    const bool tiles = FD == 0 || wt != nullptr;  // (wave-uniform)
    if (tiles) {
        for (int r = used; r < ((used + 15) & ~15); ++r) {
            wt[(size_t)(sb + r) * 64 + e] = 0.f;
            if (e == 0) { gid_s[sb + r] = n_gauss; sidx_s[sb + r] = -1; }
        }
        if (lane == 0) blk_rows[tile * GAGS_BLOCKS_PER_TILE + blk] = cnt;
    }
    if (inside) {
        const size_t pix = (size_t)pi * width + pj;
        if (tiles) Tbuf[pix] = T;
        render_alphas[pix] = 1.0f - T; last_ids[pix] = cur;
        if constexpr (FD > 0) {
            if (backgrounds != nullptr) {  // (wave-uniform)
#pragma unroll
                for (int c = 0; c < FD; ++c) facc[c] = __builtin_fmaf(T, backgrounds[c], facc[c]);
            }
            float4 *o = reinterpret_cast<float4 *>(render_colors + pix * FD);
#pragma unroll
            for (int c = 0; c < FD; c += 4) o[c >> 2] = make_float4(facc[c], facc[c + 1], facc[c + 2], facc[c + 3]);
        }
    }
}

}  // namespace

int gags_pack_isects_launch(int n, int n_isects, const int32_t *flat, const float *means2d, const float *conics,
                            const float *opacities, const int32_t *radii, void *grec, void *packed, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    if (n_isects <= 0) return GAGS_OK;
    if (grec) {
        hipLaunchKernelGGL(make_grec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, means2d, conics, opacities,
                           radii, reinterpret_cast<GRec *>(grec));
        if (packed)
            hipLaunchKernelGGL(gather_grec_kernel, dim3((n_isects + 255) / 256), dim3(256), 0, st, n_isects, flat,
                           reinterpret_cast<const GRec *>(grec), reinterpret_cast<GRec *>(packed));
    } else {
        hipLaunchKernelGGL(pack_isects_kernel, dim3((n_isects + 255) / 256), dim3(256), 0, st, n_isects, flat, means2d,
                           conics, opacities, reinterpret_cast<GRec *>(packed));
    }
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

int gags_raster_weights_launch(int width, int height, int n_gauss, const void *packed, int by_gauss, const int32_t *offsets,
                               const int32_t *flat, int n_isects, float *wt, int32_t *gid_s, int32_t *sidx_s,
                               int32_t *hit, int32_t *blk_rows, float *Tbuf, float *alphas, int32_t *last_ids,
                               hipStream_t st, const float *colors16, const float *backgrounds, float *render_colors)
{
    GAGS_CLEAR_ERR();
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h;
    // hit[i] = 1 for every intersection that blends into at least one pixel of its tile (+1 entry: an empty view)
    if (hit && hipMemsetAsync(hit, 0, sizeof(int32_t) * ((size_t)n_isects + 1), st) != hipSuccess) return GAGS_ELAUNCH;
    if (colors16)  // the 16-channel feature pass rides along (render_colors [H, W, 16] written here)
        hipLaunchKernelGGL((raster_weights_kernel<false, 16>), dim3(n_tiles * GAGS_BLOCKS_PER_TILE), dim3(64), 0, st, width, height, tile_w,
                           n_tiles, n_gauss, reinterpret_cast<const GRec *>(packed), offsets, flat, n_isects, wt, gid_s, sidx_s, hit,
                           blk_rows, Tbuf, alphas, last_ids, by_gauss, (int32_t *)nullptr, colors16, backgrounds, render_colors);
    else
        hipLaunchKernelGGL(raster_weights_kernel<false>, dim3(n_tiles * GAGS_BLOCKS_PER_TILE), dim3(64), 0, st, width, height, tile_w, n_tiles,
                           n_gauss, reinterpret_cast<const GRec *>(packed), offsets, flat, n_isects, wt, gid_s, sidx_s, hit,
                           blk_rows, Tbuf, alphas, last_ids, by_gauss, (int32_t *)nullptr, (const float *)nullptr, (const float *)nullptr,
                           (float *)nullptr);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

// ---- list trimming (round 6): need[tile], the cut lists, and the translation of last_ids back to the full lists -------------
namespace {
// one workgroup per tile: flat_out[off_new[t] + j] = flat_in[off_old[t] + j] for j < need[t]
__global__ __launch_bounds__(256) void trim_gather_kernel(int n_tiles, const int32_t *__restrict__ off_old,
                                                          const int32_t *__restrict__ off_new,
                                                          const int32_t *__restrict__ flat_in, int32_t *__restrict__ flat_out)
{
    const int t = blockIdx.x;
    const int a = off_new[t], cnt = off_new[t + 1] - a, b = off_old[t];
    for (int j = threadIdx.x; j < cnt; j += 256) flat_out[a + j] = flat_in[b + j];
}

// off_new[0 .. n_tiles] = exclusive prefix sums of need (the inclusive sums `cum` shifted by one; entry n_tiles = the total)
__global__ __launch_bounds__(256) void trim_offsets_kernel(int n_tiles, const int32_t *__restrict__ cum, int32_t *__restrict__ off_new)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t <= n_tiles) off_new[t] = t == 0 ? 0 : cum[t - 1];
}

// last_ids are sorted indices: those of the cut lists become those of the full lists again (a pixel that blended nothing --
// alpha exactly 0 -- keeps its 0)
__global__ __launch_bounds__(256) void trim_last_ids_kernel(int width, int height, int tile_w, const int32_t *__restrict__ off_old,
                                                            const int32_t *__restrict__ off_new,
                                                            const float *__restrict__ alphas, int32_t *__restrict__ last_ids)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (int64_t)width * height) return;
    const int i = (int)(p / width), j = (int)(p - (int64_t)i * width);
    const int t = (i / GAGS_TILE) * tile_w + j / GAGS_TILE;
    if (alphas[p] > 0.f) last_ids[p] += off_old[t] - off_new[t];
}
}  // namespace

int gags_list_need_launch(int width, int height, int n_gauss, const void *packed, int by_gauss, const int32_t *offsets,
                          const int32_t *flat, int n_isects, int32_t *need, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h;
    if (hipMemsetAsync(need, 0, sizeof(int32_t) * (size_t)n_tiles, st) != hipSuccess) return GAGS_ELAUNCH;
    hipLaunchKernelGGL(raster_weights_kernel<true>, dim3(n_tiles * GAGS_BLOCKS_PER_TILE), dim3(64), 0, st, width, height, tile_w, n_tiles,
                       n_gauss, reinterpret_cast<const GRec *>(packed), offsets, flat, n_isects, (float *)nullptr, (int32_t *)nullptr,
                       (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (float *)nullptr, (float *)nullptr,
                       (int32_t *)nullptr, by_gauss, need, (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

int gags_trim_offsets_launch(int n_tiles, const int32_t *cum, int32_t *off_new, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    hipLaunchKernelGGL(trim_offsets_kernel, dim3((n_tiles + 1 + 255) / 256), dim3(256), 0, st, n_tiles, cum, off_new);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

int gags_trim_gather_launch(int n_tiles, const int32_t *off_old, const int32_t *off_new, const int32_t *flat_in, int32_t *flat_out,
                            hipStream_t st)
{
    GAGS_CLEAR_ERR();
    hipLaunchKernelGGL(trim_gather_kernel, dim3(n_tiles), dim3(256), 0, st, n_tiles, off_old, off_new, flat_in, flat_out);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

int gags_trim_last_ids_launch(int width, int height, const int32_t *off_old, const int32_t *off_new, const float *alphas,
                              int32_t *last_ids, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE;
    const int64_t pix = (int64_t)width * height;
    hipLaunchKernelGGL(trim_last_ids_kernel, dim3((unsigned)((pix + 255) / 256)), dim3(256), 0, st, width, height, tile_w, off_old,
                       off_new, alphas, last_ids);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
