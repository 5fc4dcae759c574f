// K8b + the WEIGHTS pass of the matrix-core rasterizer.
//
// raster_weights: one wave per (tile, 8x4 pixel block).  It does ALL the per-(pixel, Gaussian)
// scalar work of the view exactly once -- alpha, skip rule, transmittance chain, stop rule --
// independent of the feature width, at high occupancy (no accumulators: ~45 VGPRs), and leaves
// behind what the feature-width-proportional passes need as pure streams:
//   wt[slot][32]   : alpha*T of the block's pixels for every K-step slot that blended anything,
//                    stored as the MFMA A-operand image of the BACKWARD (row = slot, columns in
//                    [k][s] pixel order); the forward reads the same rows column-wise;
//   gid[slot]      : Gaussian id of the slot (N for the unused partner of a lone last hit);
//   blk_rows[blk]  : number of slots of the block (even);
//   Tbuf / render_alphas / last_ids : per-pixel results of the chain.
// Slots of a block live in a fixed, sparse region of the slot space (no counting pre-pass):
//   region(tile, blk) = 8*(offsets[tile] + tile) + blk * even(L_tile), capacity even(L_tile).
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

__global__ __launch_bounds__(64, 4) void raster_weights_kernel(
    int width, int height, int tile_w, int n_tiles, int n_gauss, const GRec *__restrict__ packed,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int n_isects,
    float *__restrict__ wt, int32_t *__restrict__ gid_s, int32_t *__restrict__ blk_rows, float *__restrict__ Tbuf,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids)
{
    __shared__ __attribute__((aligned(16))) HRec ring[RING];

    const int logical = gags_xcd_remap(blockIdx.x, n_tiles * 8);
    const int blk = logical & 7;
    const int tile = gags_tile_of_order(logical >> 3, tile_w, n_tiles / tile_w);
    const int lane = threadIdx.x;
    BlockGeom g;
    g.init(tile, blk, tile_w, width, height, lane);
    const int p = g.p, k = g.k;

    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];
    const int sb = gags_slot_base(start, end, tile, blk);

    PixState st;
    st.T = 1.0f; st.cur = 0; st.done = !g.inside;

    HitStream hs;
    hs.init(ring, packed, flatten_ids, start, end, lane, g);

    int row = sb;
    const int wpos = (p & 1) * 16 + (p >> 1);  // pixel p inside a slot row, [k][s] order (p = 2s + k)
    hs.refill(6);
    if (!__all(st.done) && hs.rd < hs.nq) {
        bool v_n;
        HRec h_n = hs.at(hs.rd, k, v_n);
        float a_n = eval_alpha(h_n, g.px, g.py, v_n);
        int gid_n = v_n ? h_n.gid : n_gauss, sidx_n = h_n.sidx;
        auto kstep = [&]() -> bool {
            const float a_c = a_n;
            const int gid_c = gid_n, sidx_c = sidx_n;
            hs.rd += 2;
            if ((hs.nq - hs.rd) < 6 && hs.pending) hs.refill(6);
            const bool more = hs.rd < hs.nq;
            h_n = hs.at(hs.rd, k, v_n);
            a_n = eval_alpha(h_n, g.px, g.py, v_n);  // one step ahead of the chain below
            gid_n = v_n ? h_n.gid : n_gauss;
            sidx_n = h_n.sidx;
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_c), __float_as_uint(a_c), false, false);
            bool blended;
            const float wgt = step_pair(st, __uint_as_float(sw[0]), __uint_as_float(sw[1]), k, blended);
            st.cur = blended ? sidx_c : st.cur;
            if (__any(wgt != 0.f)) {  // steps nobody blends leave no slot
                wt[(size_t)(row + k) * 32 + wpos] = wgt;  // 2 x 128 B per step
                if (p == 0) gid_s[row + k] = gid_c;
                row += 2;
            }
            return more && !__all(st.done);
        };
        while (kstep()) {}
    }
    if (lane == 0) blk_rows[tile * 8 + blk] = row - sb;
    {
        const auto cs = __builtin_amdgcn_permlane32_swap((unsigned)st.cur, (unsigned)st.cur, false, false);
        st.cur = max((int)cs[0], (int)cs[1]);  // sorted indices grow along the list
    }
    if (k == 0 && g.inside) {
        const size_t pix = (size_t)g.pi * width + g.pj;
        Tbuf[pix] = st.T;
        render_alphas[pix] = 1.0f - st.T;
        last_ids[pix] = st.cur;
    }
}

}  // namespace

int gags_pack_isects_launch(int n_isects, const int32_t *flat, const float *means2d, const float *conics,
                            const float *opacities, void *packed, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    if (n_isects <= 0) return GAGS_OK;
    hipLaunchKernelGGL(pack_isects_kernel, dim3((n_isects + 255) / 256), dim3(256), 0, st, n_isects, flat, means2d,
                       conics, opacities, reinterpret_cast<GRec *>(packed));
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

int gags_raster_weights_launch(int width, int height, int n_gauss, const void *packed, const int32_t *offsets,
                               const int32_t *flat, int n_isects, float *wt, int32_t *gid_s, int32_t *blk_rows,
                               float *Tbuf, float *alphas, int32_t *last_ids, hipStream_t st)
{
    GAGS_CLEAR_ERR();
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int n_tiles = tile_w * tile_h;
    hipLaunchKernelGGL(raster_weights_kernel, dim3(n_tiles * 8), dim3(64), 0, st, width, height, tile_w, n_tiles,
                       n_gauss, reinterpret_cast<const GRec *>(packed), offsets, flat, n_isects, wt, gid_s, blk_rows,
                       Tbuf, alphas, last_ids);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
