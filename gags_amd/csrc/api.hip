// C-ABI entry points of the raster stages: argument validation + dispatch onto the VALU kernels
// (narrow / odd D, full geometry gradients) and the matrix-core kernels (D >= 16, D % 4 == 0;
// the single-kernel fallbacks: D % 32 == 0).
#include <cstdlib>
#include "common.h"
#include "scan.h"

// raster_valu.hip
int gags_raster_fwd_valu(int d, int width, int height, const float *means2d, const float *conics,
                         const float *opacities, const float *colors, const float *backgrounds,
                         const int32_t *offsets, const int32_t *flat, int n_isects, float *out, float *alphas,
                         int32_t *last_ids, hipStream_t st);
int gags_raster_bwd_valu(int d, int width, int height, const float *means2d, const float *conics,
                         const float *opacities, const float *colors, const float *backgrounds,
                         const int32_t *offsets, const int32_t *flat, int n_isects, const float *alphas,
                         const int32_t *last_ids, const float *v_out, const float *v_alpha, float *v_colors,
                         float *v_opac, float *v_m2d, float *v_con, bool geom, hipStream_t st);
// raster_weights.hip
int gags_pack_isects_launch(int n, int n_isects, const int32_t *flat, const float *means2d, const float *conics,
                            const float *opacities, const int32_t *radii, void *grec, void *packed, hipStream_t st);
int gags_raster_weights_launch(int width, int height, int n_gauss, const void *packed, int by_gauss, const int32_t *offsets,
                               const int32_t *flat, int n_isects, float *wt, int32_t *gid_s, int32_t *sidx_s,
                               int32_t *hit, int32_t *blk_rows, float *Tbuf, float *alphas, int32_t *last_ids,
                               hipStream_t st, const float *colors16 = nullptr, const float *backgrounds = nullptr,
                               float *render_colors = nullptr);
int gags_list_need_launch(int width, int height, int n_gauss, const void *packed, int by_gauss, const int32_t *offsets,
                          const int32_t *flat, int n_isects, int32_t *need, hipStream_t st);
int gags_trim_offsets_launch(int n_tiles, const int32_t *cum, int32_t *off_new, hipStream_t st);
int gags_trim_gather_launch(int n_tiles, const int32_t *off_old, const int32_t *off_new, const int32_t *flat_in, int32_t *flat_out,
                            hipStream_t st);
int gags_trim_last_ids_launch(int width, int height, const int32_t *off_old, const int32_t *off_new, const float *alphas,
                              int32_t *last_ids, hipStream_t st);
// raster_fwd_mfma.hip
int gags_raster_fwd_feat_launch(int d, int width, int height, int n_gauss, const float *colors, int colors_f16, int exact,
                                const float *backgrounds, const int32_t *offsets, int n_isects,
                                const int32_t *blk_rows, const float *wt, const int32_t *gid_s, const float *Tbuf,
                                float *out, hipStream_t st);
int gags_raster_fwd_fused_launch(int d, int width, int height, const void *packed, const float *colors,
                                 const float *backgrounds, const int32_t *offsets, const int32_t *flat, int n_isects,
                                 float *out, float *alphas, int32_t *last_ids, int by_gauss, hipStream_t st);
// raster_bwd_mfma.hip
int64_t gags_bwd_staged_scratch_bytes_impl(int64_t rows, int n_gauss, int d);
int gags_raster_bwd_staged_launch(int d, int width, int height, int n_gauss, const int32_t *offsets, int n_isects,
                                  const float *v_out, const int32_t *blk_rows, const int32_t *trow, int64_t rows,
                                  const float *wt, const int32_t *gid_s, const int32_t *trow_s, void *scratch,
                                  int64_t scratch_bytes, float *v_colors, int stage, int ch_begin, int ch_count,
                                  const int32_t *rows_dev, const int32_t *wire_pos, float *wire, const uint8_t *keep_prev,
                                  uint8_t *keep_cur, hipStream_t st);
int64_t gags_raster_bwd_geom_scratch_bytes_impl(int64_t n_isects, int width, int height, int n_gauss, int d, int64_t n_rows);
int gags_raster_bwd_geom_launch(int d, int n_gauss, int width, int height, const float *colors, const float *backgrounds,
                                const int32_t *offsets, int n_isects, const void *packed, const float *v_out,
                                const float *v_alphas, const int32_t *blk_rows, const float *wt, const int32_t *gid_s,
                                const int32_t *sidx_s, const float *Tbuf, void *scratch, int64_t scratch_bytes, float *v_geo,
                                int by_gauss, const int32_t *row_base, int64_t n_rows, const int32_t *hit,
                                const int32_t *flatten_ids, int f32mfma, hipStream_t st);
int gags_blended_mask_launch(int n_isects, const int32_t *hit, const int32_t *flatten_ids, unsigned char *mask, hipStream_t st);
int gags_bwd_slot_rows_launch(int width, int height, int n_isects, const int32_t *offsets, const int32_t *blk_rows,
                              const int32_t *sidx_s, const int32_t *trow, int32_t *trow_s, hipStream_t st);
int gags_raster_bwd_atomic_launch(int d, int width, int height, const void *packed, const int32_t *offsets,
                                  const int32_t *flat, int n_isects, const float *v_out, float *v_colors,
                                  int by_gauss, hipStream_t st);

namespace {
inline int64_t al256(int64_t x) { return (x + 255) / 256 * 256; }
struct FwdScratch {
    int64_t wt, gid, sidx, hit, tbuf, total;
};
// slot space: gags_slot_count() slots of 64 weights (common.h)
inline FwdScratch fwd_layout(int64_t n_isects, int width, int height)
{
    const int64_t tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    const int64_t slots = gags_slot_count(n_isects, tile_w * tile_h);
    FwdScratch L;
    int64_t o = 0;
    L.wt = o; o += al256(slots * 256);
    L.gid = o; o += al256(slots * 4);
    L.sidx = o; o += al256(slots * 4);
    L.hit = o; o += al256((n_isects + 1) * 4);
    L.tbuf = o; o += al256((int64_t)width * height * 4);
    L.total = o;
    return L;
}
inline bool fused_width(int d) { return d >= 32 && d % 32 == 0; }  // single-kernel forward / atomic backward
// Intersections of one view the raster kernels can index: the count itself below GAGS_MAX_ISECTS AND the slot space it
// spans (4 I + 64 tiles + 64 slots, common.h) below 2^30 -- slot numbers are multiplied by 4 in unsigned 32-bit offsets
// (the id gathers of the feature pass); near the cap the per-tile slack alone (0.5 M slots at 1080p) would wrap them.
inline bool isects_ok(int64_t n_isects, int width, int height)
{
    if (n_isects < 0 || n_isects >= GAGS_MAX_ISECTS || width <= 0 || height <= 0) return false;
    const int64_t tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    return gags_slot_count(n_isects, tile_w * tile_h) < (1ll << 30);
}
}  // namespace

extern "C" int gags_pack_isects(int n, int64_t n_isects, const int32_t *flatten_ids, const float *means2d,
                                const float *conics, const float *opacities, const int32_t *radii, void *grec,
                                void *packed, void *stream)
{
    if (n < 0 || n_isects < 0 || n_isects >= (1ll << 31)) return GAGS_EINVAL;
    if (n_isects == 0) return GAGS_OK;
    if (!flatten_ids || !means2d || !conics || !opacities || (!packed && !grec)) return GAGS_EINVAL;
    return gags_pack_isects_launch(n, (int)n_isects, flatten_ids, means2d, conics, opacities, radii, grec, packed,
                                   (hipStream_t)stream);
}

extern "C" int64_t gags_raster_fwd_scratch_bytes(int64_t n_isects, int width, int height)
{
    if (n_isects < 0 || width <= 0 || height <= 0) return 0;
    return fwd_layout(n_isects, width, height).total;
}

namespace {
// GAGS_FWD_D16_UNFUSED=1 (experiments; read once): the 16-channel feature pass as its own kernel
bool fwd_d16_unfused()
{
    static const bool v = [] { const char *e = getenv("GAGS_FWD_D16_UNFUSED"); return e && e[0] == '1'; }();
    return v;
}
}  // namespace

extern "C" int gags_raster_fwd(int d, int n, int width, int height, const float *means2d, const float *conics,
                               const float *opacities, const float *colors, const float *backgrounds,
                               const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                               const void *packed, float *render_colors, float *render_alphas, int32_t *last_ids,
                               void *scratch, int64_t scratch_bytes, int32_t *blk_rows, int flags, void *stream)
{
    if (d <= 0 || n < 0 || width <= 0 || height <= 0 || !isects_ok(n_isects, width, height)) return GAGS_EINVAL;
    if (!isect_offsets || !render_colors || !render_alphas || !last_ids) return GAGS_EINVAL;
    if (n_isects > 0 && (!means2d || !conics || !opacities || !colors || !flatten_ids)) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool split = scratch && blk_rows && gags_mfma_width(d);
    if ((flags & GAGS_FEAT_F16) && (!split || (flags & GAGS_FWD_NO_MFMA) || !packed || n <= 0))
        return GAGS_EINVAL;  // an fp16 feature table is only read by the feature pass of the split forward
    if (d == 16 && !scratch && !blk_rows && packed && n > 0 && n_isects > 0 && !(flags & (GAGS_FWD_NO_MFMA | GAGS_FEAT_F16)) &&
        !(reinterpret_cast<uintptr_t>(render_colors) & 15) && !fwd_d16_unfused()) {
        // a 16-channel render without scratch (nobody will differentiate it): the fused weights + feature pass alone, no tiles
        return gags_raster_weights_launch(width, height, n, packed, (flags & GAGS_RECS_BY_GAUSSIAN) ? 1 : 0, isect_offsets, flatten_ids,
                                          (int)n_isects, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, render_alphas, last_ids, st,
                                          colors, backgrounds, render_colors);
    }
    if (!(flags & GAGS_FWD_NO_MFMA) && (split || fused_width(d)) && (packed || n_isects == 0) && n > 0) {
        if (split) {  // split forward: weights once, then the feature stream
            const FwdScratch L = fwd_layout(n_isects, width, height);
            if (scratch_bytes < L.total) return GAGS_ESCRATCH;
            char *sb = (char *)scratch;
            float *wt = (float *)(sb + L.wt);
            int32_t *gid_s = (int32_t *)(sb + L.gid);
            float *tbuf = (float *)(sb + L.tbuf);
            int rc = GAGS_OK;
            // an fp32 table of exactly 16 channels (the reference's own width, train.py:68): the feature pass rides along with
            // the weights pass (bit-identical to the feature kernel: raster_weights.hip); the two ONLY_* flags keep the passes apart
            const bool fuse16 = d == 16 && !(flags & (GAGS_FEAT_F16 | GAGS_FWD_ONLY_FEATURES | GAGS_FWD_ONLY_WEIGHTS)) && n_isects > 0 &&
                                !(reinterpret_cast<uintptr_t>(render_colors) & 15) && !fwd_d16_unfused();
            if (!(flags & GAGS_FWD_ONLY_FEATURES))
                rc = gags_raster_weights_launch(width, height, n, packed, (flags & GAGS_RECS_BY_GAUSSIAN) ? 1 : 0, isect_offsets, flatten_ids, (int)n_isects, wt,
                                                gid_s, (int32_t *)(sb + L.sidx), (int32_t *)(sb + L.hit), blk_rows, tbuf,
                                                render_alphas, last_ids, st, fuse16 ? colors : nullptr, backgrounds, render_colors);
            if (rc != GAGS_OK || (flags & GAGS_FWD_ONLY_WEIGHTS) || fuse16) return rc;
            return gags_raster_fwd_feat_launch(d, width, height, n, colors, (flags & GAGS_FEAT_F16) ? ((flags & GAGS_FWD_F16MFMA) ? 2 : 1) : 0,
                                               (flags & GAGS_FWD_EXACT) ? 1 : 0, backgrounds, isect_offsets, (int)n_isects,
                                               blk_rows, wt, gid_s, tbuf, render_colors, st);
        }
        return gags_raster_fwd_fused_launch(d, width, height, packed, colors, backgrounds, isect_offsets, flatten_ids,
                                            (int)n_isects, render_colors, render_alphas, last_ids,
                                            (flags & GAGS_RECS_BY_GAUSSIAN) ? 1 : 0, st);
    }
    return gags_raster_fwd_valu(d, width, height, means2d, conics, opacities, colors, backgrounds, isect_offsets,
                                flatten_ids, (int)n_isects, render_colors, render_alphas, last_ids, st);
}

// ---- list trimming (round 6; include/gags_raster.h) ---------------------------------------------------------------------
extern "C" int gags_raster_list_need(int n, int width, int height, const int32_t *isect_offsets, const int32_t *flatten_ids,
                                     int64_t n_isects, const void *packed, int flags, int32_t *need, void *stream)
{
    if (n < 0 || width <= 0 || height <= 0 || n_isects < 0 || n_isects >= (1ll << 31) || !need || !isect_offsets) return GAGS_EINVAL;
    if (n_isects > 0 && (!flatten_ids || !packed || n == 0)) return GAGS_EINVAL;
    return gags_list_need_launch(width, height, n, packed, (flags & GAGS_RECS_BY_GAUSSIAN) ? 1 : 0, isect_offsets, flatten_ids,
                                 (int)n_isects, need, (hipStream_t)stream);
}

extern "C" int gags_trim_lists(int width, int height, const int32_t *isect_offsets, const int32_t *need_cum,
                               const int32_t *flatten_ids, int32_t *offsets_out, int32_t *flatten_out, void *stream)
{
    if (width <= 0 || height <= 0 || !isect_offsets || !need_cum || !offsets_out) return GAGS_EINVAL;
    const int tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    int rc = gags_trim_offsets_launch(tile_w * tile_h, need_cum, offsets_out, (hipStream_t)stream);
    if (rc != GAGS_OK || !flatten_out) return rc;  // (flatten_out NULL: the offsets only)
    if (!flatten_ids) return GAGS_EINVAL;
    return gags_trim_gather_launch(tile_w * tile_h, isect_offsets, offsets_out, flatten_ids, flatten_out, (hipStream_t)stream);
}

extern "C" int gags_trim_last_ids(int width, int height, const int32_t *isect_offsets, const int32_t *offsets_trimmed,
                                  const float *render_alphas, int32_t *last_ids, void *stream)
{
    if (width <= 0 || height <= 0 || !isect_offsets || !offsets_trimmed || !render_alphas || !last_ids) return GAGS_EINVAL;
    return gags_trim_last_ids_launch(width, height, isect_offsets, offsets_trimmed, render_alphas, last_ids, (hipStream_t)stream);
}

extern "C" int gags_raster_bwd(int d, int width, int height, const float *means2d, const float *conics,
                               const float *opacities, const float *colors, const float *backgrounds,
                               const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                               const void *packed, const float *render_alphas, const int32_t *last_ids,
                               const float *v_render_colors, const float *v_render_alphas, float *v_colors,
                               float *v_opacities, float *v_means2d, float *v_conics, int flags, void *stream)
{
    if (d <= 0 || width <= 0 || height <= 0 || !isects_ok(n_isects, width, height)) return GAGS_EINVAL;
    if (n_isects == 0) return GAGS_OK;
    if (!means2d || !conics || !opacities || !colors || !isect_offsets || !flatten_ids || !render_alphas ||
        !last_ids || !v_render_colors || !v_colors)
        return GAGS_EINVAL;
    const bool geom = !(flags & GAGS_BWD_COLORS_ONLY);
    if (geom && (!v_opacities || !v_means2d || !v_conics)) return GAGS_EINVAL;
    if (!geom && !(flags & GAGS_FWD_NO_MFMA) && packed) {
        const int rc = gags_raster_bwd_atomic_launch(d, width, height, packed, isect_offsets, flatten_ids,
                                                     (int)n_isects, v_render_colors, v_colors,
                                                     (flags & GAGS_RECS_BY_GAUSSIAN) ? 1 : 0, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    return gags_raster_bwd_valu(d, width, height, means2d, conics, opacities, colors, backgrounds, isect_offsets,
                                flatten_ids, (int)n_isects, render_alphas, last_ids, v_render_colors,
                                v_render_alphas, v_colors, v_opacities, v_means2d, v_conics, geom,
                                (hipStream_t)stream);
}

extern "C" int64_t gags_bwd_staged_scratch_bytes(int64_t rows, int n, int d)
{
    if (rows < 0 || n < 0 || d <= 0) return 0;
    return gags_bwd_staged_scratch_bytes_impl(rows, n, d);
}

namespace {
// rowmap = trow[n_isects + 1] (exclusive prefix sum of the forward's hit flags) followed, 256-B aligned, by
// trow_s[slots] (tile row of every K-step slot of the forward's slot space)
inline int64_t rowmap_slot_off(int64_t n_isects) { return al256((n_isects + 1) * 4) / 4; }
inline int64_t slot_count(int64_t n_isects, int width, int height)
{
    const int64_t tile_w = (width + GAGS_TILE - 1) / GAGS_TILE, tile_h = (height + GAGS_TILE - 1) / GAGS_TILE;
    return gags_slot_count(n_isects, tile_w * tile_h);
}
}  // namespace

extern "C" int64_t gags_bwd_rowmap_elems(int64_t n_isects, int width, int height)
{
    if (n_isects < 0 || width <= 0 || height <= 0) return 0;
    return rowmap_slot_off(n_isects) + slot_count(n_isects, width, height);
}

extern "C" int64_t gags_bwd_rowmap_scratch_bytes(int64_t n_isects)
{
    return n_isects < 0 ? 0 : gags_scan::scratch_bytes(n_isects > 0 ? n_isects : 1);
}

extern "C" int gags_bwd_rowmap(int64_t n_isects, int width, int height, const int32_t *isect_offsets,
                               const int32_t *blk_rows, const void *fwd_scratch, int64_t fwd_scratch_bytes,
                               int32_t *rowmap, int64_t rowmap_elems, int32_t *total, void *scratch,
                               int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!isects_ok(n_isects, width, height) || !fwd_scratch || !rowmap || !total ||
        !isect_offsets || !blk_rows)
        return GAGS_EINVAL;
    if (rowmap_elems < gags_bwd_rowmap_elems(n_isects, width, height)) return GAGS_ESCRATCH;
    const FwdScratch L = fwd_layout(n_isects, width, height);
    if (fwd_scratch_bytes < L.total) return GAGS_ESCRATCH;
    hipStream_t st = (hipStream_t)stream;
    int32_t *trow = rowmap;
    if (hipMemsetAsync(trow, 0, sizeof(int32_t), st) != hipSuccess) return GAGS_ELAUNCH;
    if (n_isects == 0) return hipMemsetAsync(total, 0, sizeof(int32_t), st) == hipSuccess ? GAGS_OK : GAGS_ELAUNCH;
    if (!scratch || scratch_bytes < gags_scan::scratch_bytes(n_isects)) return GAGS_ESCRATCH;
    const char *fs = (const char *)fwd_scratch;
    gags_scan::launch<false>((int)n_isects, (const int32_t *)(fs + L.hit), trow + 1, total, (int32_t *)scratch, st);
    GAGS_CHECK_LAUNCH();
    return gags_bwd_slot_rows_launch(width, height, (int)n_isects, isect_offsets, blk_rows,
                                     (const int32_t *)(fs + L.sidx), trow, rowmap + rowmap_slot_off(n_isects), st);
}

extern "C" int64_t gags_raster_bwd_geom_scratch_bytes(int64_t n_isects, int width, int height, int n, int d, int64_t n_rows)
{
    if (n_isects < 0 || width <= 0 || height <= 0 || n < 0 || d <= 0) return 0;
    return gags_raster_bwd_geom_scratch_bytes_impl(n_isects, width, height, n, d, n_rows);
}

extern "C" int gags_raster_bwd_geom(int d, int n, int width, int height, const float *colors, const float *backgrounds,
                                    const int32_t *isect_offsets, int64_t n_isects, const void *packed,
                                    const float *v_render_colors, const float *v_render_alphas, const int32_t *blk_rows,
                                    const void *fwd_scratch, int64_t fwd_scratch_bytes, void *scratch, int64_t scratch_bytes,
                                    float *v_geo, const int32_t *flatten_ids, const int32_t *row_base, int64_t n_rows,
                                    int flags, void *stream)
{
    if (d <= 0 || width <= 0 || height <= 0 || n < 0 || !isects_ok(n_isects, width, height)) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!colors || !isect_offsets || !packed || !v_render_colors || !blk_rows || !fwd_scratch || !scratch || !v_geo)
        return GAGS_EINVAL;
    const FwdScratch L = fwd_layout(n_isects, width, height);
    if (fwd_scratch_bytes < L.total) return GAGS_ESCRATCH;
    const char *fs = (const char *)fwd_scratch;
    return gags_raster_bwd_geom_launch(d, n, width, height, colors, backgrounds, isect_offsets, (int)n_isects, packed,
                                       v_render_colors, v_render_alphas, blk_rows, (const float *)(fs + L.wt),
                                       (const int32_t *)(fs + L.gid), (const int32_t *)(fs + L.sidx),
                                       (const float *)(fs + L.tbuf), scratch, scratch_bytes, v_geo,
                                       (flags & GAGS_RECS_BY_GAUSSIAN) ? 1 : 0, row_base, n_rows,
                                       (const int32_t *)(fs + L.hit), flatten_ids, (flags & 32) ? 1 : 0,
                                       (hipStream_t)stream);
}

extern "C" int gags_blended_mask(int64_t n_isects, int width, int height, int n, const int32_t *flatten_ids,
                                 const void *fwd_scratch, int64_t fwd_scratch_bytes, unsigned char *mask, void *stream)
{
    if (!isects_ok(n_isects, width, height) || n < 0) return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!mask) return GAGS_EINVAL;
    if (hipMemsetAsync(mask, 0, (size_t)n, (hipStream_t)stream) != hipSuccess) return GAGS_ELAUNCH;
    if (n_isects == 0) return GAGS_OK;
    if (!flatten_ids || !fwd_scratch) return GAGS_EINVAL;
    const FwdScratch L = fwd_layout(n_isects, width, height);
    if (fwd_scratch_bytes < L.total) return GAGS_ESCRATCH;
    return gags_blended_mask_launch((int)n_isects, (const int32_t *)((const char *)fwd_scratch + L.hit), flatten_ids, mask,
                                    (hipStream_t)stream);
}

namespace {
int staged_entry(int d, int n, int width, int height, const int32_t *isect_offsets, int64_t n_isects,
                 const float *v_render_colors, const int32_t *blk_rows, const int32_t *rowmap, int64_t rows,
                 const void *fwd_scratch, int64_t fwd_scratch_bytes, void *scratch, int64_t scratch_bytes, float *v_colors,
                 int stage, int ch_begin, int ch_count, const int32_t *rows_dev, const int32_t *wire_pos, float *wire,
                 const uint8_t *keep_prev, uint8_t *keep_cur, void *stream)
{
    if (d <= 0 || width <= 0 || height <= 0 || n < 0 || !isects_ok(n_isects, width, height) || rows < 0 ||
        rows >= (1ll << 31) || stage < 0 || (stage & 15) > 3)
        return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!isect_offsets || !blk_rows || !rowmap || !fwd_scratch || !scratch || !v_colors || !v_render_colors)
        return GAGS_EINVAL;
    // (the rows kernel reads a tile's four slot counts as ONE 16-byte scalar load)
    if (reinterpret_cast<uintptr_t>(blk_rows) & 15) return GAGS_EINVAL;
    const FwdScratch L = fwd_layout(n_isects, width, height);
    if (fwd_scratch_bytes < L.total) return GAGS_ESCRATCH;
    const char *fs = (const char *)fwd_scratch;
    return gags_raster_bwd_staged_launch(d, width, height, n, isect_offsets, (int)n_isects, v_render_colors, blk_rows,
                                         rowmap, rows, (const float *)(fs + L.wt), (const int32_t *)(fs + L.gid),
                                         rowmap + rowmap_slot_off(n_isects), scratch, scratch_bytes, v_colors, stage,
                                         ch_begin, ch_count, rows_dev, wire_pos, wire, keep_prev, keep_cur, (hipStream_t)stream);
}
}  // namespace

extern "C" int gags_raster_bwd_colors_staged_cap(int d, int n, int width, int height, const int32_t *isect_offsets,
                                                 int64_t n_isects, const float *v_render_colors, const int32_t *blk_rows,
                                                 const int32_t *rowmap, int64_t rows, const void *fwd_scratch,
                                                 int64_t fwd_scratch_bytes, void *scratch, int64_t scratch_bytes,
                                                 float *v_colors, int stage, int ch_begin, int ch_count,
                                                 const int32_t *rows_dev, void *stream)
{
    return staged_entry(d, n, width, height, isect_offsets, n_isects, v_render_colors, blk_rows, rowmap, rows, fwd_scratch,
                        fwd_scratch_bytes, scratch, scratch_bytes, v_colors, stage, ch_begin, ch_count, rows_dev, nullptr, nullptr,
                        nullptr, nullptr, stream);
}

extern "C" int gags_raster_bwd_colors_staged_wire(int d, int n, int width, int height, const int32_t *isect_offsets,
                                                  int64_t n_isects, const float *v_render_colors, const int32_t *blk_rows,
                                                  const int32_t *rowmap, int64_t rows, const void *fwd_scratch,
                                                  int64_t fwd_scratch_bytes, void *scratch, int64_t scratch_bytes,
                                                  float *v_colors, int stage, int ch_begin, int ch_count,
                                                  const int32_t *wire_pos, float *wire, const uint8_t *keep_prev,
                                                  uint8_t *keep_cur, void *stream)
{
    if ((wire != nullptr) != (wire_pos != nullptr) || (wire && (stage & 64))) return GAGS_EINVAL;  // (the block is fp32)
    if ((keep_prev != nullptr) != (keep_cur != nullptr) || (keep_cur && (keep_prev == keep_cur || (stage & 128)))) return GAGS_EINVAL;
    return staged_entry(d, n, width, height, isect_offsets, n_isects, v_render_colors, blk_rows, rowmap, rows, fwd_scratch,
                        fwd_scratch_bytes, scratch, scratch_bytes, v_colors, stage, ch_begin, ch_count, nullptr, wire_pos, wire,
                        keep_prev, keep_cur, stream);
}

extern "C" int gags_raster_bwd_colors_staged_keep(int d, int n, int width, int height, const int32_t *isect_offsets,
                                                  int64_t n_isects, const float *v_render_colors, const int32_t *blk_rows,
                                                  const int32_t *rowmap, int64_t rows, const void *fwd_scratch,
                                                  int64_t fwd_scratch_bytes, void *scratch, int64_t scratch_bytes,
                                                  float *v_colors, int stage, int ch_begin, int ch_count,
                                                  const uint8_t *keep_prev, uint8_t *keep_cur, void *stream)
{
    if (!keep_prev || !keep_cur || keep_prev == keep_cur || (stage & 128)) return GAGS_EINVAL;
    return staged_entry(d, n, width, height, isect_offsets, n_isects, v_render_colors, blk_rows, rowmap, rows, fwd_scratch,
                        fwd_scratch_bytes, scratch, scratch_bytes, v_colors, stage, ch_begin, ch_count, nullptr, nullptr, nullptr,
                        keep_prev, keep_cur, stream);
}

extern "C" int gags_raster_bwd_colors_staged_range(int d, int n, int width, int height, const int32_t *isect_offsets,
                                                   int64_t n_isects, const float *v_render_colors,
                                                   const int32_t *blk_rows, const int32_t *rowmap, int64_t rows,
                                                   const void *fwd_scratch, int64_t fwd_scratch_bytes, void *scratch,
                                                   int64_t scratch_bytes, float *v_colors, int stage, int ch_begin,
                                                   int ch_count, void *stream)
{
    return gags_raster_bwd_colors_staged_cap(d, n, width, height, isect_offsets, n_isects, v_render_colors, blk_rows, rowmap, rows,
                                             fwd_scratch, fwd_scratch_bytes, scratch, scratch_bytes, v_colors, stage, ch_begin,
                                             ch_count, nullptr, stream);
}

extern "C" int gags_raster_bwd_colors_staged(int d, int n, int width, int height, const int32_t *isect_offsets,
                                             int64_t n_isects, const float *v_render_colors, const int32_t *blk_rows,
                                             const int32_t *rowmap, int64_t rows, const void *fwd_scratch,
                                             int64_t fwd_scratch_bytes, void *scratch, int64_t scratch_bytes,
                                             float *v_colors, int stage, void *stream)
{
    return gags_raster_bwd_colors_staged_range(d, n, width, height, isect_offsets, n_isects, v_render_colors, blk_rows,
                                               rowmap, rows, fwd_scratch, fwd_scratch_bytes, scratch, scratch_bytes,
                                               v_colors, stage, 0, d, stream);
}
