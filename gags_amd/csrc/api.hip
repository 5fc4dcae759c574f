// Dispatch of the two raster entry points onto the VALU (narrow D) and MFMA (wide D) kernels.
#include "common.h"

int gags_raster_fwd_valu(int d, int width, int height, const float *means2d, const float *conics,
                         const float *opacities, const float *colors, const float *backgrounds,
                         const int32_t *offsets, const int32_t *flat, int n_isects, float *out, float *alphas,
                         int32_t *last_ids, hipStream_t st);
int gags_raster_bwd_valu(int d, int width, int height, const float *means2d, const float *conics,
                         const float *opacities, const float *colors, const float *backgrounds,
                         const int32_t *offsets, const int32_t *flat, int n_isects, const float *alphas,
                         const int32_t *last_ids, const float *v_out, const float *v_alpha, float *v_colors,
                         float *v_opac, float *v_m2d, float *v_con, bool geom, hipStream_t st);

int gags_raster_fwd_mfma(int d, int width, int height, const void *packed, const float *colors,
                         const float *backgrounds, const int32_t *offsets, const int32_t *flat, int n_isects,
                         float *out, float *alphas, int32_t *last_ids, int32_t *blk_rows, int dbg, hipStream_t st);
int64_t gags_bwd_staged_scratch_bytes_impl(int64_t rows, int64_t n_isects, int n_gauss, int d);
int gags_raster_bwd_colors_staged(int d, int width, int height, int n_gauss, const void *packed,
                                  const int32_t *offsets, const int32_t *flat, int n_isects, const float *v_out,
                                  const int32_t *blk_rows, const int32_t *row_end, int64_t rows, void *scratch,
                                  int64_t scratch_bytes, float *v_colors, int stage, hipStream_t st);
int gags_pack_isects_launch(int n_isects, const int32_t *flat, const float *means2d, const float *conics,
                            const float *opacities, void *packed, hipStream_t st);

int gags_raster_bwd_colors_mfma(int d, int width, int height, const void *packed, const int32_t *offsets,
                                const int32_t *flat, int n_isects, const float *v_out, float *v_colors, int dbg,
                                hipStream_t st);

extern "C" int gags_raster_fwd(int d, int width, int height, const float *means2d, const float *conics,
                               const float *opacities, const float *colors, const float *backgrounds,
                               const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                               const void *packed, float *render_colors, float *render_alphas, int32_t *last_ids,
                               int32_t *blk_rows, int flags, void *stream)
{
    GAGS_CLEAR_ERR();
    if (d <= 0 || width <= 0 || height <= 0 || n_isects < 0 || n_isects >= (1ll << 31)) return GAGS_EINVAL;
    if (!isect_offsets || !render_colors || !render_alphas || !last_ids) return GAGS_EINVAL;
    if (n_isects > 0 && (!means2d || !conics || !opacities || !colors || !flatten_ids)) return GAGS_EINVAL;
    if (!(flags & GAGS_FWD_NO_MFMA) && (packed || n_isects == 0)) {
        const int rc = gags_raster_fwd_mfma(d, width, height, packed, colors, backgrounds, isect_offsets, flatten_ids,
                                            (int)n_isects, render_colors, render_alphas, last_ids, blk_rows,
                                            flags >> 8, (hipStream_t)stream);
        if (rc != 1) return rc;  // taken (GAGS_OK) or failed (<0); 1 = width not eligible
    }
    return gags_raster_fwd_valu(d, width, height, means2d, conics, opacities, colors, backgrounds, isect_offsets,
                                flatten_ids, (int)n_isects, render_colors, render_alphas, last_ids,
                                (hipStream_t)stream);
}

extern "C" int gags_raster_bwd(int d, int width, int height, const float *means2d, const float *conics,
                               const float *opacities, const float *colors, const float *backgrounds,
                               const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                               const void *packed, const float *render_alphas, const int32_t *last_ids,
                               const float *v_render_colors,
                               const float *v_render_alphas, float *v_colors, float *v_opacities, float *v_means2d,
                               float *v_conics, int flags, void *stream)
{
    GAGS_CLEAR_ERR();
    if (d <= 0 || width <= 0 || height <= 0 || n_isects < 0 || n_isects >= (1ll << 31)) return GAGS_EINVAL;
    if (n_isects == 0) return GAGS_OK;
    if (!means2d || !conics || !opacities || !colors || !isect_offsets || !flatten_ids || !render_alphas ||
        !last_ids || !v_render_colors || !v_colors)
        return GAGS_EINVAL;
    const bool geom = !(flags & GAGS_BWD_COLORS_ONLY);
    if (geom && (!v_opacities || !v_means2d || !v_conics)) return GAGS_EINVAL;
    if (!geom && !(flags & GAGS_FWD_NO_MFMA) && packed) {
        const int rc = gags_raster_bwd_colors_mfma(d, width, height, packed, isect_offsets, flatten_ids,
                                                   (int)n_isects, v_render_colors, v_colors, flags >> 8,
                                                   (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    return gags_raster_bwd_valu(d, width, height, means2d, conics, opacities, colors, backgrounds, isect_offsets,
                                flatten_ids, (int)n_isects, render_alphas, last_ids, v_render_colors,
                                v_render_alphas, v_colors, v_opacities, v_means2d, v_conics, geom,
                                (hipStream_t)stream);
}

extern "C" int gags_pack_isects(int64_t n_isects, const int32_t *flatten_ids, const float *means2d,
                                const float *conics, const float *opacities, void *packed, void *stream)
{
    if (n_isects < 0 || n_isects >= (1ll << 31)) return GAGS_EINVAL;
    if (n_isects == 0) return GAGS_OK;
    if (!flatten_ids || !means2d || !conics || !opacities || !packed) return GAGS_EINVAL;
    return gags_pack_isects_launch((int)n_isects, flatten_ids, means2d, conics, opacities, packed,
                                   (hipStream_t)stream);
}

extern "C" int64_t gags_bwd_staged_scratch_bytes(int64_t rows, int64_t n_isects, int n, int d)
{
    if (rows < 0 || n_isects < 0 || n < 0 || d <= 0) return 0;
    return gags_bwd_staged_scratch_bytes_impl(rows, n_isects, n, d);
}

extern "C" int gags_raster_bwd_colors_staged(int d, int width, int height, int n, const void *packed,
                                             const int32_t *isect_offsets, const int32_t *flatten_ids,
                                             int64_t n_isects, const float *v_render_colors, const int32_t *blk_rows,
                                             const int32_t *row_end, int64_t rows, void *scratch,
                                             int64_t scratch_bytes, float *v_colors, int stage, void *stream)
{
    if (d <= 0 || width <= 0 || height <= 0 || n < 0 || n_isects < 0 || n_isects >= (1ll << 31) || rows < 0 ||
        rows >= (1ll << 31) || stage < 0 || (stage & 15) > 4)
        return GAGS_EINVAL;
    if (n == 0) return GAGS_OK;
    if (!isect_offsets || !blk_rows || !row_end || !scratch || !v_colors || !v_render_colors) return GAGS_EINVAL;
    if (rows > 0 && (!packed || !flatten_ids)) return GAGS_EINVAL;
    return gags_raster_bwd_colors_staged(d, width, height, n, packed, isect_offsets, flatten_ids, (int)n_isects,
                                         v_render_colors, blk_rows, row_end, rows, scratch, scratch_bytes, v_colors,
                                         stage, (hipStream_t)stream);
}
