/* CPU twins of the C ABI (include/gags_cpu.h): the entry points of include/gags_raster.h restated on HOST pointers by
 * calling oracle/gags_oracle.c, argument for argument.  TEST INFRASTRUCTURE ONLY, same rules and same parity status as the
 * oracle itself ("parity unpinned": see the header of gags_oracle.c).  Each twin cites the entry point it mirrors. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/gags_cpu.h"

#define CPU_OK 0
#define CPU_EINVAL (-1)
#define CPU_TILE 16
#define CPU_BWD_COLORS_ONLY 1

/* gags_oracle.c */
void orc_project_fwd(int N, const float *means, const float *quats, const float *scales, const float *viewmat,
                     const float *Kmat, int width, int height, float eps2d, float near_plane, float far_plane,
                     float radius_clip, int32_t *radii, float *means2d, float *depths, float *conics);
void orc_tile_count(int N, const float *means2d, const int32_t *radii, int tile_w, int tile_h, int32_t *tiles_per_gauss);
void orc_sort_pairs(int64_t n, int nbits, const int64_t *keys_in, const int32_t *vals_in, int64_t *keys_out, int32_t *vals_out);
void orc_tile_offsets(int64_t n_isects, const int64_t *sorted_ids, int n_tiles, int32_t *offsets);
void orc_raster_fwd(int D, int width, int height, int tile_w, int tile_h, const float *means2d, const float *conics,
                    const float *opacities, const float *colors, const float *backgrounds, const int32_t *tile_offsets,
                    const int32_t *flatten_ids, int64_t n_isects, int tile_begin, int tile_step, float *render_colors,
                    float *render_alphas, int32_t *last_ids, int64_t *n_eval, int64_t *n_blend);
void orc_raster_bwd(int D, int width, int height, int tile_w, int tile_h, const float *means2d, const float *conics,
                    const float *opacities, const float *colors, const float *backgrounds, const int32_t *tile_offsets,
                    const int32_t *flatten_ids, int64_t n_isects, const float *render_alphas, const int32_t *last_ids,
                    const float *v_render_colors, const float *v_render_alphas, int tile_begin, int tile_step,
                    float *v_colors, float *v_opacities, float *v_means2d, float *v_conics);
void orc_project_bwd(int N, const float *means, const float *quats, const float *scales, const float *viewmat,
                     const float *Kmat, int width, int height, float eps2d, const int32_t *radii, const float *v_means2d,
                     const float *v_depths, const float *v_conics, float *v_means, float *v_quats, float *v_scales);
void orc_sh_fwd(int N, int Kc, int deg, const float *means, const float *campos, const float *coeffs, const int32_t *radii,
                float *out);
void orc_adam_step(int64_t n, float *p, const float *g, float *m, float *v, double lr, double beta1, double beta2, double eps,
                   int step);

/* gags_project_fwd (include/gags_raster.h; K1 + K4) */
int gags_cpu_project_fwd(int n, const float *means, const float *quats, const float *scales, const float *viewmat,
                         const float *K, int width, int height, float eps2d, float near_plane, float far_plane,
                         float radius_clip, int32_t *radii, float *means2d, float *depths, float *conics,
                         int32_t *tiles_per_gauss, void *stream)
{
    (void)stream;
    if (n < 0 || width <= 0 || height <= 0) return CPU_EINVAL;
    if (n == 0) return CPU_OK;
    if (!means || !quats || !scales || !viewmat || !K || !radii || !means2d || !depths || !conics || !tiles_per_gauss)
        return CPU_EINVAL;
    orc_project_fwd(n, means, quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip, radii,
                    means2d, depths, conics);
    orc_tile_count(n, means2d, radii, (width + CPU_TILE - 1) / CPU_TILE, (height + CPU_TILE - 1) / CPU_TILE, tiles_per_gauss);
    return CPU_OK;
}

int64_t gags_cpu_scan_scratch_bytes(int n) { (void)n; return 0; }

/* gags_cumsum_i32 (K5): total = -1 when the sum does not fit an int32 */
int gags_cpu_cumsum_i32(int n, const int32_t *in, int32_t *cum, int32_t *total, void *scratch, int64_t scratch_bytes,
                        void *stream)
{
    (void)scratch; (void)scratch_bytes; (void)stream;
    if (n < 0 || !total || (n > 0 && (!in || !cum))) return CPU_EINVAL;
    int64_t s = 0;
    for (int i = 0; i < n; ++i) { s += in[i]; cum[i] = (int32_t)s; }
    total[0] = s > 0x7fffffffll ? -1 : (int32_t)s;
    return CPU_OK;
}

/* gags_tile_emit (K6): Gaussian after Gaussian in the order given (order == NULL: by index); cum = inclusive prefix sum
 * of the tile counts IN THAT ORDER */
int gags_cpu_tile_emit(int n, const float *means2d, const int32_t *radii, const float *depths, const int32_t *cum,
                       const int32_t *order, int tile_w, int tile_h, int64_t *isect_ids, int32_t *flatten_ids, void *stream)
{
    (void)stream;
    if (n < 0 || tile_w <= 0 || tile_h <= 0) return CPU_EINVAL;
    if (n == 0) return CPU_OK;
    if (!means2d || !radii || !depths || !cum || !isect_ids || !flatten_ids) return CPU_EINVAL;
    for (int j = 0; j < n; ++j) {
        const int i = order ? order[j] : j;
        if (radii[i] <= 0) continue;
        const float tr = (float)radii[i] / (float)CPU_TILE, tx = means2d[2 * i] / (float)CPU_TILE,
                    ty = means2d[2 * i + 1] / (float)CPU_TILE;
        int x0 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(tx - tr), 0.f), (float)tile_w);
        int x1 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_ceilf(tx + tr), 0.f), (float)tile_w);
        int y0 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(ty - tr), 0.f), (float)tile_h);
        int y1 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_ceilf(ty + tr), 0.f), (float)tile_h);
        int64_t cur = j == 0 ? 0 : cum[j - 1];
        uint32_t dbits;
        memcpy(&dbits, &depths[i], 4);
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                isect_ids[cur] = (((int64_t)y * tile_w + x) << 32) | (int64_t)dbits;
                flatten_ids[cur] = i;
                ++cur;
            }
    }
    return CPU_OK;
}

int64_t gags_cpu_sort_scratch_bytes(int64_t n_isects) { (void)n_isects; return 0; }

/* gags_sort_pairs (K7): stable on key bits [0, 32 + tile_bits); an input already in depth order gives the same result */
int gags_cpu_sort_pairs(int64_t n_isects, int tile_bits, int depth_sorted, const int64_t *keys_in, const int32_t *vals_in,
                        int64_t *keys_out, int32_t *vals_out, void *scratch, int64_t scratch_bytes, void *stream)
{
    (void)depth_sorted; (void)scratch; (void)scratch_bytes; (void)stream;
    if (n_isects < 0 || tile_bits < 0 || tile_bits > 31) return CPU_EINVAL;
    if (n_isects == 0) return CPU_OK;
    if (!keys_in || !vals_in || !keys_out || !vals_out) return CPU_EINVAL;
    orc_sort_pairs(n_isects, 32 + ((tile_bits + 7) / 8) * 8, keys_in, vals_in, keys_out, vals_out);
    return CPU_OK;
}

/* gags_tile_offsets (K8): n_tiles + 1 entries, the last one = the intersection count (ABI version 2) */
int gags_cpu_tile_offsets(int64_t n_isects, const int64_t *sorted_ids, int n_tiles, int32_t *isect_offsets, void *stream)
{
    (void)stream;
    if (n_isects < 0 || n_tiles <= 0 || !isect_offsets || (n_isects > 0 && !sorted_ids)) return CPU_EINVAL;
    orc_tile_offsets(n_isects, sorted_ids, n_tiles, isect_offsets);
    int64_t count = n_isects;  /* (sentinel keys of a capacity-sized input carry tile id n_tiles) */
    while (count > 0 && (int64_t)(((uint64_t)sorted_ids[count - 1]) >> 32) >= n_tiles) --count;
    isect_offsets[n_tiles] = (int32_t)count;
    return CPU_OK;
}

int64_t gags_cpu_raster_fwd_scratch_bytes(int64_t n_isects, int width, int height)
{
    (void)n_isects; (void)width; (void)height;
    return 0;
}

/* gags_raster_fwd (K9) */
int gags_cpu_raster_fwd(int d, int n, int width, int height, const float *means2d, const float *conics,
                        const float *opacities, const float *colors, const float *backgrounds,
                        const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects, const void *packed,
                        float *render_colors, float *render_alphas, int32_t *last_ids, void *scratch, int64_t scratch_bytes,
                        int32_t *blk_rows, int flags, void *stream)
{
    (void)n; (void)packed; (void)scratch; (void)scratch_bytes; (void)blk_rows; (void)flags; (void)stream;
    if (d <= 0 || width <= 0 || height <= 0 || n_isects < 0) return CPU_EINVAL;
    if (!isect_offsets || !render_colors || !render_alphas || !last_ids) return CPU_EINVAL;
    if (n_isects > 0 && (!means2d || !conics || !opacities || !colors || !flatten_ids)) return CPU_EINVAL;
    const int tile_w = (width + CPU_TILE - 1) / CPU_TILE, tile_h = (height + CPU_TILE - 1) / CPU_TILE;
    orc_raster_fwd(d, width, height, tile_w, tile_h, means2d, conics, opacities, colors, backgrounds, isect_offsets,
                   flatten_ids, isect_offsets[tile_w * tile_h], 0, 1, render_colors, render_alphas, last_ids, NULL, NULL);
    return CPU_OK;
}

/* gags_raster_bwd (K10): outputs zero-filled by the caller, results accumulated */
int gags_cpu_raster_bwd(int d, int width, int height, const float *means2d, const float *conics, const float *opacities,
                        const float *colors, const float *backgrounds, const int32_t *isect_offsets,
                        const int32_t *flatten_ids, int64_t n_isects, const void *packed, const float *render_alphas,
                        const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                        float *v_colors, float *v_opacities, float *v_means2d, float *v_conics, int flags, void *stream)
{
    (void)packed; (void)stream;
    if (d <= 0 || width <= 0 || height <= 0 || n_isects < 0) return CPU_EINVAL;
    if (n_isects == 0) return CPU_OK;
    if (!means2d || !conics || !opacities || !colors || !isect_offsets || !flatten_ids || !render_alphas || !last_ids ||
        !v_render_colors || !v_colors)
        return CPU_EINVAL;
    const int geom = !(flags & CPU_BWD_COLORS_ONLY);
    if (geom && (!v_opacities || !v_means2d || !v_conics)) return CPU_EINVAL;
    const int tile_w = (width + CPU_TILE - 1) / CPU_TILE, tile_h = (height + CPU_TILE - 1) / CPU_TILE;
    orc_raster_bwd(d, width, height, tile_w, tile_h, means2d, conics, opacities, colors, backgrounds, isect_offsets,
                   flatten_ids, isect_offsets[tile_w * tile_h], render_alphas, last_ids, v_render_colors, v_render_alphas, 0, 1,
                   v_colors, geom ? v_opacities : NULL, geom ? v_means2d : NULL, geom ? v_conics : NULL);
    return CPU_OK;
}

/* gags_project_bwd (K2) */
int gags_cpu_project_bwd(int n, const float *means, const float *quats, const float *scales, const float *viewmat,
                         const float *K, int width, int height, float eps2d, const int32_t *radii, const float *conics,
                         const float *v_means2d, const float *v_depths, const float *v_conics, float *v_means,
                         float *v_quats, float *v_scales, void *stream)
{
    (void)conics; (void)stream;
    if (n < 0 || width <= 0 || height <= 0) return CPU_EINVAL;
    if (n == 0) return CPU_OK;
    if (!means || !quats || !scales || !viewmat || !K || !radii || !v_means2d || !v_conics || !v_means || !v_quats || !v_scales)
        return CPU_EINVAL;
    orc_project_bwd(n, means, quats, scales, viewmat, K, width, height, eps2d, radii, v_means2d, v_depths, v_conics, v_means,
                    v_quats, v_scales);
    return CPU_OK;
}

/* gags_sh_fwd (K3) */
int gags_cpu_sh_fwd(int n, int kc, int degree, const float *means, const float *campos, const float *coeffs,
                    const int32_t *radii, float *out, void *stream)
{
    (void)stream;
    if (n < 0 || degree < 0 || degree > 3 || kc < (degree + 1) * (degree + 1)) return CPU_EINVAL;
    if (n == 0) return CPU_OK;
    if (!means || !campos || !coeffs || !out) return CPU_EINVAL;
    orc_sh_fwd(n, kc, degree, means, campos, coeffs, radii, out);
    return CPU_OK;
}

/* gags_ed_normalize (K12) */
int gags_cpu_ed_normalize(int64_t n_pix, int d, float *render_colors, const float *render_alphas, void *stream)
{
    (void)stream;
    if (n_pix < 0 || d <= 0 || (n_pix > 0 && (!render_colors || !render_alphas))) return CPU_EINVAL;
    for (int64_t i = 0; i < n_pix; ++i) {
        const float a = render_alphas[i] > 1e-10f ? render_alphas[i] : 1e-10f;
        render_colors[i * d + d - 1] = render_colors[i * d + d - 1] / a;
    }
    return CPU_OK;
}

/* gags_adam_step (R9) */
int gags_cpu_adam_step(int64_t numel, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, double lr,
                       double beta1, double beta2, double eps, int step, void *stream)
{
    (void)stream;
    if (numel < 0 || step < 1 || (numel > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return CPU_EINVAL;
    orc_adam_step(numel, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step);
    return CPU_OK;
}
