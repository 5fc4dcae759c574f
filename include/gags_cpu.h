/*
 * gags_cpu.h -- CPU twins of the core entry points of gags_raster.h, with IDENTICAL signatures ("CPU twins
 * gmi_cpu_* with identical signatures").  TEST INFRASTRUCTURE: they live in oracle/libgags_oracle.so (oracle/gags_cpu.c
 * on top of oracle/gags_oracle.c), never in libgags_hip.so, and the product never loads them.  Their purpose: a caller
 * bound to the C ABI through a table of function pointers / ctypes signatures can be pointed at `gags_cpu_<name>` instead of
 * `gags_<name>` with HOST pointers and A/B'd against the GPU library without changing a call site
 * (tests/test_abi_cpu.py::test_cpu_twins_share_the_c_abi_signatures does exactly that).
 *
 * Every pointer is a HOST pointer; `stream`, `scratch` and `packed` are accepted and ignored; `flags` honours
 * GAGS_BWD_COLORS_ONLY and ignores the kernel-selection bits.  Return codes as in gags_raster.h.
 */
#ifndef GAGS_CPU_H
#define GAGS_CPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int gags_cpu_project_fwd(int n, const float *means, const float *quats, const float *scales,
                         const float *viewmat, const float *K, int width, int height,
                         float eps2d, float near_plane, float far_plane, float radius_clip,
                         int32_t *radii, float *means2d, float *depths, float *conics,
                         int32_t *tiles_per_gauss, void *stream);
int64_t gags_cpu_scan_scratch_bytes(int n);
int gags_cpu_cumsum_i32(int n, const int32_t *in, int32_t *cum, int32_t *total,
                        void *scratch, int64_t scratch_bytes, void *stream);
int gags_cpu_tile_emit(int n, const float *means2d, const int32_t *radii, const float *depths,
                       const int32_t *cum, const int32_t *order, int tile_w, int tile_h,
                       int64_t *isect_ids, int32_t *flatten_ids, void *stream);
int64_t gags_cpu_sort_scratch_bytes(int64_t n_isects);
int gags_cpu_sort_pairs(int64_t n_isects, int tile_bits, int depth_sorted,
                        const int64_t *keys_in, const int32_t *vals_in,
                        int64_t *keys_out, int32_t *vals_out,
                        void *scratch, int64_t scratch_bytes, void *stream);
int gags_cpu_tile_offsets(int64_t n_isects, const int64_t *sorted_ids, int n_tiles,
                          int32_t *isect_offsets, void *stream);
int64_t gags_cpu_raster_fwd_scratch_bytes(int64_t n_isects, int width, int height);
int gags_cpu_raster_fwd(int d, int n, int width, int height, const float *means2d, const float *conics,
                        const float *opacities, const float *colors, const float *backgrounds,
                        const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                        const void *packed,
                        float *render_colors, float *render_alphas, int32_t *last_ids,
                        void *scratch, int64_t scratch_bytes, int32_t *blk_rows,
                        int flags, void *stream);
int gags_cpu_raster_bwd(int d, int width, int height, const float *means2d, const float *conics,
                        const float *opacities, const float *colors, const float *backgrounds,
                        const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                        const void *packed,
                        const float *render_alphas, const int32_t *last_ids,
                        const float *v_render_colors, const float *v_render_alphas,
                        float *v_colors, float *v_opacities, float *v_means2d, float *v_conics,
                        int flags, void *stream);
int gags_cpu_project_bwd(int n, const float *means, const float *quats, const float *scales,
                         const float *viewmat, const float *K, int width, int height, float eps2d,
                         const int32_t *radii, const float *conics,
                         const float *v_means2d, const float *v_depths, const float *v_conics,
                         float *v_means, float *v_quats, float *v_scales, void *stream);
int gags_cpu_sh_fwd(int n, int kc, int degree, const float *means, const float *campos,
                    const float *coeffs, const int32_t *radii, float *out, void *stream);
int gags_cpu_ed_normalize(int64_t n_pix, int d, float *render_colors, const float *render_alphas,
                          void *stream);
int gags_cpu_adam_step(int64_t numel, float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                       double lr, double beta1, double beta2, double eps, int step, void *stream);

#ifdef __cplusplus
}
#endif
#endif
