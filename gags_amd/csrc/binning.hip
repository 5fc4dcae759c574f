// K5 / K6 / K8: prefix sum, intersection emit, per-tile offsets.  All HBM-bound integer work
// (SURVEY.md 8a R5); coalesced, wave64 scans, no GEMM reshaping.
#include "common.h"
#include "scan.h"

namespace {

__global__ __launch_bounds__(256) void tile_emit_kernel(int n, const float *__restrict__ means2d,
                                                        const int32_t *__restrict__ radii,
                                                        const float *__restrict__ depths,
                                                        const int32_t *__restrict__ cum,
                                                        const int32_t *__restrict__ order, int tile_w, int tile_h,
                                                        int64_t *__restrict__ isect_ids,
                                                        int32_t *__restrict__ flatten_ids, int64_t cap)
{
    const int j = blockIdx.x * 256 + threadIdx.x;  // position in the emission order
    if (j >= n) return;
    const int i = order ? order[j] : j;            // Gaussian
    const int rad = radii[i];
    if (rad <= 0) return;
    const float2 m = reinterpret_cast<const float2 *>(means2d)[i];
    int x0, x1, y0, y1;
    gags_tile_aabb(m.x, m.y, rad, tile_w, tile_h, x0, x1, y0, y1);
    int cur = (j == 0) ? 0 : cum[j - 1];
    const int64_t dbits = (int64_t)(uint32_t)__float_as_int(depths[i]);
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) {
            const int64_t tile_id = (int64_t)ty * tile_w + tx;
            if (cur < cap) {  // (capacity-sized buffers: a count above the capacity is detected by the caller afterwards)
                isect_ids[cur] = (tile_id << 32) | dbits;
                flatten_ids[cur] = i;
            }
            ++cur;
        }
}

// capacity-sized buffers: entries [total, cap) become sentinels of a tile past the last one (they sort to the end and
// give isect_offsets its final entry = the true count) pointing at Gaussian 0 (never read: no tile's range holds them)
__global__ __launch_bounds__(256) void isect_tail_kernel(int64_t cap, const int32_t *__restrict__ total, int n_tiles,
                                                         int64_t *__restrict__ isect_ids, int32_t *__restrict__ flatten_ids)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cap || i < (int64_t)total[0]) return;
    isect_ids[i] = (int64_t)n_tiles << 32;
    flatten_ids[i] = 0;
}

__global__ __launch_bounds__(256) void tile_offsets_kernel(int64_t n_isects, const int64_t *__restrict__ sorted_ids,
                                                           int n_tiles, int32_t *__restrict__ offsets)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_isects) return;
    const int cur = (int)((uint64_t)sorted_ids[idx] >> 32);
    if (idx == 0) {
        for (int t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
        const int prev = (int)((uint64_t)sorted_ids[idx - 1] >> 32);
        for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)idx;
    }
    if (idx == n_isects - 1)
        for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n_isects;
}

__global__ void fill_i32_kernel(int n, int32_t v, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

__global__ __launch_bounds__(256) void ed_normalize_kernel(int64_t n_pix, int d, float *__restrict__ colors,
                                                           const float *__restrict__ alphas)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix) return;
    colors[i * d + (d - 1)] = colors[i * d + (d - 1)] / fmaxf(alphas[i], 1e-10f);
}

}  // namespace

extern "C" int gags_abi_version(void) { return 2; }  // 2: isect_offsets has n_tiles + 1 entries (the last one = n_isects)

extern "C" const char *gags_strerror(int code)
{
    switch (code) {
        case GAGS_OK: return "ok";
        case GAGS_EINVAL: return "invalid argument";
        case GAGS_ELAUNCH: return "HIP launch/runtime error";
        case GAGS_ESCRATCH: return "scratch buffer too small";
        case GAGS_ENODEV: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

extern "C" int gags_device_count(void)
{
    GAGS_CLEAR_ERR();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return GAGS_ENODEV;
    return n;
}

extern "C" int64_t gags_scan_scratch_bytes(int n)
{
    return gags_scan::scratch_bytes(n);
}

extern "C" int gags_cumsum_i32(int n, const int32_t *in, int32_t *cum, int32_t *total, void *scratch,
                               int64_t scratch_bytes, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (total) hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, 1, 0, total);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (!in || !cum || !scratch) return GAGS_EINVAL;
    if (scratch_bytes < gags_scan_scratch_bytes(n)) return GAGS_ESCRATCH;
    gags_scan::launch<false>(n, in, cum, total, (int32_t *)scratch, st);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_read_i32(const int32_t *src, int32_t *dst_host, void *stream)
{
    GAGS_CLEAR_ERR();
    if (!src || !dst_host) return GAGS_EINVAL;
    if (hipMemcpyAsync(dst_host, src, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        return GAGS_ELAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GAGS_ELAUNCH;
    return GAGS_OK;
}

extern "C" int gags_tile_emit_cap(int n, const float *means2d, const int32_t *radii, const float *depths, const int32_t *cum,
                                  const int32_t *order, int tile_w, int tile_h, int64_t *isect_ids, int32_t *flatten_ids,
                                  int64_t cap, const int32_t *total, void *stream)
{
    GAGS_CLEAR_ERR();
    if (n < 0 || tile_w <= 0 || tile_h <= 0 || cap < 0 || cap >= (1ll << 31)) return GAGS_EINVAL;
    if (n == 0 && cap == 0) return GAGS_OK;
    if (!isect_ids || !flatten_ids || (n > 0 && (!means2d || !radii || !depths || !cum))) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n > 0)
        hipLaunchKernelGGL(tile_emit_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, means2d, radii, depths, cum, order, tile_w,
                           tile_h, isect_ids, flatten_ids, cap);
    if (total && cap > 0)
        hipLaunchKernelGGL(isect_tail_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, cap, total, tile_w * tile_h,
                           isect_ids, flatten_ids);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_tile_emit(int n, const float *means2d, const int32_t *radii, const float *depths,
                              const int32_t *cum, const int32_t *order, int tile_w, int tile_h, int64_t *isect_ids,
                              int32_t *flatten_ids, void *stream)
{
    if (n == 0) return GAGS_OK;
    return gags_tile_emit_cap(n, means2d, radii, depths, cum, order, tile_w, tile_h, isect_ids, flatten_ids, (1ll << 31) - 1, nullptr,
                              stream);
}

extern "C" int gags_tile_offsets(int64_t n_isects, const int64_t *sorted_ids, int n_tiles, int32_t *isect_offsets,
                                 void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_isects < 0 || n_tiles <= 0 || !isect_offsets) return GAGS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // n_tiles + 1 entries: [n_tiles] = the intersection count (with capacity-sized inputs: where the sentinel keys of
    // tile `n_tiles` begin), so that no raster kernel needs the count from the host
    if (n_isects == 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((n_tiles + 1 + 255) / 256), dim3(256), 0, st, n_tiles + 1, 0, isect_offsets);
        GAGS_CHECK_LAUNCH();
        return GAGS_OK;
    }
    if (!sorted_ids) return GAGS_EINVAL;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((unsigned)((n_isects + 255) / 256)), dim3(256), 0, st, n_isects,
                       sorted_ids, n_tiles + 1, isect_offsets);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_ed_normalize(int64_t n_pix, int d, float *render_colors, const float *render_alphas,
                                 void *stream)
{
    GAGS_CLEAR_ERR();
    if (n_pix < 0 || d <= 0) return GAGS_EINVAL;
    if (n_pix == 0) return GAGS_OK;
    if (!render_colors || !render_alphas) return GAGS_EINVAL;
    hipLaunchKernelGGL(ed_normalize_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_pix, d, render_colors, render_alphas);
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
