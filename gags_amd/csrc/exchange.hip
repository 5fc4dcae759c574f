// Pack / unpack of the gradient rows a by-view step exchanges ("gradients are sparse in rows").
// A view's feature gradient lives in the rows of the Gaussians that blended into one of its pixels; the ranks agree on
// the union of those rows and every channel range [c0, c0 + cw) of the gradient travels as the dense block
// [rows, cw] (gags_amd/dist.py: OverlappedGradReducer).  Both kernels are plain HBM streams: one lane per 4
// channels, consecutive lanes on consecutive channels of a row.
//   pack   : wire[r, :] = grad[idx[r], c0 : c0 + cw]                      (idx NULL: every row)
//   unpack : grad[idx[r], c0 : c0 + cw]  = wire[r, :]                     (local NULL: assign)
//            grad[idx[r], c0 : c0 + cw] += wire[r, :] - local[r, :]       (local: the packed rows before the sum --
//                                                                          a gradient that already holds other terms)
// Element types: 0 = fp32, 1 = fp16 (the gradient of an fp16 feature table), 2 = bf16 (the opt-in 16-bit wire).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include "common.h"

namespace {

template <int T> struct Elem;
template <> struct Elem<0> {
    using type = float;
    static __device__ __forceinline__ float ld(const void *p, size_t i) { return reinterpret_cast<const float *>(p)[i]; }
    static __device__ __forceinline__ void st(void *p, size_t i, float v) { reinterpret_cast<float *>(p)[i] = v; }
};
template <> struct Elem<1> {
    using type = __half;
    static __device__ __forceinline__ float ld(const void *p, size_t i) { return __half2float(reinterpret_cast<const __half *>(p)[i]); }
    static __device__ __forceinline__ void st(void *p, size_t i, float v) { reinterpret_cast<__half *>(p)[i] = __float2half_rn(v); }
};
template <> struct Elem<2> {
    using type = __hip_bfloat16;
    static __device__ __forceinline__ float ld(const void *p, size_t i) { return __bfloat162float(reinterpret_cast<const __hip_bfloat16 *>(p)[i]); }
    static __device__ __forceinline__ void st(void *p, size_t i, float v) { reinterpret_cast<__hip_bfloat16 *>(p)[i] = __float2bfloat16(v); }
};

// one thread per (row, 4 channels); the 4 channels are handled element-wise (rows of an odd width D are not
// 16-byte aligned; the compiler still merges the accesses where the types allow)
template <int TG, int TW>
__global__ __launch_bounds__(256) void pack_rows_kernel(int64_t n_rows, const int64_t *__restrict__ idx,
                                                        const void *__restrict__ grad, int d, int c0, int cw,
                                                        void *__restrict__ wire)
{
    const int q = (cw + 3) >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rows * q) return;
    const int64_t r = t / q;
    const int c = (int)(t - r * q) * 4;
    const int64_t g = idx ? idx[r] : r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e < cw) Elem<TW>::st(wire, (size_t)r * cw + c + e, Elem<TG>::ld(grad, (size_t)g * d + c0 + c + e));
}

template <int TG, int TW, bool DELTA>
__global__ __launch_bounds__(256) void unpack_rows_kernel(int64_t n_rows, const int64_t *__restrict__ idx,
                                                          const void *__restrict__ wire, const void *__restrict__ local,
                                                          void *__restrict__ grad, int d, int c0, int cw)
{
    const int q = (cw + 3) >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rows * q) return;
    const int64_t r = t / q;
    const int c = (int)(t - r * q) * 4;
    const int64_t g = idx ? idx[r] : r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (c + e >= cw) continue;
        const size_t wi = (size_t)r * cw + c + e, gi = (size_t)g * d + c0 + c + e;
        float v = Elem<TW>::ld(wire, wi);
        if constexpr (DELTA) v = Elem<TG>::ld(grad, gi) + (v - Elem<TW>::ld(local, wi));
        Elem<TG>::st(grad, gi, v);
    }
}

inline bool ok_types(int tg, int tw) { return (tg == 0 || tg == 1) && (tw == 0 || tw == 1 || tw == 2); }

}  // namespace

extern "C" int gags_pack_rows(int64_t n_rows, const int64_t *idx, const void *grad, int grad_type, int d, int c0, int cw,
                              void *wire, int wire_type, void *stream)
{
    if (n_rows < 0 || d <= 0 || c0 < 0 || cw <= 0 || c0 + cw > d || !ok_types(grad_type, wire_type)) return GAGS_EINVAL;
    if (n_rows == 0) return GAGS_OK;
    if (!grad || !wire) return GAGS_EINVAL;
    GAGS_CLEAR_ERR();
    const int64_t items = n_rows * ((cw + 3) >> 2);
    const dim3 grid((unsigned)((items + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define GO(TG, TW) hipLaunchKernelGGL((pack_rows_kernel<TG, TW>), grid, dim3(256), 0, st, n_rows, idx, grad, d, c0, cw, wire)
    if (grad_type == 0) { if (wire_type == 0) GO(0, 0); else if (wire_type == 1) GO(0, 1); else GO(0, 2); }
    else { if (wire_type == 0) GO(1, 0); else if (wire_type == 1) GO(1, 1); else GO(1, 2); }
#undef GO
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}

extern "C" int gags_unpack_rows(int64_t n_rows, const int64_t *idx, const void *wire, int wire_type, const void *local,
                                void *grad, int grad_type, int d, int c0, int cw, void *stream)
{
    if (n_rows < 0 || d <= 0 || c0 < 0 || cw <= 0 || c0 + cw > d || !ok_types(grad_type, wire_type)) return GAGS_EINVAL;
    if (n_rows == 0) return GAGS_OK;
    if (!grad || !wire) return GAGS_EINVAL;
    GAGS_CLEAR_ERR();
    const int64_t items = n_rows * ((cw + 3) >> 2);
    const dim3 grid((unsigned)((items + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define GO(TG, TW)                                                                                                         \
    do {                                                                                                                   \
        if (local) hipLaunchKernelGGL((unpack_rows_kernel<TG, TW, true>), grid, dim3(256), 0, st, n_rows, idx, wire, local, grad, d, c0, cw); \
        else hipLaunchKernelGGL((unpack_rows_kernel<TG, TW, false>), grid, dim3(256), 0, st, n_rows, idx, wire, local, grad, d, c0, cw);      \
    } while (0)
    if (grad_type == 0) { if (wire_type == 0) GO(0, 0); else if (wire_type == 1) GO(0, 1); else GO(0, 2); }
    else { if (wire_type == 0) GO(1, 0); else if (wire_type == 1) GO(1, 1); else GO(1, 2); }
#undef GO
    GAGS_CHECK_LAUNCH();
    return GAGS_OK;
}
