// Microbenchmark behind the shape of the weights pass: how should the transmittance chain
// T_g = prod_{j < g} (1 - alpha_j) of a pixel be evaluated on a 64-wide wavefront?
//   A  "lane = pixel":    every lane owns a pixel and walks the tile's depth-sorted Gaussians one by one -- the chain is a
//                         sequential product per lane, exactly the reference's order (bit-reproducible against a scalar loop);
//   B  "lane = Gaussian": 64 Gaussians of the list at a time for ONE pixel, alpha per lane, the chain as a wavefront
//                         prefix product over the lanes (six DPP multiply steps), the running T carried across the
//                         chunks of 64, the stop rule by ballot.  north_star names this form ("wavefront-prefix-scan alpha
//                         compositing"); its association order differs from the sequential product, so T -- and with it
//                         the knife-edge decisions T <= 1e-4 -- is not reproducible against the scalar restatement.
// Both variants evaluate the same alpha (same polynomial exp) for every (pixel, Gaussian) pair of a 64-pixel block and
// L Gaussians and leave alpha * T summed per pixel.  Reported: ms per view-sized launch, SIMD cycles per 64 pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__device__ __forceinline__ float exp_neg(float sigma)
{
    float t = sigma * -1.44269504088896341f;
    t = fmaxf(t, -125.0f);
    const float n = __builtin_rintf(t);
    const float f = t - n;
    float p = 0x1.444p-13f;
    p = __builtin_fmaf(p, f, 0x1.5f48cp-10f);
    p = __builtin_fmaf(p, f, 0x1.3b2a1cp-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aeccp-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbep-3f);
    p = __builtin_fmaf(p, f, 0x1.62e43p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

struct Rec { float x, y, a, b, c, o, pad0, pad1; };

__device__ __forceinline__ float alpha_of(const Rec &r, float px, float py)
{
    const float dx = r.x - px, dy = r.y - py;
    const float sigma = 0.5f * (r.a * dx * dx + r.c * dy * dy) + r.b * dx * dy;
    const float alpha = fminf(0.999f, r.o * exp_neg(sigma));
    return (sigma < 0.f || alpha < 1.0f / 255.0f) ? 0.f : alpha;
}

// A: lane = pixel of an 8x8 block; records broadcast from LDS
__global__ __launch_bounds__(64, 8) void chain_lane_pixel(int L, const Rec *__restrict__ recs, float *__restrict__ out)
{
    __shared__ Rec ring[64];
    const int lane = threadIdx.x;
    const float px = (float)(lane & 7) + 0.5f, py = (float)(lane >> 3) + 0.5f;
    const Rec *mine = recs + (size_t)blockIdx.x * L;
    float T = 1.f, acc = 0.f;
    bool done = false;
    for (int base = 0; base < L; base += 64) {
        ring[lane] = mine[min(base + lane, L - 1)];
        __builtin_amdgcn_wave_barrier();
        const int m = min(64, L - base);
        for (int j = 0; j < m; ++j) {
            const float a = alpha_of(ring[j], px, py);
            const float t1 = T * (1.0f - a);
            const bool ok = !done && a > 0.f;
            const bool stop = ok && t1 <= 1e-4f;
            const bool bl = ok && !stop;
            acc += bl ? a * T : 0.f;
            T = bl ? t1 : T;
            done = done || stop;
        }
        __builtin_amdgcn_wave_barrier();
        if (__all(done)) break;
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc + T;
}

// inclusive prefix product over the 64 lanes: row_shr 1, 2, 4, 8 inside rows of 16, then row_bcast:15 and row_bcast:31
__device__ __forceinline__ float wave_prefix_prod(float v)
{
#define DPP_MUL(ctrl, rmask, bmask)                                                                               \
    {                                                                                                            \
        const int t = __builtin_amdgcn_update_dpp(__float_as_int(1.0f), __float_as_int(v), ctrl, rmask, bmask, false); \
        v *= __int_as_float(t);                                                                                  \
    }
    DPP_MUL(0x111, 0xf, 0xf)  // row_shr:1
    DPP_MUL(0x112, 0xf, 0xf)  // row_shr:2
    DPP_MUL(0x114, 0xf, 0xf)  // row_shr:4
    DPP_MUL(0x118, 0xf, 0xf)  // row_shr:8
    DPP_MUL(0x142, 0xa, 0xf)  // row_bcast:15 -> rows 1 and 3
    DPP_MUL(0x143, 0xc, 0xf)  // row_bcast:31 -> rows 2 and 3
#undef DPP_MUL
    return v;
}

// B: lane = Gaussian; the block's 64 pixels one after the other
__global__ __launch_bounds__(64, 8) void chain_lane_gauss(int L, const Rec *__restrict__ recs, float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const Rec *mine = recs + (size_t)blockIdx.x * L;
    float res = 0.f;
    for (int p = 0; p < 64; ++p) {
        const float px = (float)(p & 7) + 0.5f, py = (float)(p >> 3) + 0.5f;
        float Tin = 1.f, acc = 0.f;
        for (int base = 0; base < L; base += 64) {
            const bool live = base + lane < L;
            const Rec r = mine[min(base + lane, L - 1)];
            const float a = live ? alpha_of(r, px, py) : 0.f;
            const float incl = wave_prefix_prod(1.0f - a) * Tin;  // T after this lane's Gaussian
            // stop rule: the first lane whose T falls to 1e-4 (and every later one) does not blend
            const unsigned long long stopm = __ballot(a > 0.f && incl <= 1e-4f);
            const int first = stopm ? __ffsll((long long)stopm) - 1 : 64;
            const float excl = incl / fmaxf(1.0f - a, 1e-30f);    // T before it (a <= 0.999)
            float w = (lane < first && a > 0.f) ? a * excl : 0.f;
            // sum of the weights over the wave (what variant A's lane accumulates for its pixel)
            for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
            acc += w;
            const int lastl = first < 64 ? max(first - 1, 0) : 63;
            Tin = (first == 0) ? Tin : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), lastl));
            if (first < 64) break;
        }
        if (lane == p) res = acc + Tin;
    }
    out[(size_t)blockIdx.x * 64 + lane] = res;
}

int main()
{
    const int blocks = 8160 * 4;  // the 8x8 blocks of a 1080p view
    for (int L : {64, 256, 900}) {
        for (float opac : {0.05f, 0.5f}) {  // faint splats (no pixel saturates) / strong ones (early stop)
            std::vector<Rec> h((size_t)blocks * L);
            unsigned s = 12345;
            auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
            for (auto &r : h) {
                r.x = 8.f * rnd(); r.y = 8.f * rnd();
                const float sg = 1.0f + 3.0f * rnd();
                r.a = 1.0f / (sg * sg); r.c = 1.0f / (sg * sg); r.b = 0.f; r.o = opac * (0.5f + rnd());
                r.pad0 = r.pad1 = 0.f;
            }
            Rec *d; float *o;
            (void)hipMalloc(&d, h.size() * sizeof(Rec)); (void)hipMalloc(&o, (size_t)blocks * 64 * 4);
            (void)hipMemcpy(d, h.data(), h.size() * sizeof(Rec), hipMemcpyHostToDevice);
            float ms[2];
            std::vector<float> ra((size_t)blocks * 64), rb((size_t)blocks * 64);
            for (int v = 0; v < 2; ++v) {
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(e0);
                    if (v == 0) hipLaunchKernelGGL(chain_lane_pixel, dim3(blocks), dim3(64), 0, 0, L, d, o);
                    else hipLaunchKernelGGL(chain_lane_gauss, dim3(blocks), dim3(64), 0, 0, L, d, o);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                }
                (void)hipEventElapsedTime(&ms[v], e0, e1);
                (void)hipMemcpy((v ? rb : ra).data(), o, (size_t)blocks * 64 * 4, hipMemcpyDeviceToHost);
            }
            double diff = 0, differ = 0;
            for (size_t i = 0; i < ra.size(); ++i) { diff = fmax(diff, fabs(ra[i] - rb[i])); differ += ra[i] != rb[i]; }
            const double pairs = (double)blocks * 64 * L;
            printf("L=%4d opacity~%.2f  lane=pixel %.3f ms (%.1f SIMD cycles per 64 pairs)  lane=Gaussian+prefix scan %.3f ms (%.1f)  "
                   "max |difference| %.2e, %.1f %% of the pixels differ in the last bits\n", L, opac, ms[0],
                   ms[0] * 1e-3 * 2.4e9 * 1024 / pairs * 64, ms[1], ms[1] * 1e-3 * 2.4e9 * 1024 / pairs * 64, diff,
                   100.0 * differ / ra.size());
            (void)hipFree(d); (void)hipFree(o);
        }
    }
    return 0;
}
