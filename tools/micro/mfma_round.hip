// How v_mfma_f32_32x32x16_bf16 rounds: probes of the accumulation c + sum_k a_k b_k with hand-picked bf16 operands.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_round.hip -o /tmp/mfma_round && /tmp/mfma_round
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ short bf(float x) { unsigned u = __float_as_uint(x); return (short)(u >> 16); }  // exact for bf16-representable x

// every row m, column n computes c + sum_k a[k] * b[k] with the same 16 (a_k, b_k) pairs
__global__ void probe(const float *a, const float *b, float c, float *out)
{
    const int lane = threadIdx.x, kg = lane >> 5;
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = bf(a[8 * kg + i]); B[i] = bf(b[8 * kg + i]); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static float run(const float (&a)[16], const float (&b)[16], float c)
{
    float *da, *db, *dout, h;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return h;
}

int main()
{
    const float u = ldexpf(1.f, -23);  // ulp of 1.0
    struct T { const char *name; float c; float a[16]; float b[16]; double exact; };
    auto show = [&](const char *name, float c, const float (&a)[16], const float (&b)[16]) {
        double ex = c;
        for (int k = 0; k < 16; ++k) ex += (double)a[k] * (double)b[k];
        const float got = run(a, b, c);
        const float rne = (float)ex;
        printf("%-58s got %.10e  exact %.17e  RNE(exact) %.10e  diff/ulp(c) %+.3f  %s\n", name, got, ex, rne,
               (got - ex) / (double)(fabsf(c) > 0 ? ldexpf(1.f, ilogbf(c) - 23) : 1.f), got == rne ? "= RNE" : "");
    };
    float a[16], b[16];
    auto clr = [&]() { for (int k = 0; k < 16; ++k) { a[k] = 0.f; b[k] = 1.f; } };
    clr(); a[0] = 0.75f * u;                         show("1 + 0.75 ulp (RNE: 1+ulp, RZ: 1)", 1.f, a, b);
    clr(); a[0] = 0.5f * u;                          show("1 + 0.5 ulp (tie: RNE -> 1)", 1.f, a, b);
    clr(); a[0] = 0.5f * u; a[1] = 0.25f * u;        show("1 + 0.5 ulp + 0.25 ulp in two products (sum first: 1+ulp)", 1.f, a, b);
    clr(); a[0] = 0.375f * u; a[9] = 0.375f * u;     show("1 + 2 x 0.375 ulp, products in different K halves", 1.f, a, b);
    clr(); for (int k = 0; k < 16; ++k) a[k] = 0.0625f * u * 1.5f;  show("1 + 16 x 0.09375 ulp (sum 1.5 ulp)", 1.f, a, b);
    clr(); a[0] = -0.75f * u;                        show("1 - 0.75 ulp(1) (= 1.5 ulp below 1: exact 1 - 1.5 ulp')", 1.f, a, b);
    clr(); a[0] = 0.75f * u;                         show("-1 + 0.75 ulp (RZ -> -1+ulp? toward zero)", -1.f, a, b);
    clr(); a[0] = -0.75f * u;                        show("-1 - 0.75 ulp (RNE: -1-ulp, RZ: -1)", -1.f, a, b);
    clr(); a[0] = 1.f; a[1] = -1.f; a[2] = 0.75f * u; show("1 - 1 + 0.75u + c=1 (cancellation inside the sum)", 1.f, a, b);
    clr(); a[0] = 1.5f; b[0] = 1.5f; a[1] = 0.75f * u; show("c=1: 2.25 + 0.75 ulp(1)", 1.f, a, b);
    clr(); a[0] = ldexpf(1.f, 20); a[1] = 1.f + 0.0078125f; show("c=0: 2^20 + 1.0078125 (needs 28 bits)", 0.f, a, b);
    clr(); a[0] = ldexpf(1.f, 30); a[1] = 1.f; a[2] = -ldexpf(1.f, 30);  show("c=0: 2^30 + 1 - 2^30 (wide adder? exact = 1)", 0.f, a, b);
    clr(); a[0] = ldexpf(1.f, 40); a[1] = 1.f; a[2] = -ldexpf(1.f, 40);  show("c=0: 2^40 + 1 - 2^40", 0.f, a, b);
    clr(); a[0] = ldexpf(1.f, 24); a[1] = 1.f; a[8] = -ldexpf(1.f, 24);  show("c=0: 2^24 + 1 - 2^24 across K halves", 0.f, a, b);
    // how many bits below ulp(c) survive the alignment: c = 1, p1 = 0.5 ulp (a tie), p2 = 2^-j ulp; a kept bit breaks the tie upwards
    for (int j = 1; j <= 12; ++j) {
        clr(); a[0] = 0.5f * u; a[1] = ldexpf(u, -j);
        char nm[96]; snprintf(nm, sizeof nm, "tie + 2^-%d ulp, same K group", j);
        show(nm, 1.f, a, b);
    }
    // grouping: two products of 0.375 ulp at K positions (0, q)
    for (int q : {1, 2, 3, 4, 7, 8, 12, 15}) {
        clr(); a[0] = 0.375f * u; a[q] = 0.375f * u;
        char nm[96]; snprintf(nm, sizeof nm, "1 + 0.375 ulp at k=0 and k=%d", q);
        show(nm, 1.f, a, b);
    }
    // truncation or rounding of the aligned small terms?  c = 1, sixteen products of 2^-5 ulp * 1.9 each (sum 0.95 ulp -> 1 + ulp)
    clr(); for (int k = 0; k < 16; ++k) a[k] = 1.875f * ldexpf(u, -5);  show("1 + 16 x 1.875 * 2^-5 ulp (sum 0.9375 ulp)", 1.f, a, b);
    clr(); for (int k = 0; k < 8; ++k) a[k] = 1.875f * ldexpf(u, -4);   show("1 + 8 x 1.875 * 2^-4 ulp in K group 0 (sum 0.9375 ulp)", 1.f, a, b);
    clr(); for (int k = 0; k < 4; ++k) a[k] = 1.875f * ldexpf(u, -3);   show("1 + 4 x 1.875 * 2^-3 ulp, k = 0..3 (sum 0.9375 ulp)", 1.f, a, b);
    return 0;
}
