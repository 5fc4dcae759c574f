// Does MODE.FP16_OVFL (hwreg(HW_REG_MODE) bit 23) make v_cvt_pk_f16_f32 saturate at +-65504 on gfx950?
// (the f16 decoder tier clamps every packed pair with v_pk_min_f16 + v_pk_max_f16: two instructions per conversion)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

__global__ void k(const float *in, unsigned *out, int n, int set)
{
    if (set) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = threadIdx.x;
    if (i < n) {
        f32x2 v = {in[i], -in[i]};
        out[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2));
    }
}

int main()
{
    const float h[8] = {1.0f, 65504.f, 65519.f, 65520.f, 70000.f, 1e6f, INFINITY, NAN};
    float *d; unsigned *o, r[8];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 8, set);
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", set);
        for (int i = 0; i < 8; ++i) printf("  %g -> %04x/%04x", h[i], r[i] & 0xffff, r[i] >> 16);
        printf("\n");
    }
    return 0;
}
