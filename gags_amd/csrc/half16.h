// The 16-bit operand type of the decoder kernels (decoder.hip, decoder_fused.hip, decoder_scale.hip).  Each of those files is
// compiled TWICE from the same source:
//   default        bfloat16: the "bf16" mode (8 significand bits; entry points gags_decoder_*, gags_scale_decoder_*)
//   -DGAGS_H16     IEEE half: the "f16" tier (11 significand bits -- exactly the TF32 significand the reference's convolutions
//                  run with under PyTorch's default cudnn.allow_tf32 -- fp32 accumulation; entry points ..._h16)
// v_mfma_f32_32x32x16_f16 and _bf16 run at the same rate, the kernels' bit tricks on packed pairs (ReLU as v_pk_max_i16 with
// 0, decisions as v_pk_min_u16 with 1, masks applied with v_pk_mul_lo_u16) only need "negative <=> sign bit" and hold for
// both.  What half does NOT have is fp32's exponent range: conversions saturate at +-65504 instead of producing inf, and the
// gradients travel multiplied by a power of two chosen per call (gags_amd/decoders.py: f16 tier).
#pragma once
#include <hip/hip_runtime.h>

namespace gags_h16 {
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

#ifdef GAGS_H16
#define GAGS_DEC(name) name##_h16
typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8v __attribute__((ext_vector_type(8)));
// two floats -> packed pair, round to nearest even (v_cvt_pk_f16_f32), saturated to the largest finite half
__device__ __forceinline__ unsigned h16_pack(float lo, float hi)
{
    const f32x2v v = {lo, hi};
    unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2v));
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(u) : "v"(u), "v"(0x7bff7bffu));
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(u) : "v"(u), "v"(0xfbfffbffu));
    return u;
}
// ... of values known to be >= 0 afterwards (a ReLU follows): only the upper clamp
__device__ __forceinline__ unsigned h16_pack_raw(float lo, float hi)
{
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2v));
}
__device__ __forceinline__ unsigned h16_clamp_hi(unsigned u)
{
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(u) : "v"(u), "v"(0x7bff7bffu));
    return u;
}
// The same saturation from the hardware (round 6): with MODE.FP16_OVFL set, v_cvt_pk_f16_f32 itself returns +-65504 for every
// finite value beyond half's range (tools/micro/f16_ovfl.hip on gfx950: 65520, 7e4, 1e6 -> 7bff; an fp32 infinity stays an
// infinity and a NaN a NaN, where the explicit clamps made 65504 of the former -- neither can come out of fp32 sums of products
// of halves).  A kernel that calls h16_saturate_mode() FIRST packs with h16_pack_sat: the two clamps per conversion were
// ~470 VALU instructions per 64-pixel tile of the two fused CNN_decoder kernels, 0.1 ms each per 1080p iteration.
__device__ __forceinline__ void h16_saturate_mode() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
__device__ __forceinline__ unsigned h16_pack_sat(float lo, float hi) { return h16_pack_raw(lo, hi); }
__device__ __forceinline__ float h16_lo(unsigned u) { return (float)__builtin_bit_cast(h16x2v, u)[0]; }
__device__ __forceinline__ float h16_hi(unsigned u) { return (float)__builtin_bit_cast(h16x2v, u)[1]; }
__device__ __forceinline__ f32x16v h16_mfma(s16x8v a, s16x8v b, f32x16v c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8v, a), __builtin_bit_cast(h16x8v, b), c, 0, 0, 0);
}
#else
#define GAGS_DEC(name) name
typedef __bf16 h16x2v __attribute__((ext_vector_type(2)));
// two floats -> packed pair, round to nearest even, in one instruction (v_cvt_pk_bf16_f32, new on gfx950)
__device__ __forceinline__ unsigned h16_pack(float lo, float hi)
{
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2v));
}
__device__ __forceinline__ unsigned h16_pack_raw(float lo, float hi) { return h16_pack(lo, hi); }
__device__ __forceinline__ unsigned h16_clamp_hi(unsigned u) { return u; }
__device__ __forceinline__ void h16_saturate_mode() {}
__device__ __forceinline__ unsigned h16_pack_sat(float lo, float hi) { return h16_pack(lo, hi); }
__device__ __forceinline__ float h16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float h16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ f32x16v h16_mfma(s16x8v a, s16x8v b, f32x16v c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif
__device__ __forceinline__ unsigned short h16_from(float f) { return (unsigned short)(h16_pack(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float h16_to(unsigned short h) { return h16_lo((unsigned)h); }
}  // namespace gags_h16
