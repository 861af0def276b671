// The exact operand split of the "split" matrix path (spconv_split.cuh has the arithmetic): helpers shared by the sparse kernels
// (spconv.hip) and the pixel-GEMMs of the BEV neck (gemm2d.hip).
#pragma once

namespace toda {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// {bf16(x0) in the low half, bf16(x1) in the high half}, both truncated (upper 16 bits of the fp32 patterns)
__device__ __forceinline__ unsigned sp_pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
__device__ __forceinline__ float sp_trunc(float x) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u);
}
#ifndef SP_RNE
#define SP_RNE 0         // experiment: 1 = round-to-nearest planes (v_cvt_pk_bf16_f32) instead of truncated ones
#endif
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sp_pack_rne(float x0, float x1) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{x0, x1}), bf16x2v));
}
// two fp32 values -> their three packed bf16 planes
__device__ __forceinline__ void sp_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
#if SP_RNE
    h = sp_pack_rne(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xFFFF0000u);
    m = sp_pack_rne(r0, r1);
    const float l0 = r0 - __builtin_bit_cast(float, m << 16), l1 = r1 - __builtin_bit_cast(float, m & 0xFFFF0000u);
    l = sp_pack_rne(l0, l1);
#else
    h = sp_pack_hi(x0, x1);
    const float r0 = x0 - sp_trunc(x0), r1 = x1 - sp_trunc(x1);
    m = sp_pack_hi(r0, r1);
    l = sp_pack_hi(r0 - sp_trunc(r0), r1 - sp_trunc(r1));
#endif
}

}  // namespace toda
