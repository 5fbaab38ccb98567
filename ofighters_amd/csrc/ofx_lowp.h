// ofx_lowp.h - operand packing and matrix instructions of the OPT-IN reduced-precision forward (OFX_OPT_POLICY_BF16:
// 1 = bf16 operands, 2 = fp16 operands; fp32 accumulation either way; never the default).  An operand quadruple is
// rounded to nearest even on its way into the matrix instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) and carried
// as 64 bits; the fp16 form keeps 11 significant bits instead of 8 and overflows past 65504 (activations behind a
// BatchNorm + ReLU and folded weights are far below that; an overflow would show as inf / NaN in the heat map).
#pragma once
#include <hip/hip_runtime.h>

typedef short lp_x4 __attribute__((ext_vector_type(4)));            // the 64-bit operand, whatever the format
typedef _Float16 lp_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 lp_h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lp_b16x2 __attribute__((ext_vector_type(2)));
typedef float lp_f32x2 __attribute__((ext_vector_type(2)));
typedef float lp_f32x4 __attribute__((ext_vector_type(4)));

template <int LP>
__device__ __forceinline__ lp_x4 lp_pk4(float a, float b, float c, float d) {
  static_assert(LP == 1 || LP == 2, "1 = bf16, 2 = fp16");
  if constexpr (LP == 1) {
    const lp_b16x2 lo = __builtin_convertvector((lp_f32x2){a, b}, lp_b16x2), hi = __builtin_convertvector((lp_f32x2){c, d}, lp_b16x2);
    const uint2 u = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(lp_x4, u);
  } else {
    const lp_h16x2 lo = __builtin_convertvector((lp_f32x2){a, b}, lp_h16x2), hi = __builtin_convertvector((lp_f32x2){c, d}, lp_h16x2);
    const uint2 u = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(lp_x4, u);
  }
}

// D[16][16] += A[16][16] B[16][16]: lane l holds A[l % 16][4 (l / 16) + i], B[4 (l / 16) + i][l % 16], i = 0..3
template <int LP>
__device__ __forceinline__ lp_f32x4 lp_mfma16(lp_x4 a, lp_x4 b, lp_f32x4 c) {
  if constexpr (LP == 1) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(lp_h16x4, a), __builtin_bit_cast(lp_h16x4, b), c, 0, 0, 0);
}

// 16 blocks of D[4][4] += A[4][4] B[4][4], the A block ABID broadcast to every block (CBSZ = 4)
template <int LP, int ABID>
__device__ __forceinline__ lp_f32x4 lp_mfma4_bcast(lp_x4 a, lp_x4 b, lp_f32x4 c) {
  if constexpr (LP == 1) return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 4, ABID, 0);
  else return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(lp_h16x4, a), __builtin_bit_cast(lp_h16x4, b), c, 4, ABID, 0);
}
