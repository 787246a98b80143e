// Helpers shared by the MFMA convolution kernels (variants 2, 4, 5).
#pragma once
#include <hip/hip_runtime.h>

// Matrix-instruction shape of variants 4 / 5: v_mfma_f32_16x16x32_bf16 (default) or, with -DKK_MFMA32, the round-2 v_mfma_f32_32x32x16_bf16.
// KK_XLD = elements per LDS row of the X slab: 160 B keeps the 16-row fragments' ds_read_b128 conflict-free (see kk_conv_mfma4.hip), 144 B the 32-row ones.
#if !defined(KK_MFMA32) && !defined(KK_EXP_MFMA16)
#define KK_MFMA16 1
#endif
#ifdef KK_MFMA16
#define KK_XLD 80
#else
#define KK_XLD 72
#endif

namespace {
// Two floats WITHOUT the packed-f32 instructions (v_pk_fma_f32 ...): beside another wave's MFMAs on the same SIMD the packed forms run at
// about half rate (variant 5's service waves showed it first; 2-6 % per fused launch of variant 4).  The files that include this are built
// with -fno-slp-vectorize for the same reason (build.py FILE_FLAGS).
struct v2f {
  float x, y;
};
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2f& operator+=(v2f& a, v2f b) { a.x += b.x; a.y += b.y; return a; }
__device__ __forceinline__ v2f& operator*=(v2f& a, v2f b) { a.x *= b.x; a.y *= b.y; return a; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return v2f{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
// nn.gelu (exact erf form, modules.py / transformer feed-forward)
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// KK_MFMA32(acc, a, b, ks): one v_mfma_f32_32x32x16_bf16.  -DKK_EXP_MFMA16 (build.py --exp16; TIMING ONLY, WRONG RESULTS): the same
// operands through TWO v_mfma_f32_16x16x32_bf16 on quarter accumulators -- equal cycles, FLOPs, register and LDS traffic -- to measure what the
// other MFMA shape does to the clock the chip holds (MI355X_MICROARCH.md, DVFS give-back item 7) before re-laying the kernels out for it.
typedef float kk_f32x4 __attribute__((ext_vector_type(4)));
typedef float kk_f32x8 __attribute__((ext_vector_type(8)));
typedef float kk_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 kk_bf16x8 __attribute__((ext_vector_type(8)));
#ifdef KK_EXP_MFMA16
#define KK_MFMA_PER 2
template <int KS>
__device__ __forceinline__ kk_f32x16 kk_mfma32(kk_bf16x8 a, kk_bf16x8 b, kk_f32x16 c) {
  kk_f32x4 p0 = __builtin_shufflevector(c, c, 0, 1, 2, 3), p1 = __builtin_shufflevector(c, c, 4, 5, 6, 7);
  kk_f32x4 p2 = __builtin_shufflevector(c, c, 8, 9, 10, 11), p3 = __builtin_shufflevector(c, c, 12, 13, 14, 15);
  if (KS & 1) {
    p2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, p2, 0, 0, 0);
    p3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, p3, 0, 0, 0);
  } else {
    p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, p0, 0, 0, 0);
    p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, p1, 0, 0, 0);
  }
  const kk_f32x8 lo = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(p2, p3, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}
#else
#define KK_MFMA_PER 1
template <int KS>
__device__ __forceinline__ kk_f32x16 kk_mfma32(kk_bf16x8 a, kk_bf16x8 b, kk_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif
}  // namespace
