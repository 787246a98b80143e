// Helpers shared by the MFMA convolution kernels (variants 2, 4, 5).
#pragma once
#include <hip/hip_runtime.h>

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
}  // namespace
