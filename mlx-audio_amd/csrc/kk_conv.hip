// Generic tap-loop convolution / transposed convolution / linear on frames-major tensors.
//
//   KK_CONV :  out[b][q][co]          = sum_t sum_ci w[t][ci][co] * x[b][(q*stride - pad + t*dil) >> in_shift][ci]
//   KK_CONVT:  out[b][r + stride*q][co] = sum_m sum_ci w[k0 + m*stride][ci][co] * x[b][q + (r+pad-k0)/stride - m][ci]
//              (polyphase form of the transposed conv: phase r = blockIdx.z % stride, k0 = (r+pad) % stride)
//   linear  :  KK_CONV with Kw = 1.
//
// Replaces mx.conv1d / mx.conv_transpose1d / nn.Linear at every call site of the reference's
// hot path (istftnet.py:137-157, modules.py nn.Linear).  fp32 VALU implicit GEMM, 128x64x16
// tiles, 8x4 register micro-tile.  This is the exact-arithmetic path used for parity; the bf16
// MFMA kernel (kk_conv_mfma.hip) takes over the large stride-1 convolutions in bf16 mode.
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 16;

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ELU: elu(x, 1) on the input while it is staged (Mimi SEANet) -- a template parameter: as a run-time select the exp() was evaluated
// for every staged element of every conv (+16 % on the fp32 Kokoro forward)
template <typename TI, typename TO, bool ELU>
__global__ __launch_bounds__(256) void conv_generic_kernel(KKConvArgs a) {
  __shared__ __attribute__((aligned(16))) float As[BK][BM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[BK][BN + 4];
  const int tid = threadIdx.x;
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  const int b = blockIdx.z / nphase, phase = blockIdx.z - b * nphase;
  const int q0 = blockIdx.x * BM, co0 = blockIdx.y * BN;
  const int Lin = kk_len(a.lin, b), Lout = kk_len(a.lout, b);
  int ntaps, k0 = 0, ibase = 0;
  if (a.mode == KK_CONV) {
    ntaps = a.Kw;
  } else {
    k0 = (phase + a.pad) % a.stride;
    ntaps = (a.Kw - k0 + a.stride - 1) / a.stride;
    ibase = (phase + a.pad - k0) / a.stride;
  }
  const TI* xb = (const TI*)a.x + (long long)b * a.xbs;
  const int am = tid >> 1, ak0 = (tid & 1) * 8;  // A loader: row am, channels ak0..ak0+7 of the K slab
  const int bk = tid >> 4, bn = (tid & 15) * 4;  // B loader: K row bk, 4 output channels
  const int tx = tid & 15, ty = tid >> 4;        // compute: rows ty*8.., cols tx*4..
  float acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  // first output row of this tile: anything to do at all?
  const int op_first = a.mode == KK_CONV ? q0 : phase + a.stride * q0;
  const bool tile_live = op_first < Lout;

  if (tile_live) {
    for (int t = 0; t < ntaps; ++t) {
      int r, widx;
      const int q = q0 + am;
      if (a.mode == KK_CONV) {
        r = q * a.stride - a.pad + t * a.dil;
        if (a.in_shift) r >>= a.in_shift;
        widx = t;
      } else {
        r = q + ibase - t;
        widx = k0 + t * a.stride;
      }
      const bool rv = (q < a.Q) && r >= 0 && r < Lin;
      const TI* xr = xb + (long long)(rv ? r : 0) * a.ldx;
      const float* wt = a.w + (long long)widx * a.Cin * a.ldw;
      for (int c0 = 0; c0 < a.Cin; c0 += BK) {
        float av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = c0 + ak0 + i;
          float v = (rv && c < a.Cin) ? kk_ld(xr + c) : 0.f;
          av[i] = ELU ? (v > 0.f ? v : expf(v) - 1.0f)  // nn.elu: where(x > 0, x, exp(x) - 1)
                      : (v > 0.f ? v : v * a.in_slope);
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + bk < a.Cin) bv = *(const float4*)(wt + (long long)(c0 + bk) * a.ldw + co0 + bn);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) As[ak0 + i][am] = av[i];
        *(float4*)&Bs[bk][bn] = bv;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK; ++k) {
          const float4 a0 = *(const float4*)&As[k][ty * 8];
          const float4 a1 = *(const float4*)&As[k][ty * 8 + 4];
          const float4 b4 = *(const float4*)&Bs[k][tx * 4];
          const float ar[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const float br[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(ar[i], br[j], acc[i][j]);
        }
      }
    }
  }

  TO* ob = (TO*)a.out + (long long)b * a.obs;
  const TO* rb = a.res ? (const TO*)a.res + (long long)b * a.rbs : nullptr;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = q0 + ty * 8 + i;
    if (q >= a.Q) continue;
    const int op = a.mode == KK_CONV ? q : phase + a.stride * q;
    if (op >= a.Lo_rows) continue;
    const bool live = op < Lout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + tx * 4 + j;
      if (co >= a.Cout) continue;
      float v = 0.f;
      if (live) {
        v = acc[i][j] + (a.bias ? a.bias[co] : 0.f);
        if (a.act == KK_ACT_LRELU) v = v > 0.f ? v : v * a.act_slope;
        else if (a.act == KK_ACT_GELU) v = gelu_exact(v);
        else if (a.act == KK_ACT_GELU_TANH) v = 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v))));  // nn.gelu_approx
        if (rb) v += kk_ld(rb + (long long)op * a.ldr + co);
        v *= a.scale;
        if (a.accumulate) v += kk_ld(ob + (long long)op * a.ldo + co);
      }
      kk_st(ob + (long long)op * a.ldo + co, v);
    }
  }
}

}  // namespace

int kk_launch_conv_generic(const KKConvArgs& a, int B, int in_dtype, int out_dtype, hipStream_t st) {
  if (a.ldw % 4 != 0 || a.ldw < kk_cdiv(a.Cout, BN) * BN) return kk_fail("conv_generic: ldw must be a multiple of 64 covering Cout");
  if (a.Q <= 0 || B <= 0) return 0;
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  dim3 grid(kk_cdiv(a.Q, BM), kk_cdiv(a.Cout, BN), B * nphase);
  const bool elu = a.in_act == KK_ACT_ELU;
#define KK_GO(TI, TO)                                                                                      \
  do {                                                                                                     \
    if (elu) hipLaunchKernelGGL((conv_generic_kernel<TI, TO, true>), grid, dim3(256), 0, st, a);           \
    else hipLaunchKernelGGL((conv_generic_kernel<TI, TO, false>), grid, dim3(256), 0, st, a);              \
  } while (0)
  if (in_dtype == KK_F32 && out_dtype == KK_F32) KK_GO(float, float);
  else if (in_dtype == KK_BF16 && out_dtype == KK_BF16) KK_GO(bf16_t, bf16_t);
  else if (in_dtype == KK_BF16 && out_dtype == KK_F32) KK_GO(bf16_t, float);
  else if (in_dtype == KK_F32 && out_dtype == KK_BF16) KK_GO(float, bf16_t);
  else
    return kk_fail("conv_generic: bad dtype");
#undef KK_GO
  KK_CHECK_LAUNCH();
  return 0;
}
