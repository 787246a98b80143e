// MX-fp8 linears ("8-bit quantised -> fp8 MFMA").
//
// The reference quantises nn.Linear / nn.Embedding weights with MLX's affine 8-bit group format (group_size 64) in
// load_model's quantization branch (mlx_audio/tts/utils.py:241-260, predicate :349-369) and multiplies through
// mx.quantized_matmul.  Here the same layer set runs on CDNA4's block-scaled matrix instruction
//   v_mfma_scale_f32_32x32x64_f8f6f4   (A, B: OCP e4m3; one E8M0 scale per 32 consecutive k of a row; fp32 accumulate)
// with the (dequantised) weights re-quantised ONCE at kk_finalize to e4m3 with one power-of-two scale per `group` (64)
// input channels -- the hardware's two 32-blocks of a group carry the same scale byte -- and the bf16 activations
// quantised per (row, 32-block) by a small pre-pass.  Both operands are stored in MFMA FRAGMENT ORDER so that a wave's
// operand load is one contiguous, fully coalesced 1-KiB `global_load_dwordx4` and nothing goes through LDS:
//   lane l of a wave (r = l & 31, h = l >> 5) holds row 32*blk + r of its operand; its first four operand registers belong to
//   the k-step's FIRST 32-block (k = 64*ks + 16*h + [0, 16)), its last four to the SECOND (k = 64*ks + 32 + 16*h + [0, 16)); the scale
//   byte of lane l governs block h of row r, i.e. registers 4h .. 4h+3 of BOTH lanes r and r + 32 (measured with
//   tools/probe_mxscale.hip: the instruction pairs scale lane r with registers 0-3 and scale lane r + 32 with registers 4-7)
//   q[((blk*KS + ks)*2 + half)*64 + l]  (16 bytes: k = 64*ks + 32*half + 16*h + [0, 16))
//   s[(blk*KS + ks)*64 + l]             (1 byte, E8M0: value 2^(byte - 127), block h of row r)
// Quantisation rule of a block (the same for weights on the host and activations on the device, restated in
// oracle/mxfp8_oracle.py):  e = floor(log2(amax)) - 8;  if amax * 2^-e > 448: e += 1;  e clamped to [-127, 127];
// q = rne_e4m3(x * 2^-e);  amax == 0 -> e = 0.
#include <math.h>
#include <string.h>

#include "kk_kernels.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ---- activation pre-pass: bf16 rows -> e4m3 fragments + scale bytes.  One wave per (32-row block, k-step).
__global__ __launch_bounds__(256) void mxfp8_quant_rows_kernel(const bf16_t* x, int ldx, int M, int K, uint4* aq, unsigned char* as) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int KS = K >> 6;
  const int MB = (M + 31) >> 5;
  const long long frag = (long long)blockIdx.x * 4 + wv;
  if (frag >= (long long)MB * KS) return;
  const int mb = (int)(frag / KS), ks = (int)(frag - (long long)mb * KS);
  const int r = lane & 31, h = lane >> 5;
  const int m = mb * 32 + r;
  uint4 raw[4];  // raw[0..1]: k = 16h + [0, 16) of block 0; raw[2..3]: the same columns of block 1
  if (m < M) {
    const uint4* p = (const uint4*)(x + (long long)m * ldx + ks * 64 + 16 * h);
    raw[0] = p[0]; raw[1] = p[1]; raw[2] = p[4]; raw[3] = p[5];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) raw[i] = make_uint4(0, 0, 0, 0);
  }
  float v[32];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[8 * i + 2 * j] = __uint_as_float(w[j] << 16);
      v[8 * i + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
  }
  int e[2];
  float inv[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[16 * blk + i]));
    amax = fmaxf(amax, __shfl_xor(amax, 32));  // the other half of the 32-block lives in lane r + 32 (or r)
    int ee = 0;
    if (amax > 0.f) {
      ee = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
      if (ldexpf(amax, -ee) > 448.f) ee += 1;
      ee = max(-127, min(127, ee));
    }
    e[blk] = ee;
    inv[blk] = ldexpf(1.0f, -ee);
  }
  unsigned pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float sc = inv[i >> 2];
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i] * sc, v[4 * i + 1] * sc, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 2] * sc, v[4 * i + 3] * sc, w, true);
    pk[i] = (unsigned)w;
  }
  aq[(frag * 2 + 0) * 64 + lane] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  aq[(frag * 2 + 1) * 64 + lane] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
  as[frag * 64 + lane] = (unsigned char)((h ? e[1] : e[0]) + 127);
}

// ---- the product: out[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]), zero for rows past the utterance's length.
// Workgroup = 4 waves as 2 x 2, each wave 64 rows x 64 columns = 2 x 2 accumulators of 32 x 32; per k-step (K = 64) a wave
// loads 4 operand fragments (8 x 1 KiB) + 4 scale bytes per lane and issues 4 MFMAs; the next k-step's operands are in
// flight while the current MFMAs run.
struct Frag {
  uint4 lo, hi;
  int s;
};
__device__ __forceinline__ Frag load_frag(const uint4* q, const unsigned char* s, long long frag, int lane) {
  Frag f;
  f.lo = q[(frag * 2 + 0) * 64 + lane];
  f.hi = q[(frag * 2 + 1) * 64 + lane];
  f.s = s[frag * 64 + lane];
  return f;
}
__device__ __forceinline__ v8i as_v8i(const Frag& f) {
  v8i r;
  r[0] = (int)f.lo.x; r[1] = (int)f.lo.y; r[2] = (int)f.lo.z; r[3] = (int)f.lo.w;
  r[4] = (int)f.hi.x; r[5] = (int)f.hi.y; r[6] = (int)f.hi.z; r[7] = (int)f.hi.w;
  return r;
}

__global__ __launch_bounds__(256) void linear_mxfp8_kernel(KKFp8Args a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int KS = a.K >> 6;
  const int MB = (a.M + 31) >> 5, NB = a.N >> 5;
  // XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive ids must NOT be the
  // column blocks of one row block -- every XCD would then fetch every activation fragment.  The linear id is re-dealt so that an XCD
  // walks whole row blocks (all column blocks of a row block on one L2) in id order.
  const int nbx = gridDim.x, nby = gridDim.y, total = nbx * nby;
  int lid = blockIdx.y * nbx + blockIdx.x;
  {
    const int per = total / 8, rem = total - per * 8;  // the first `rem` XCDs own one tile more
    const int xcd = lid & 7, idx = lid >> 3;
    lid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int by = lid / nbx, bx = lid - by * nbx;
  const int mb0 = by * 4 + wm * 2, nb0 = bx * 4 + wn * 2;
  if (mb0 >= MB || nb0 >= NB) return;  // whole wave; no barriers in this kernel
  long long fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    fa[i] = (long long)min(mb0 + i, MB - 1) * KS;  // clamped: the duplicate tile is never stored
    fb[i] = (long long)min(nb0 + i, NB - 1) * KS;
  }
  v16f acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // operands of k-steps ks + 1 and ks + 2 are in flight while the MFMAs of ks run (an L2 round trip is ~5x one k-step of MFMAs)
  Frag A0[2], B0[2], A1[2], B1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    A0[i] = load_frag(a.aq, a.as, fa[i], lane);
    B0[i] = load_frag(a.wq, a.ws, fb[i], lane);
    A1[i] = load_frag(a.aq, a.as, fa[i] + (KS > 1 ? 1 : 0), lane);
    B1[i] = load_frag(a.wq, a.ws, fb[i] + (KS > 1 ? 1 : 0), lane);
  }
  for (int ks = 0; ks < KS; ++ks) {
    Frag A2[2], B2[2];
    const int kn = ks + 2 < KS ? ks + 2 : KS - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      A2[i] = load_frag(a.aq, a.as, fa[i] + kn, lane);
      B2[i] = load_frag(a.wq, a.ws, fb[i] + kn, lane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(as_v8i(A0[i]), as_v8i(B0[j]), acc[i][j], 0 /*A e4m3*/, 0 /*B e4m3*/, 0,
                                                                    A0[i].s, 0, B0[j].s);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      A0[i] = A1[i]; B0[i] = B1[i];
      A1[i] = A2[i]; B1[i] = B2[i];
    }
  }
  // epilogue: C/D layout of the 32x32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  // A 32-row block crosses at most one utterance boundary when rows_per_item >= 32: one division and two length loads per row block
  // (a division and a length load per ROW, 64 of each per lane, was a large part of this kernel's time).
  const int rpi = a.rows_per_item;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (mb0 + i >= MB) continue;
    const int m0 = (mb0 + i) * 32;
    const int b0 = m0 / rpi, t0 = m0 - b0 * rpi;
    const int nitems = (a.M + rpi - 1) / rpi;
    const int len0 = kk_len(a.lout, b0), len1 = kk_len(a.lout, b0 + 1 < nitems ? b0 + 1 : b0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (nb0 + j >= NB) continue;
      const int n = (nb0 + j) * 32 + (lane & 31);
      const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = m0 + dr;
        if (m >= a.M) continue;
        bool valid;
        if (rpi >= 32) {
          const int t = t0 + dr;
          valid = t < rpi ? t < len0 : t - rpi < len1;
        } else {
          const int bb = m / rpi;
          valid = m - bb * rpi < kk_len(a.lout, bb);
        }
        float v = 0.f;
        if (valid) {
          v = acc[i][j][r] + bias;
          if (a.act == KK_ACT_GELU) v = gelu_erf(v);
        }
        a.out[(long long)m * a.ldo + n] = (bf16_t)v;
      }
    }
  }
}

// ---- host: OCP e4m3 (e4m3fn) round-to-nearest-even, |x| <= 448 by construction (larger magnitudes saturate to 448)
unsigned char f32_to_e4m3(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const unsigned char sign = (unsigned char)((u >> 31) << 7);
  const float a = fabsf(f);
  if (a != a) return sign | 0x7f;
  if (a > 448.f) return sign | 0x7e;
  if (a == 0.f) return sign;
  int ex;
  (void)frexpf(a, &ex);  // a = m * 2^ex, m in [0.5, 1)
  int E = ex - 1;
  if (E < -6) E = -6;
  const float quantum = ldexpf(1.0f, E - 3);
  int q = (int)nearbyintf(a / quantum);  // ties to even (default rounding mode); a / quantum is exact (power of two)
  if (q == 0) return sign;
  if (E == -6 && q < 8) return sign | (unsigned char)q;  // subnormal: exponent field 0
  if (q == 16) {
    q = 8;
    E += 1;
  }
  return sign | (unsigned char)(((E + 7) << 3) | (q - 8));
}

}  // namespace

size_t kk_mxfp8_q_bytes(int rows, int K) { return (size_t)((rows + 31) / 32) * (K / 64) * 2 * 64 * 16; }
size_t kk_mxfp8_s_bytes(int rows, int K) { return (size_t)((rows + 31) / 32) * (K / 64) * 64; }

int kk_mxfp8_pack_weight_host(const float* w, int N, int K, int group, unsigned char* q, unsigned char* s) {
  if (N <= 0 || K <= 0 || (K & 63) || (N & 31)) return kk_fail("mxfp8 pack: K must be a multiple of 64 and N of 32");
  if (group < 32 || (group & 31) || K % group) return kk_fail("mxfp8 pack: group must be a multiple of 32 that divides K");
  const int KS = K / 64, NB = N / 32;
  for (int n = 0; n < N; ++n) {
    const float* row = w + (size_t)n * K;
    for (int g0 = 0; g0 < K; g0 += group) {
      float amax = 0.f;
      for (int k = g0; k < g0 + group; ++k) amax = fmaxf(amax, fabsf(row[k]));
      int e = 0;
      if (amax > 0.f && amax == amax) {
        uint32_t u;
        memcpy(&u, &amax, 4);
        e = (int)((u >> 23) & 0xff) - 127 - 8;
        if (ldexpf(amax, -e) > 448.f) e += 1;
        if (e < -127) e = -127;
        if (e > 127) e = 127;
      }
      const float inv = ldexpf(1.0f, -e);
      for (int k = g0; k < g0 + group; ++k) {
        const int nb = n >> 5, r = n & 31, ks = k >> 6, half = (k >> 5) & 1, h = (k >> 4) & 1, byte = k & 15;
        const size_t frag = (size_t)nb * KS + ks;
        q[((frag * 2 + half) * 64 + r + 32 * h) * 16 + byte] = f32_to_e4m3(row[k] * inv);
        if ((k & 31) == 0) s[frag * 64 + r + 32 * half] = (unsigned char)(e + 127);
      }
    }
  }
  (void)NB;
  return 0;
}

bool kk_mxfp8_eligible(int K, int N) { return K > 0 && N > 0 && (K & 63) == 0 && (N & 63) == 0; }

int kk_launch_mxfp8_quant_rows(const void* x, int ldx, int M, int K, void* aq, void* as, hipStream_t st) {
  if (M <= 0) return 0;
  if ((K & 63) || (ldx & 7) || (((uintptr_t)x | (uintptr_t)aq) & 15)) return kk_fail("mxfp8 quant: K % 64, ldx % 8 and 16-byte alignment required");
  const long long frags = (long long)((M + 31) / 32) * (K / 64);
  hipLaunchKernelGGL(mxfp8_quant_rows_kernel, dim3((unsigned)((frags + 3) / 4)), dim3(256), 0, st, (const bf16_t*)x, ldx, M, K, (uint4*)aq,
                     (unsigned char*)as);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_linear_mxfp8(const KKFp8Args& a, hipStream_t st) {
  if (a.M <= 0) return 0;
  if (!kk_mxfp8_eligible(a.K, a.N) || a.rows_per_item <= 0) return kk_fail("mxfp8 linear: K % 64 == 0 and N % 64 == 0 required");
  if (a.act != KK_ACT_NONE && a.act != KK_ACT_GELU) return kk_fail("mxfp8 linear: activation must be none or exact GELU");
  const int MB = (a.M + 31) / 32, NB = a.N / 32;
  hipLaunchKernelGGL(linear_mxfp8_kernel, dim3(kk_cdiv(NB, 4), kk_cdiv(MB, 4)), dim3(256), 0, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}
