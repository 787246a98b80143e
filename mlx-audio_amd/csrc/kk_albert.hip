// Albert pieces that are not plain linears:
//   * AlbertEmbeddings (modules.py:451-465): word + position + token_type(0) -> LayerNorm(eps 1e-12)
//   * AlbertSelfAttention core (modules.py:497-512): softmax(Q K^T / sqrt(64) + mask) V per head.
// Keys past the utterance's length do not exist in the reference's batch-1 call; here they are
// simply not visited.  One workgroup per (head, utterance, 32-query tile); see attention_kernel.
#include <stdlib.h>

#include "kk_common.h"
#include "kk_kernels.h"

namespace {

union U16 {
  uint4 u;
  bf16_t h[8];
};

template <typename T>
__global__ __launch_bounds__(256) void albert_embed_kernel(KKEmbedArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wv, b = blockIdx.y;
  if (t >= a.Tmax) return;
  const int L = kk_len(a.len, b);
  T* orow = (T*)a.out + (long long)b * a.obs + (long long)t * a.ldo;
  if (t >= L) {
    for (int c = lane; c < a.E; c += 64) kk_st(orow + c, 0.f);
    return;
  }
  const int id = a.ids[(long long)b * a.Tmax + t];
  constexpr int MAXV = 8;  // E <= 512
  float v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    float x = 0.f;
    if (c < a.E) x = (a.word[(long long)id * a.E + c] + a.pos[(long long)t * a.E + c]) + a.type[c];
    v[i] = x;
    s += x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)a.E;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < a.E) {
      const float d = v[i] - mean;
      ss = __builtin_fmaf(d, d, ss);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = 1.0f / sqrtf(ss / (float)a.E + a.eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < a.E) kk_st(orow + c, (v[i] - mean) * rstd * a.ln_w[c] + a.ln_b[c]);
  }
}

constexpr int HD = 64, QT = 32, KP = 8, KC = 128, KLD = 68;  // 32 queries x 8 key partitions per workgroup, 128-key chunks
constexpr int ATT_LDS = 2 * KC * KLD * 4;

// One workgroup per (head, utterance, 32-query tile).  Thread (qi, part) owns query qi and the keys k = part (mod 8): its own
// online softmax over ~L/8 keys, then the 8 partials of a query (adjacent lanes) are merged with shuffles.  K / V chunks sit in
// LDS as fp32 rows of 68 floats: the 8 partitions of a wave read 8 consecutive rows (16 B each, bank offset 4 per row ->
// conflict-free) and the 8 queries sharing a partition read the same address (broadcast).
template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(KKAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float att_lds[];
  float* Ks = att_lds;
  float* Vs = att_lds + KC * KLD;
  const int tid = threadIdx.x, part = tid & (KP - 1), ql = tid >> 3;
  const int h = blockIdx.x, b = blockIdx.y, qi = blockIdx.z * QT + ql;
  const int L = kk_len(a.len, b);
  const T* base = (const T*)a.qkv + (long long)b * a.bs;
  T* ob = (T*)a.out + (long long)b * a.obs;
  const int qoff = h * HD, koff = a.hs + h * HD, voff = 2 * a.hs + h * HD;
  const bool qv = qi < L;
  if (blockIdx.z * QT >= L) {  // whole tile is past the utterance: rows are zero (uniform over the workgroup)
    if (qi < a.Tmax)
#pragma unroll
      for (int d = 0; d < 8; ++d) kk_st(ob + (long long)qi * a.ldo + qoff + part * 8 + d, 0.f);
    return;
  }
  float q[HD], o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    q[d] = qv ? kk_ld(base + (long long)qi * a.ld + qoff + d) * a.scale : 0.f;
    o[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < L; k0 += KC) {
    if (k0 > 0) __syncthreads();
    for (int e = tid; e < KC * (HD / 4); e += 256) {
      const int kr = e >> 4, d = (e & 15) * 4;
      const int ki = k0 + kr;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (ki < L) {
        const T* kp = base + (long long)ki * a.ld + koff + d;
        const T* vp = base + (long long)ki * a.ld + voff + d;
        kv = make_float4(kk_ld(kp), kk_ld(kp + 1), kk_ld(kp + 2), kk_ld(kp + 3));
        vv = make_float4(kk_ld(vp), kk_ld(vp + 1), kk_ld(vp + 2), kk_ld(vp + 3));
      }
      *(float4*)(Ks + kr * KLD + d) = kv;
      *(float4*)(Vs + kr * KLD + d) = vv;
    }
    __syncthreads();
    const int kn = min(KC, L - k0);
    for (int kr = part; kr < kn; kr += KP) {
      const float* krow = Ks + kr * KLD;
      const float* vrow = Vs + kr * KLD;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 k4 = *(const float4*)(krow + d);
        s0 = __builtin_fmaf(q[d], k4.x, s0);
        s1 = __builtin_fmaf(q[d + 1], k4.y, s1);
        s2 = __builtin_fmaf(q[d + 2], k4.z, s2);
        s3 = __builtin_fmaf(q[d + 3], k4.w, s3);
      }
      const float sc = (s0 + s1) + (s2 + s3);
      const float mn = fmaxf(m, sc);
      const float corr = expf(m - mn);
      const float p = expf(sc - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 v4 = *(const float4*)(vrow + d);
        o[d] = __builtin_fmaf(p, v4.x, o[d] * corr);
        o[d + 1] = __builtin_fmaf(p, v4.y, o[d + 1] * corr);
        o[d + 2] = __builtin_fmaf(p, v4.z, o[d + 2] * corr);
        o[d + 3] = __builtin_fmaf(p, v4.w, o[d + 3] * corr);
      }
      m = mn;
    }
  }
  // merge the 8 partials of this query (lanes part = 0..7 are adjacent)
  float mg = m;
#pragma unroll
  for (int x = 1; x < KP; x <<= 1) mg = fmaxf(mg, __shfl_xor(mg, x));
  const float w = (m == -INFINITY) ? 0.f : expf(m - mg);  // a partition that saw no key contributes nothing
  l *= w;
#pragma unroll
  for (int x = 1; x < KP; x <<= 1) l += __shfl_xor(l, x);
  const float inv = (qv && l > 0.f) ? 1.0f / l : 0.f;
  float mine[8];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    float v = o[d] * w;
#pragma unroll
    for (int x = 1; x < KP; x <<= 1) v += __shfl_xor(v, x);
    if ((d >> 3) == part) mine[d & 7] = v;  // lane `part` keeps dims 8*part .. 8*part+7
  }
  if (qi < a.Tmax) {
#pragma unroll
    for (int d = 0; d < 8; ++d) kk_st(ob + (long long)qi * a.ldo + qoff + part * 8 + d, qv ? mine[d] * inv : 0.f);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// bf16 mode: flash-style attention on the matrix cores, ONE WAVE per (head, utterance, 32-query tile), no barriers
// between waves, 4.6 KB LDS.  Per 32-key tile:
//   S^T[key][q]  = K[key][:] . Q[q][:]          4 x mfma_32x32x16 (A = K rows straight from global, B = Q rows, loaded once)
//   online softmax per query: in the S^T accumulator layout a lane owns ONE query column (its 16 keys in registers, the other
//   16 in lane ^ 32), so max / sum are register reductions plus one shuffle and the rescale factor is per lane
//   O^T[d][q]   += V^T[d][key] . P^T[key][q]    4 x mfma (2 d-tiles x 2 k-steps).  P^T is the accumulator itself: MFMA k-slot
//   (half h, j) of k-step s is DEFINED as the key the lane already holds in register 8s + j, i.e. key = (j&3) + 4h + 8(j>>2) +
//   16s -- a fixed permutation of the 32 keys, applied to V^T when its tile is transposed through LDS (key bits 2 and 3 swap).
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef float af32x16 __attribute__((ext_vector_type(16)));
constexpr int VT_LD = 40;  // bf16 elements per V^T row in LDS (32 keys + 8 pad: 80-byte pitch, 16-byte aligned, conflict-light)

__global__ __launch_bounds__(64) void attention_mfma_kernel(KKAttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t vt[64 * VT_LD];
  const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
  const int h = blockIdx.x, b = blockIdx.y, q0 = blockIdx.z * 32;
  const int L = kk_len(a.len, b);
  const bf16_t* base = (const bf16_t*)a.qkv + (long long)b * a.bs;
  bf16_t* ob = (bf16_t*)a.out + (long long)b * a.obs;
  const int qoff = h * HD, koff = a.hs + h * HD, voff = 2 * a.hs + h * HD;
  const int qi = q0 + c;
  if (q0 >= L) {  // whole tile is past the utterance: rows are zero (uniform)
    if (qi < a.Tmax) {
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int j = 0; j < 4; ++j) *(uint4*)(ob + (long long)qi * a.ldo + qoff + hh * 32 + j * 8) = z;
    }
    return;
  }
  const int qc = qi < L ? qi : L - 1;  // clamped: rows past L compute garbage-free duplicates that are never stored
  abf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const abf16x8*)(base + (long long)qc * a.ld + qoff + ks * 16 + hh * 8);

  af32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float sc = a.scale * 1.4426950408889634f;  // softmax in base 2: exp(x) = exp2(x * log2 e)

  for (int k0 = 0; k0 < L; k0 += 32) {
    // K fragments of this tile: lane = key row (clamped), 8 consecutive d per k-step
    const int kr = k0 + c < L ? k0 + c : L - 1;
    abf16x8 kf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const abf16x8*)(base + (long long)kr * a.ld + koff + ks * 16 + hh * 8);
    // V tile [32 keys][64 d]: lane loads keys (lane >> 3) + 8 i, d chunk (lane & 7) * 8
    uint4 vreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = k0 + (lane >> 3) + 8 * i;
      const int kc = key < L ? key : L - 1;
      vreg[i] = *(const uint4*)(base + (long long)kc * a.ld + voff + (lane & 7) * 8);
    }
    af32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], st, 0, 0, 0);
    // scores of query c against keys k0 + (r&3) + 8(r>>2) + 4hh
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      st[r] = key < L ? st[r] * sc : -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);  // finite: every tile holds at least one valid key
    const float corr = exp2f(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = exp2f(st[r] - mn);
      ps += st[r];
    }
    l = l * corr + ps;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] *= corr;
      o1[r] *= corr;
    }
    abf16x8 pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[s2][j] = (bf16_t)st[8 * s2 + j];
    // V^T through LDS with the key permutation (bits 2 and 3 of the key index swap); keys past L carry weight exp2(-inf) = 0
    __syncthreads();  // one wave: orders the previous tile's fragment reads before these writes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kl = (lane >> 3) + 8 * i;
      const int kp = (kl & 0x13) | ((kl & 4) << 1) | ((kl & 8) >> 1);
      U16 t;
      t.u = vreg[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) vt[((lane & 7) * 8 + e) * VT_LD + kp] = t.h[e];
    }
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const abf16x8 v0 = *(const abf16x8*)(vt + c * VT_LD + s2 * 16 + hh * 8);
      const abf16x8 v1 = *(const abf16x8*)(vt + (32 + c) * VT_LD + s2 * 16 + hh * 8);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf[s2], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf[s2], o1, 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 32);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (qi < a.Tmax) {
    const bool qv = qi < L;
    // lane owns query c and d = 32 t + (r&3) + 8(r>>2) + 4hh: four consecutive d per register group -> 8-byte stores
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        union { uint2 u; bf16_t hv[4]; } pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk.hv[e] = (bf16_t)(qv ? (t2 ? o1[4 * g + e] : o0[4 * g + e]) * inv : 0.f);
        *(uint2*)(ob + (long long)qi * a.ldo + qoff + 32 * t2 + 8 * g + 4 * hh) = pk.u;
      }
  }
}

}  // namespace

int kk_launch_albert_embed(const KKEmbedArgs& a, int B, int dtype, hipStream_t st) {
  if (B <= 0 || a.Tmax <= 0) return 0;
  if (a.E > 512) return kk_fail("albert_embed: E > 512");
  dim3 grid(kk_cdiv(a.Tmax, 4), B);
  if (dtype == KK_F32)
    hipLaunchKernelGGL(albert_embed_kernel<float>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(albert_embed_kernel<bf16_t>, grid, dim3(256), 0, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_attention(const KKAttnArgs& a, int B, int dtype, hipStream_t st) {
  if (B <= 0 || a.Tmax <= 0) return 0;
  if (a.hs != a.heads * HD) return kk_fail("attention: head size must be 64");
  static KKDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)attention_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS);
    (void)hipFuncSetAttribute((const void*)attention_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS);
    attr_once.done();
  }
  dim3 grid(a.heads, B, kk_cdiv(a.Tmax, QT));
  static int no_mfma = -1;
  if (no_mfma < 0) no_mfma = getenv("KK_ATTN_VALU") ? 1 : 0;  // A/B switch: the fp32 VALU kernel also handles bf16 tensors
  if (dtype == KK_BF16 && !no_mfma && a.ld % 8 == 0 && a.ldo % 8 == 0 && a.hs % 8 == 0 && !(((uintptr_t)a.qkv | (uintptr_t)a.out) & 15)) {
    hipLaunchKernelGGL(attention_mfma_kernel, grid, dim3(64), 0, st, a);
    KK_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == KK_F32)
    hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), ATT_LDS, st, a);
  else
    hipLaunchKernelGGL(attention_kernel<bf16_t>, grid, dim3(256), ATT_LDS, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}
