// Albert pieces that are not plain linears:
//   * AlbertEmbeddings (modules.py:451-465): word + position + token_type(0) -> LayerNorm(eps 1e-12)
//   * AlbertSelfAttention core (modules.py:497-512): softmax(Q K^T / sqrt(64) + mask) V per head.
// Keys past the utterance's length do not exist in the reference's batch-1 call; here they are
// simply not visited.  One workgroup per (utterance, head); each thread owns one query row and
// runs an online softmax over 64-key tiles of K/V staged in LDS (head size 64).
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

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

constexpr int HD = 64, KT = 64;

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(KKAttnArgs a) {
  // every lane reads the SAME key row (broadcast), so no padding is needed and rows can be read as float4
  __shared__ __attribute__((aligned(16))) float Ks[KT][HD];
  __shared__ __attribute__((aligned(16))) float Vs[KT][HD];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, h = blockIdx.x;
  const int L = kk_len(a.len, b);
  const T* base = (const T*)a.qkv + (long long)b * a.bs;
  T* ob = (T*)a.out + (long long)b * a.obs;
  const int qoff = h * HD, koff = a.hs + h * HD, voff = 2 * a.hs + h * HD;
  for (int q0 = 0; q0 < a.Tmax; q0 += 256) {
    const int qi = q0 + tid;
    const bool qv = qi < L;
    float q[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = qv ? kk_ld(base + (long long)qi * a.ld + qoff + d) * a.scale : 0.f;
      o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < L; k0 += KT) {
      __syncthreads();
      for (int e = tid; e < KT * HD; e += 256) {
        const int kr = e / HD, d = e - kr * HD;
        const int ki = k0 + kr;
        Ks[kr][d] = ki < L ? kk_ld(base + (long long)ki * a.ld + koff + d) : 0.f;
        Vs[kr][d] = ki < L ? kk_ld(base + (long long)ki * a.ld + voff + d) : 0.f;
      }
      __syncthreads();
      const int kn = min(KT, L - k0);
      for (int kr = 0; kr < kn; ++kr) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const float4 k4 = *(const float4*)&Ks[kr][d];
          s = __builtin_fmaf(q[d], k4.x, s);
          s = __builtin_fmaf(q[d + 1], k4.y, s);
          s = __builtin_fmaf(q[d + 2], k4.z, s);
          s = __builtin_fmaf(q[d + 3], k4.w, s);
        }
        const float mn = fmaxf(m, s);
        const float corr = expf(m - mn);
        const float p = expf(s - mn);
        l = l * corr + p;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const float4 v4 = *(const float4*)&Vs[kr][d];
          o[d] = __builtin_fmaf(p, v4.x, o[d] * corr);
          o[d + 1] = __builtin_fmaf(p, v4.y, o[d + 1] * corr);
          o[d + 2] = __builtin_fmaf(p, v4.z, o[d + 2] * corr);
          o[d + 3] = __builtin_fmaf(p, v4.w, o[d + 3] * corr);
        }
        m = mn;
      }
    }
    if (qi < a.Tmax) {
      const float inv = (qv && l > 0.f) ? 1.0f / l : 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) kk_st(ob + (long long)qi * a.ldo + qoff + d, qv ? o[d] * inv : 0.f);
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
  dim3 grid(a.heads, B);
  if (dtype == KK_F32)
    hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(attention_kernel<bf16_t>, grid, dim3(256), 0, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}
