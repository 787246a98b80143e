// Mimi codec DECODE path: Mimi.decode, mlx_audio/codec/models/mimi/mimi.py:147-154
//   codes [B][nq][Nf] -> split-RVQ decode -> depth-wise transposed-conv upsample (12.5 -> 25 Hz) -> 8-layer transformer ->
//   SEANet decoder (conv k7, 4 x {ELU, transposed conv ratio r, residual block}, ELU, conv k3) -> pcm [B][1920 Nf]
// Host orchestration + the kernels that exist only here (RVQ gather-sum, depth-wise upsample, RoPE).  Convolutions, linears,
// LayerNorm and attention are the Kokoro library's kernels (kk_conv.hip, kk_norm.hip, kk_albert.hip).  Activations are
// frames-major channels-last [B][L][C] fp32 (round 1: the fp32 path only; the bf16 MFMA kernels are shape-compatible).
//
// Reference semantics restated (file:line relative to the reference root):
//   quantization.py:25-28,41-43,97-101,135-139,178-182   embedding = embedding_sum / max(cluster_usage, 1e-5); sum of rows; 1x1 proj
//   conv.py:244-263   causal conv = left pad (k-1)*d zeros;  conv.py:323-333 causal transposed conv = drop the last k - stride
//   conv.py:82-95,379-401   depth-wise transposed conv k = 2*stride (dense eye-masked weight in the reference)
//   transformer.py:62-104   in_proj -> q,k,v; RoPE(traditional, base 10000) on q,k; softmax(q k^T / 8) v WITHOUT a mask in the
//                           non-streaming call (transformer.py:171 passes none) -> bidirectional; out_proj
//   transformer.py:126-134,163-177   x += ls1 * attn(LN1 x);  x += ls2 * W2 gelu_approx(W1 LN2 x)   (LayerScale folded into W)
//   seanet.py:98-107,219-225,270-283   ELU before every conv, true-skip residual blocks
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kokoro_hip.h"
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

struct PackedConv {  // generic-kernel pack [K][Cin][ldw] fp32 (+ bias); bf16 mode: MFMA fragment-order pack as well
  size_t w_off = 0, b_off = 0, wf_off = 0;
  bool has_bias = false, mfma = false;
  int Cin = 0, Cout = 0, K = 0, ldw = 0, CinP = 0, CoutP = 0;
  const float* w = nullptr;
  const float* b = nullptr;  // [CoutP] when mfma (zero padded), else [Cout]
  const bf16_t* wf = nullptr;
};
struct PackedVec {
  size_t off = 0;
  int n = 0;
  const float* p = nullptr;
};
struct MimiLayer {
  PackedVec n1w, n1b, n2w, n2b;
  PackedConv in_proj, out_proj, lin1, lin2;  // out_proj / lin2 carry the LayerScale
};
struct SeaLayer {
  PackedConv up, b0, b1;
  int ratio = 1;
};
struct DebugBuf {
  const void* p;
  int rows, C, ld;
  long long bs;
  int B, dtype;
};

}  // namespace

struct kk_mimi {
  kk_mimi_config cfg;
  std::map<std::string, std::vector<float>> host;
  std::vector<float> pack;
  float* dev = nullptr;
  bool finalized = false;
  int adt = KK_F32;  // activation dtype: KK_F32 (parity path) or KK_BF16 (MFMA convolutions, bf16 activations)
  PackedVec codebooks;  // [nq][bins][qdim]
  PackedVec inv_freq;   // [32]
  PackedVec up_w;       // [2*stride][dim]
  PackedConv proj_first, proj_rest, init_conv, final_conv;
  std::vector<MimiLayer> layers;
  std::vector<SeaLayer> sea;
  // encode side (present when the checkpoint holds "encoder.*"): SEANet encoder, encoder transformer, resampler, RVQ search
  bool has_encoder = false;
  PackedConv enc_init, enc_final, enc_down, inproj_first, inproj_rest;
  std::vector<SeaLayer> enc_sea;  // up = the strided down-sampling conv here
  std::vector<MimiLayer> enc_layers;
  std::vector<PackedConv> cb_dot;  // per code book: E^T as a 1x1 conv qdim -> bins (the x.e term of the distance)
  PackedVec c2;                    // [nq][bins] |e|^2 / 2
  std::map<std::string, DebugBuf> dbg;  // debug notes of the LAST call (any thread); dbg_mu makes concurrent callers of one codec safe
  std::mutex dbg_mu;
};

namespace {

// ------------------------------------------------------------------------------------------------------------- kernels
// one workgroup per (frame, item): column c of the first code book's row, and of the sum of the other rows in ascending
// code-book order (the reference's accumulation order, quantization.py:97-101)
template <typename T>
__global__ __launch_bounds__(256) void rvq_sum_kernel(const int* codes, const float* cb, int nq, int bins, int qdim, int Nf, int ld, T* q_first,
                                                      T* q_rest) {
  const int t = blockIdx.x, b = blockIdx.y;
  for (int c = threadIdx.x; c < qdim; c += blockDim.x) {
    const int* cd = codes + (long long)b * nq * Nf + t;
    int id = cd[0];
    id = id < 0 ? 0 : (id >= bins ? bins - 1 : id);
    kk_st(q_first + ((long long)b * Nf + t) * ld + c, cb[(long long)id * qdim + c]);
    float s = 0.f;
    for (int i = 1; i < nq; ++i) {
      int idi = cd[(long long)i * Nf];
      idi = idi < 0 ? 0 : (idi >= bins ? bins - 1 : idi);
      const float e = cb[((long long)i * bins + idi) * qdim + c];
      s = i == 1 ? e : s + e;
    }
    kk_st(q_rest + ((long long)b * Nf + t) * ld + c, s);
  }
}

// depth-wise transposed conv, kernel 2*s, stride s, causal (last s outputs dropped): output row p = s*t + j gets
// x[t] w[j] + x[t-1] w[j + s]
template <typename T>
__global__ __launch_bounds__(256) void upsample_dw_kernel(const T* x, long long xbs, const float* w, int C, int ld, int Lin, int s, T* out) {
  const int p = blockIdx.x, b = blockIdx.y;
  const int tt = p / s, j = p - tt * s;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = kk_ld(x + (long long)b * xbs + (long long)tt * ld + c) * w[(long long)j * C + c];
    if (tt > 0) v += kk_ld(x + (long long)b * xbs + (long long)(tt - 1) * ld + c) * w[(long long)(j + s) * C + c];
    kk_st(out + ((long long)b * Lin * s + p) * ld + c, v);
  }
}

// nn.RoPE(traditional): pairs (2i, 2i+1) of every q and k head rotated by pos * inv_freq[i]; qkv [B][T][3*D] in place
template <typename TT>
__global__ __launch_bounds__(256) void rope_kernel(TT* qkv, const float* inv_freq, int T, int D, int ld, int hd) {
  const int t = blockIdx.x, b = blockIdx.y;
  TT* row = qkv + ((long long)b * T + t) * ld;
  const int half = hd / 2, npairs = D / 2;
  for (int e = threadIdx.x; e < 2 * npairs; e += blockDim.x) {
    const int which = e / npairs, pr = e - which * npairs;  // q (0) or k (1)
    const int h = pr / half, i = pr - h * half;
    const float ang = (float)t * inv_freq[i];
    const float c = cosf(ang), s = sinf(ang);
    TT* p2 = row + which * D + h * hd + 2 * i;
    const float x0 = kk_ld(p2), x1 = kk_ld(p2 + 1);
    kk_st(p2, x0 * c - x1 * s);
    kk_st(p2 + 1, x0 * s + x1 * c);
  }
}

// SEANet's last conv (seanet.py:258-268): Cout = 1, so it is a dot product per sample -- HBM-bound (one read of x), not a GEMM.
// out[b][t] = bias + sum_k sum_c elu(x[b][t - pad + k][c]) w[k][c]   (rows before 0 are zero padding); w is the generic pack
// [K][Cin][ldw] read at column 0.  One lane per output sample; weights (K * Cin floats) sit in LDS.
template <typename T>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const T* x, int ld, int L, int Cin, int K, int pad, const float* w, int ldw,
                                                         const float* bias, int elu, float* out) {
  extern __shared__ float wsm[];
  for (int e = threadIdx.x; e < K * Cin; e += 256) wsm[e] = w[(long long)e * ldw];
  __syncthreads();
  const int b = blockIdx.y;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  float acc = bias ? bias[0] : 0.f;
  for (int k = 0; k < K; ++k) {
    const long long r = t - pad + k;
    if (r < 0 || r >= L) continue;
    const T* xr = x + ((long long)b * L + r) * ld;
    const float* wk = wsm + k * Cin;
    if (sizeof(T) == 2 && (Cin & 7) == 0 && (ld & 7) == 0) {  // bf16 rows as 16-byte vectors
      for (int c = 0; c < Cin; c += 8) {
        const uint4 q = *(const uint4*)(xr + c);
        const unsigned wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo = __uint_as_float(wd[j] << 16), hi = __uint_as_float(wd[j] & 0xFFFF0000u);
          if (elu) {
            lo = lo > 0.f ? lo : __expf(lo) - 1.0f;
            hi = hi > 0.f ? hi : __expf(hi) - 1.0f;
          }
          acc = __builtin_fmaf(lo, wk[c + 2 * j], acc);
          acc = __builtin_fmaf(hi, wk[c + 2 * j + 1], acc);
        }
      }
    } else {
      for (int c = 0; c < Cin; ++c) {
        float v = kk_ld(xr + c);
        if (elu) v = v > 0.f ? v : expf(v) - 1.0f;
        acc = __builtin_fmaf(v, wk[c], acc);
      }
    }
  }
  out[(long long)b * L + t] = acc;
}

// bf16 form of the same conv: the workgroup's 256 + K - 1 input rows are staged ONCE in LDS with coalesced 16-byte loads (ELU applied
// once per element, rows pitched 144 B so the per-lane row reads below are conflict-free), then one lane per output sample.
constexpr int C1_LD = 72;  // bf16 elements per LDS row (64 channels + 8 pad)
__global__ __launch_bounds__(256) void conv_cout1_bf16_kernel(const bf16_t* x, int ld, int L, int Cin, int K, int pad, const float* w, int ldw,
                                                              const float* bias, int elu, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c1sm[];
  bf16_t* xs = (bf16_t*)c1sm;                                     // [256 + K - 1][72]
  float* wsm = (float*)(c1sm + (size_t)(256 + K - 1) * C1_LD * 2);  // [K][Cin]
  const int b = blockIdx.y, tid = threadIdx.x;
  const long long t0 = (long long)blockIdx.x * 256;
  for (int e = tid; e < K * Cin; e += 256) wsm[e] = w[(long long)e * ldw];
  const int cpr = Cin / 8, nrow = 256 + K - 1;
  for (int e = tid; e < nrow * cpr; e += 256) {
    const int rr = e / cpr, ch = e - rr * cpr;
    const long long r = t0 - pad + rr;
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (r >= 0 && r < L) {
      q = *(const uint4*)(x + ((long long)b * L + r) * ld + ch * 8);
      if (elu) {
        unsigned wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo = __uint_as_float(wd[j] << 16), hi = __uint_as_float(wd[j] & 0xFFFF0000u);
          lo = lo > 0.f ? lo : __expf(lo) - 1.0f;
          hi = hi > 0.f ? hi : __expf(hi) - 1.0f;
          typedef __bf16 b2 __attribute__((ext_vector_type(2)));
          const b2 pk = {(bf16_t)lo, (bf16_t)hi};
          wd[j] = __builtin_bit_cast(unsigned, pk);
        }
        q = make_uint4(wd[0], wd[1], wd[2], wd[3]);
      }
    }
    *(uint4*)(xs + rr * C1_LD + ch * 8) = q;
  }
  __syncthreads();
  const long long t = t0 + tid;
  if (t >= L) return;
  float acc = bias ? bias[0] : 0.f;
  for (int k = 0; k < K; ++k) {
    const bf16_t* xr = xs + (tid + k) * C1_LD;
    const float* wk = wsm + k * Cin;
    for (int c = 0; c < Cin; c += 8) {
      const uint4 q = *(const uint4*)(xr + c);
      const float4 wa = *(const float4*)(wk + c), wb = *(const float4*)(wk + c + 4);
      acc = __builtin_fmaf(__uint_as_float(q.x << 16), wa.x, acc);
      acc = __builtin_fmaf(__uint_as_float(q.x & 0xFFFF0000u), wa.y, acc);
      acc = __builtin_fmaf(__uint_as_float(q.y << 16), wa.z, acc);
      acc = __builtin_fmaf(__uint_as_float(q.y & 0xFFFF0000u), wa.w, acc);
      acc = __builtin_fmaf(__uint_as_float(q.z << 16), wb.x, acc);
      acc = __builtin_fmaf(__uint_as_float(q.z & 0xFFFF0000u), wb.y, acc);
      acc = __builtin_fmaf(__uint_as_float(q.w << 16), wb.z, acc);
      acc = __builtin_fmaf(__uint_as_float(q.w & 0xFFFF0000u), wb.w, acc);
    }
  }
  out[(long long)b * L + t] = acc;
}

// 'edge' padding of the resampler (conv.py:350-367, mx.pad mode="edge"): out row p = x[clamp(p - left, 0, L-1)]
__global__ __launch_bounds__(256) void edge_pad_kernel(const float* x, int L, int C, int left, int Lp, float* out) {
  const int p = blockIdx.x, b = blockIdx.y;
  int r = p - left;
  r = r < 0 ? 0 : (r >= L ? L - 1 : r);
  for (int c = threadIdx.x; c < C; c += 256) out[((long long)b * Lp + p) * C + c] = x[((long long)b * L + r) * C + c];
}

// EuclideanCodebook.encode (quantization.py:35-39): idx = argmin_j (c2[j] - x.e_j), first index on ties; then the residual loses
// the chosen row (quantization.py:84-92).  dots [B][T][bins] come from the 1x1 conv with E^T.
__global__ __launch_bounds__(256) void rvq_argmin_kernel(const float* dots, const float* c2, const float* cb, int bins, int qdim, int T, int nq,
                                                         int cbi, float* residual, int* codes) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* d = dots + ((long long)b * T + t) * bins;
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int j = tid; j < bins; j += 256) {
    const float v = c2[j] - d[j];
    if (v < best) { best = v; bi = j; }  // ascending j per thread: the first minimum wins
  }
  sv[tid] = best; si[tid] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float v = sv[tid + s];
      const int i = si[tid + s];
      if (v < sv[tid] || (v == sv[tid] && i < si[tid])) { sv[tid] = v; si[tid] = i; }
    }
    __syncthreads();
  }
  const int idx = si[0];
  if (tid == 0) codes[((long long)b * nq + cbi) * T + t] = idx;
  float* r = residual + ((long long)b * T + t) * qdim;
  for (int c = tid; c < qdim; c += 256) r[c] -= cb[(long long)idx * qdim + c];
}


// ---- linears and convolutions over a FEW rows (the streaming steps: 2 rows per item and frame): the work is reading W once.  The generic
// conv tile (64 rows x 64 columns per workgroup) runs a 2-row linear at ~100 us and the first SEANet layers (k 7 512 -> 1024 on 2 + 6 rows,
// transposed k 16 1024 -> 512 on 1 + 2 rows: 15 / 34 MB of weights) at 320-360 us.  Here a workgroup owns 16 output columns, its 256 threads
// are 16 columns x 16 k-slices, a thread walks its slice of K' with LR_U loads in flight and keeps MT row accumulators; the 16 slices meet in
// LDS in slice order (fixed: a row's bits do not depend on its batch neighbours or on MT).  Three forms share the loop:
//   * linear:                      K' = Cin;            row m = (item, r) reads x row r;
//   * causal conv, stride 1, pad 0 over rows with pitch == Cin: the k taps of output row r are the CONTIGUOUS window x[r .. r + k - 1], so
//     it is a linear with K' = k * Cin on the flattened window and W [k][Cin][N] read as [K'][N];
//   * transposed conv with k = 2 * stride (polyphase, blockIdx.z = phase p): output row p + s * q = W[p] x[q] + W[p + s] x[q - 1], a linear
//     with K' = 2 * Cin over the window x[q - 1 .. q] whose first half meets weight rows (p + s) * Cin .. and second half p * Cin ..;
//     q runs from 1 (row 0 of x is the carried previous input row; output rows [0, s) are not written).
// An ELU on the input is applied by a pre-pass into scratch (Run::elu_tmp): inside the loop every element would be transformed by 16 lanes
// of every workgroup.  Epilogue as conv_generic_kernel's: bias, tanh-GELU, + residual, + old output.
constexpr int LR_COLS = 16, LR_SL = 16, LR_U = 8;
constexpr int LR_MAX_ROWS = 128;  // output rows per item up to which the few-rows kernel is used (beyond: the tiled generic kernel)
struct LinRowsArgs {
  const float* x; long long xbs; int ldx, rows;   // rows: output rows per item (per phase for the transposed form)
  const float* w; int ldw;
  const float* bias;
  const float* res; long long rbs; int ldr;
  float* out; long long obs; int ldo;
  int K, N, M, act, accumulate;
  int Cin, tstride;  // tstride > 0: transposed form with that stride
};
template <int MT>
__global__ __launch_bounds__(256) void linear_rows_kernel(LinRowsArgs a) {
  __shared__ float red[LR_SL][MT][LR_COLS + 1];
  const int tid = threadIdx.x, col = tid & (LR_COLS - 1), sl = tid >> 4;
  const int n = blockIdx.x * LR_COLS + col, m0 = blockIdx.y * MT, phase = blockIdx.z;
  const int nc = n < a.N ? n : a.N - 1;
  const int kper = (a.K + LR_SL - 1) / LR_SL, k0 = sl * kper, k1 = min(a.K, k0 + kper);
  const float* xr[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mm = m0 + m < a.M ? m0 + m : a.M - 1;
    xr[m] = a.x + (long long)(mm / a.rows) * a.xbs + (long long)(mm % a.rows) * a.ldx;  // (transposed: window rows q - 1, q with q = mm % rows + 1)
  }
  // weight row of flattened index k: identity, or the two taps of this phase
  const int wr0 = a.tstride ? (phase + a.tstride) * a.Cin : 0, wr1 = a.tstride ? phase * a.Cin - a.Cin : 0;
  float acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
  for (int k = k0; k < k1; k += LR_U) {
    float wv[LR_U];
#pragma unroll
    for (int u = 0; u < LR_U; ++u) {
      const int kk = k + u < k1 ? k + u : k1 - 1;  // (clamped: no load under a condition)
      const int wrow = a.tstride ? (kk < a.Cin ? wr0 + kk : wr1 + kk) : kk;
      wv[u] = a.w[(long long)wrow * a.ldw + nc];
    }
#pragma unroll
    for (int u = 0; u < LR_U; ++u) {
      const float live = k + u < k1 ? 1.0f : 0.0f;
      const int kk = k + u < k1 ? k + u : k1 - 1;
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_fmaf(xr[m][kk] * live, wv[u], acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) red[sl][m][col] = acc[m];
  __syncthreads();
  for (int o = tid; o < MT * LR_COLS; o += 256) {
    const int m = o / LR_COLS, c = o - m * LR_COLS;
    const int mg = m0 + m, ng = blockIdx.x * LR_COLS + c;
    if (mg >= a.M || ng >= a.N) continue;
    float v = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < LR_SL; ++s2) v += red[s2][m][c];  // slice order
    v += a.bias ? a.bias[ng] : 0.f;
    if (a.act == KK_ACT_GELU_TANH) v = 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v))));  // nn.gelu_approx
    const long long b = mg / a.rows;
    const long long r = a.tstride ? phase + (long long)a.tstride * (mg % a.rows + 1) : mg % a.rows;
    if (a.res) v += a.res[b * a.rbs + r * a.ldr + ng];
    float* o_ = a.out + b * a.obs + r * a.ldo + ng;
    if (a.accumulate) v += *o_;
    *o_ = v;
  }
}
// nn.elu of [B][rows][C] (pitch C) into dense scratch: the pre-pass of the few-rows kernel
__global__ __launch_bounds__(256) void elu_rows_kernel(const float* x, long long xbs, long long n, float* out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float v = x[(long long)blockIdx.y * xbs + e];
  out[(long long)blockIdx.y * n + e] = v > 0.f ? v : expf(v) - 1.0f;
}

// ------------------------------------------------------------------------------------------------------------- host
int rup(int v, int m) { return (v + m - 1) / m * m; }
uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

struct Packer {
  kk_mimi* m;
  std::string err;
  size_t alloc(size_t n) {
    const size_t off = (m->pack.size() + 63) & ~(size_t)63;
    m->pack.resize(off + n, 0.f);
    return off;
  }
  const std::vector<float>* get(const std::string& name, size_t n) {
    auto it = m->host.find(name);
    if (it == m->host.end()) {
      if (err.empty()) err = "missing parameter: " + name;
      return nullptr;
    }
    if (it->second.size() != n) {
      if (err.empty()) err = "unexpected size for " + name;
      return nullptr;
    }
    return &it->second;
  }
  PackedVec vec(const std::string& name, size_t n) {
    PackedVec r;
    const std::vector<float>* v = get(name, n);
    if (!v) return r;
    r.n = (int)n;
    r.off = alloc(n);
    memcpy(&m->pack[r.off], v->data(), n * 4);
    return r;
  }
  // MLX conv / conv-transpose weight [O][K][I] (+ bias [O]) -> [K][I][ldw]; `row_scale` (LayerScale) multiplies output row o
  PackedConv conv(const std::string& wname, const std::string& bname, int O, int K, int I, const std::vector<float>* row_scale = nullptr) {
    PackedConv c;
    const std::vector<float>* w = get(wname, (size_t)O * K * I);
    if (!w) return c;
    c.Cin = I; c.Cout = O; c.K = K; c.ldw = rup(O, 64);
    c.w_off = alloc((size_t)K * I * c.ldw);
    float* dst = &m->pack[c.w_off];
    for (int o = 0; o < O; ++o)
      for (int k = 0; k < K; ++k)
        for (int i = 0; i < I; ++i) {
          float v = (*w)[((size_t)o * K + k) * I + i];
          if (row_scale) v *= (*row_scale)[o];
          dst[((size_t)k * I + i) * c.ldw + o] = v;
        }
    // bf16 mode: the same weights in MFMA fragment order for the variant-4 kernel (needs Cout % 8 == 0, >= 16 channels both ways)
    c.mfma = m->adt == KK_BF16 && O % 8 == 0 && O >= 16 && I >= 16;
    if (c.mfma) {
      c.CinP = rup(I, 64);
      c.CoutP = rup(O, 128);
      const size_t nel = (size_t)K * c.CoutP * c.CinP;
      c.wf_off = alloc((nel + 1) / 2);
      w = get(wname, (size_t)O * K * I);  // (alloc may have moved nothing here, but keep the pointer fresh)
      uint16_t* dfr = (uint16_t*)&m->pack[c.wf_off];
      for (int o = 0; o < O; ++o)
        for (int k = 0; k < K; ++k)
          for (int i = 0; i < I; ++i) {
            float v = (*w)[((size_t)o * K + k) * I + i];
            if (row_scale) v *= (*row_scale)[o];
            dfr[kk_mfma4_pack_index(k, o, i, c.CoutP, c.CinP)] = f32_to_bf16_rne(v);
          }
    }
    const int nb = c.mfma ? c.CoutP : O;
    if (!bname.empty() || c.mfma) {
      c.has_bias = true;
      c.b_off = alloc(nb);  // zero filled
      if (!bname.empty()) {
        const std::vector<float>* b = get(bname, (size_t)O);
        if (!b) return c;
        memcpy(&m->pack[c.b_off], b->data(), (size_t)O * 4);
      }
    }
    return c;
  }
};

void resolve(kk_mimi* m, PackedConv& c) {
  c.w = m->dev + c.w_off;
  c.b = c.has_bias ? m->dev + c.b_off : nullptr;
  c.wf = c.mfma ? (const bf16_t*)(m->dev + c.wf_off) : nullptr;
}
void resolve(kk_mimi* m, PackedVec& v) { v.p = v.n ? m->dev + v.off : nullptr; }

struct Act {  // an activation tensor [B][rows][ld], C valid channels
  void* p = nullptr;
  int rows = 0, C = 0, ld = 0, dtype = KK_F32;
  long long bstride = 0;  // 0: dense (rows * ld); set for a row range of a larger per-item buffer
  long long bs() const { return bstride ? bstride : (long long)rows * ld; }
};

struct Run {
  kk_mimi* m;
  hipStream_t st;
  int B;
  char* base;
  size_t cap, used;
  bool dry;
  bool oom = false;
  int adt = -1;  // activation dtype of this run (-1: the model's)
  bool no_lin_rows = false;  // (A/B: the generic conv kernel for the few-row layers too)
  float* elu_tmp = nullptr;  // dense scratch for an ELU'd input of the few-rows kernel (streaming steps allocate it)
  size_t elu_floats = 0;
  void* raw(size_t bytes) {
    const size_t off = (used + 255) & ~(size_t)255;
    used = off + bytes;
    if (dry) return nullptr;
    if (used > cap) { oom = true; return nullptr; }
    return base + off;
  }
  // bf16 tensors get a pitch that is a multiple of 64 so any of them can feed the MFMA kernel (pad channels are never read as
  // data: the kernel masks channels >= Cin)
  Act act(int rows, int C, int dtype = -1) {
    Act t;
    t.dtype = dtype < 0 ? (adt < 0 ? m->adt : adt) : dtype;
    t.rows = rows; t.C = C;
    t.ld = t.dtype == KK_BF16 ? rup(C, 64) : C;
    t.p = raw((size_t)B * rows * t.ld * (t.dtype == KK_BF16 ? 2 : 4));
    return t;
  }
  // conv / transposed conv / linear
  int conv(const PackedConv& w, const Act& x, const Act& out, int pad, int dil, bool transposed, int stride, int in_act, int act_,
           const Act* res, int accumulate) {
    if (dry) return 0;
    const int Lin = x.rows, Lout = out.rows;
    if (w.mfma && x.dtype == KK_BF16 && out.dtype == KK_BF16 && x.ld >= w.CinP && kk_mfma_eligible(w.Cin, w.Cout, w.K, transposed ? KK_CONVT : KK_CONV, stride, dil)) {
      KKMfmaArgs g;
      memset(&g, 0, sizeof g);
      g.x = (const bf16_t*)x.p; g.xbs = x.bs(); g.ldx = x.ld; g.wf = w.wf; g.CinP = w.CinP; g.Cin = w.Cin; g.CoutP = w.CoutP; g.bias = w.b;
      g.out = out.p; g.obs = out.bs(); g.ldo = out.ld;
      if (res) { g.res = res->p; g.rbs = res->bs(); g.ldr = res->ld; }
      g.Cout = w.Cout; g.Kw = w.K; g.mode = transposed ? KK_CONVT : KK_CONV; g.stride = stride; g.pad = pad; g.dil = dil;
      g.Q = transposed ? kk_cdiv(Lout, stride) : Lout; g.Lo_rows = Lout;
      g.lin = KKLen{nullptr, 0, Lin}; g.lout = KKLen{nullptr, 0, Lout};
      g.in_slope = 1.f; g.scale = 1.f; g.accumulate = accumulate; g.act = act_; g.in_act = in_act;
      return kk_launch_conv_mfma4(g, B, KK_BF16, st);
    }
    // few rows per item (the streaming steps): the weight-streaming kernel instead of the 64-row conv tile.  Chosen by the rows per ITEM,
    // never by B (batch invariance).  Forms: linear; causal conv over contiguous rows (pad 0, stride 1, pitch == Cin); transposed conv k = 2 s.
    {
      const bool f32io = x.dtype == KK_F32 && out.dtype == KK_F32 && dil == 1 && pad == 0 && w.Cout >= 32 && !no_lin_rows &&
                         (act_ == KK_ACT_NONE || act_ == KK_ACT_GELU_TANH) && (in_act == 0 || (in_act == KK_ACT_ELU && elu_tmp));
      const bool lin = f32io && !transposed && stride == 1 && x.ld == w.Cin && x.rows == out.rows + w.K - 1 && out.rows <= LR_MAX_ROWS;
      const bool ctr = f32io && transposed && w.K == 2 * stride && x.ld == w.Cin && out.rows == x.rows * stride && x.rows >= 2 &&
                       x.rows - 1 <= LR_MAX_ROWS && !res && !accumulate;
      if (lin || ctr) {
        LinRowsArgs g;
        memset(&g, 0, sizeof g);
        g.x = (const float*)x.p; g.xbs = x.bs(); g.ldx = x.ld;
        if (in_act == KK_ACT_ELU) {  // ELU once, into dense scratch
          const long long n = (long long)x.rows * x.ld;
          if ((size_t)B * n > elu_floats) return kk_fail("kk_mimi: internal: ELU scratch too small");
          hipLaunchKernelGGL(elu_rows_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, (const float*)x.p, x.bs(), n, elu_tmp);
          KK_CHECK_LAUNCH();
          g.x = elu_tmp; g.xbs = n;
        }
        g.rows = ctr ? x.rows - 1 : out.rows;
        g.w = w.w; g.ldw = w.ldw; g.bias = w.b;
        if (res) { g.res = (const float*)res->p; g.rbs = res->bs(); g.ldr = res->ld; }
        g.out = (float*)out.p; g.obs = out.bs(); g.ldo = out.ld;
        g.Cin = w.Cin; g.tstride = ctr ? stride : 0;
        g.K = ctr ? 2 * w.Cin : w.K * w.Cin; g.N = w.Cout; g.M = B * g.rows; g.act = act_; g.accumulate = accumulate;
        const int M = g.M, nz = ctr ? stride : 1;
        if (M <= 4) hipLaunchKernelGGL(linear_rows_kernel<4>, dim3(kk_cdiv(g.N, LR_COLS), kk_cdiv(M, 4), nz), dim3(256), 0, st, g);
        else if (M <= 8) hipLaunchKernelGGL(linear_rows_kernel<8>, dim3(kk_cdiv(g.N, LR_COLS), kk_cdiv(M, 8), nz), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(linear_rows_kernel<16>, dim3(kk_cdiv(g.N, LR_COLS), kk_cdiv(M, 16), nz), dim3(256), 0, st, g);
        KK_CHECK_LAUNCH();
        return 0;
      }
    }
    KKConvArgs a;
    memset(&a, 0, sizeof a);
    a.x = x.p; a.xbs = x.bs(); a.ldx = x.ld;
    a.w = w.w; a.ldw = w.ldw; a.bias = w.b;
    a.out = out.p; a.obs = out.bs(); a.ldo = out.ld;
    if (res) { a.res = res->p; a.rbs = res->bs(); a.ldr = res->ld; }
    a.Cin = w.Cin; a.Cout = w.Cout; a.Kw = w.K;
    a.mode = transposed ? KK_CONVT : KK_CONV; a.stride = stride; a.pad = pad; a.dil = dil;
    a.Q = transposed ? kk_cdiv(Lout, stride) : Lout; a.Lo_rows = Lout;
    a.lin = KKLen{nullptr, 0, Lin}; a.lout = KKLen{nullptr, 0, Lout};
    a.in_slope = 1.f; a.scale = 1.f; a.accumulate = accumulate; a.act = act_; a.in_act = in_act;
    return kk_launch_conv_generic(a, B, x.dtype, out.dtype, st);
  }
  int layernorm(const Act& x, const Act& out, const float* w, const float* b) {
    if (dry) return 0;
    KKLnArgs a;
    memset(&a, 0, sizeof a);
    a.x = x.p; a.xbs = x.bs(); a.ldx = x.ld; a.out = out.p; a.obs = out.bs(); a.ldo = out.ld; a.C = x.C; a.Lmax = x.rows;
    a.len = KKLen{nullptr, 0, x.rows}; a.w = w; a.bias = b; a.eps = 1e-5f; a.act = KK_ACT_NONE;
    return kk_launch_layernorm(a, B, x.dtype, st);
  }
  void note(const char* name, const Act& t) {
    if (!dry) {
      std::lock_guard<std::mutex> lk(m->dbg_mu);
      m->dbg[name] = DebugBuf{t.p, t.rows, t.C, t.ld, t.bs(), B, t.dtype};
    }
  }
};

#define MM_TRY(x)        \
  do {                   \
    const int rc__ = (x); \
    if (rc__ != 0) return rc__; \
  } while (0)

// x [B][T][D] updated in place by the layers (transformer.py:137-177); n, qkv, att, hbuf are scratch of the same row count
int run_transformer(Run& r, const std::vector<MimiLayer>& layers, Act& x, Act& n, Act& qkv, Act& att, Act& hbuf) {
  kk_mimi* m = r.m;
  const kk_mimi_config& c = m->cfg;
  const int B = r.B, D = c.dim, T = x.rows;
  const bool bf = x.dtype == KK_BF16;
  for (size_t l = 0; l < layers.size(); ++l) {
    const MimiLayer& L = layers[l];
    MM_TRY(r.layernorm(x, n, L.n1w.p, L.n1b.p));
    MM_TRY(r.conv(L.in_proj, n, qkv, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    if (!r.dry) {
      if (bf)
        hipLaunchKernelGGL(rope_kernel<bf16_t>, dim3(T, B), dim3(256), 0, r.st, (bf16_t*)qkv.p, m->inv_freq.p, T, D, qkv.ld, D / c.num_heads);
      else
        hipLaunchKernelGGL(rope_kernel<float>, dim3(T, B), dim3(256), 0, r.st, (float*)qkv.p, m->inv_freq.p, T, D, qkv.ld, D / c.num_heads);
      KK_CHECK_LAUNCH();
      KKAttnArgs a;
      memset(&a, 0, sizeof a);
      a.qkv = qkv.p; a.bs = qkv.bs(); a.ld = qkv.ld; a.out = att.p; a.obs = att.bs(); a.ldo = att.ld;
      a.heads = c.num_heads; a.hs = D; a.Tmax = T; a.len = KKLen{nullptr, 0, T}; a.scale = 1.0f / sqrtf((float)(D / c.num_heads));
      MM_TRY(kk_launch_attention(a, B, qkv.dtype, r.st));
    }
    MM_TRY(r.conv(L.out_proj, att, x, 0, 1, false, 1, 0, KK_ACT_NONE, &x, 0));  // x += ls1 * (W att)
    MM_TRY(r.layernorm(x, n, L.n2w.p, L.n2b.p));
    MM_TRY(r.conv(L.lin1, n, hbuf, 0, 1, false, 1, 0, KK_ACT_GELU_TANH, nullptr, 0));
    MM_TRY(r.conv(L.lin2, hbuf, x, 0, 1, false, 1, 0, KK_ACT_NONE, &x, 0));     // x += ls2 * (W2 gelu(W1 n))
  }
  return 0;
}

// SEANet decoder (seanet.py:228-283) on x [B][T][dim] -> pcm [B][T * prod(ratios)]
int run_seanet(Run& r, const Act& x, float* pcm, const char* who) {
  kk_mimi* m = r.m;
  const kk_mimi_config& c = m->cfg;
  const int B = r.B, T = x.rows;
  (void)who;
  Act y = r.act(T, m->init_conv.Cout);
  if (r.oom) return kk_fail("kk_mimi_decode: workspace too small");
  MM_TRY(r.conv(m->init_conv, x, y, (c.ksize - 1), 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  static const char* lname[8] = {"layer0", "layer1", "layer2", "layer3", "layer4", "layer5", "layer6", "layer7"};
  for (size_t l = 0; l < m->sea.size(); ++l) {
    const SeaLayer& S = m->sea[l];
    const int Lo = y.rows * S.ratio, Co = S.up.Cout;
    Act u = r.act(Lo, Co), hb = r.act(Lo, S.b0.Cout), o = r.act(Lo, Co);
    if (r.oom) return kk_fail("kk_mimi_decode: workspace too small");
    MM_TRY(r.conv(S.up, y, u, 0, 1, true, S.ratio, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    MM_TRY(r.conv(S.b0, u, hb, (c.residual_ksize - 1), 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    MM_TRY(r.conv(S.b1, hb, o, 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, &u, 0));
    r.note(l < 8 ? lname[l] : "layerN", o);
    y = o;
  }
  Act out;
  out.p = pcm; out.rows = y.rows; out.C = 1; out.ld = 1; out.dtype = KK_F32;
  if (m->final_conv.Cout == 1 && (size_t)m->final_conv.K * m->final_conv.Cin * 4 <= 32768) {
    if (!r.dry) {
      const PackedConv& w = m->final_conv;
      const size_t lds = (size_t)w.K * w.Cin * 4;
      dim3 grid(kk_cdiv(y.rows, 256), B);
      if (y.dtype == KK_BF16 && w.Cin % 8 == 0 && w.Cin <= 64 && y.ld % 8 == 0 && w.K <= 16) {
        const size_t l2 = (size_t)(256 + w.K - 1) * C1_LD * 2 + (size_t)w.K * w.Cin * 4;
        hipLaunchKernelGGL(conv_cout1_bf16_kernel, grid, dim3(256), l2, r.st, (const bf16_t*)y.p, y.ld, y.rows, w.Cin, w.K, c.last_ksize - 1, w.w, w.ldw, w.b, 1, pcm);
      } else if (y.dtype == KK_BF16)
        hipLaunchKernelGGL(conv_cout1_kernel<bf16_t>, grid, dim3(256), lds, r.st, (const bf16_t*)y.p, y.ld, y.rows, w.Cin, w.K, c.last_ksize - 1, w.w, w.ldw, w.b, 1, pcm);
      else
        hipLaunchKernelGGL(conv_cout1_kernel<float>, grid, dim3(256), lds, r.st, (const float*)y.p, y.ld, y.rows, w.Cin, w.K, c.last_ksize - 1, w.w, w.ldw, w.b, 1, pcm);
      KK_CHECK_LAUNCH();
    }
    return 0;
  }
  MM_TRY(r.conv(m->final_conv, y, out, (c.last_ksize - 1), 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
  return 0;
}

int run_decode(Run& r, int Nf, const int* codes, float* pcm) {
  kk_mimi* m = r.m;
  const kk_mimi_config& c = m->cfg;
  const int B = r.B, D = c.dim, Q = c.qdim, T = Nf * c.upsample_stride;
  const bool bf = m->adt == KK_BF16;
  // ---- split RVQ decode
  Act q1 = r.act(Nf, Q), q2 = r.act(Nf, Q), x0 = r.act(Nf, D), xu = r.act(T, D), x = r.act(T, D);
  Act n = r.act(T, D), qkv = r.act(T, 3 * D), att = r.act(T, D), hbuf = r.act(T, c.dim_feedforward);
  if (r.oom) return kk_fail("kk_mimi_decode: workspace too small");
  if (!r.dry) {
    if (bf) {
      hipLaunchKernelGGL(rvq_sum_kernel<bf16_t>, dim3(Nf, B), dim3(256), 0, r.st, codes, m->codebooks.p, c.nq, c.bins, Q, Nf, q1.ld, (bf16_t*)q1.p, (bf16_t*)q2.p);
    } else {
      hipLaunchKernelGGL(rvq_sum_kernel<float>, dim3(Nf, B), dim3(256), 0, r.st, codes, m->codebooks.p, c.nq, c.bins, Q, Nf, q1.ld, (float*)q1.p, (float*)q2.p);
    }
    KK_CHECK_LAUNCH();
  }
  MM_TRY(r.conv(m->proj_first, q1, x0, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  if (c.nq > 1) MM_TRY(r.conv(m->proj_rest, q2, x0, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 1));
  r.note("quantized", x0);
  if (!r.dry) {
    if (bf)
      hipLaunchKernelGGL(upsample_dw_kernel<bf16_t>, dim3(T, B), dim3(256), 0, r.st, (const bf16_t*)x0.p, x0.bs(), m->up_w.p, D, x0.ld, Nf, c.upsample_stride, (bf16_t*)xu.p);
    else
      hipLaunchKernelGGL(upsample_dw_kernel<float>, dim3(T, B), dim3(256), 0, r.st, (const float*)x0.p, x0.bs(), m->up_w.p, D, x0.ld, Nf, c.upsample_stride, (float*)xu.p);
    KK_CHECK_LAUNCH();
    // xu is kept for the debug hook: the transformer updates x in place
    if (hipMemcpyAsync(x.p, xu.p, (size_t)B * T * xu.ld * (bf ? 2 : 4), hipMemcpyDeviceToDevice, r.st) != hipSuccess) return kk_fail("kk_mimi_decode: copy failed");
  }
  r.note("upsampled", xu);
  // ---- transformer
  MM_TRY(run_transformer(r, m->layers, x, n, qkv, att, hbuf));
  r.note("transformer", x);
  // ---- SEANet decoder
  return run_seanet(r, x, pcm, "kk_mimi_decode");
}


// ------------------------------------------------------------------------------------------------------------- streaming
// Mimi.decode_step / encode_step (mimi.py:156-168), MimiStreamingDecoder (mimi.py:264-306).  Every module carries the state the
// reference's streaming modules carry (conv.py:265-351):
//   * StreamableConv1d._prev_xs: the last (k - 1) * dilation + 1 - stride INPUT rows of each causal convolution.  Here a StateBuf per
//     convolution, [item][S carried rows | n rows of this step]: the producer writes its rows straight behind the carried ones, the
//     convolution runs without padding over S + n rows, then the last S rows move to the front.  A fresh stream holds zeros there
//     (the left padding of the first step; 'edge' for the resampler, filled from the first row).
//   * StreamableConvTranspose1d._prev_ys (k = 2 * stride): the partial sums of the last stride output rows.  An output row depends on two
//     input rows, so carrying the previous INPUT row (one row instead of stride rows of partial sums) and running the transposed
//     convolution over [previous | new] rows gives the same sums; the first `stride` rows of that run repeat the previous step's and are
//     dropped.  ELU(0) = 0 and no bias on a zero row: a zero row stands for "no previous input".
//   * the transformers' KV caches (positions of one step see each other and the last `context` cached ones, transformer.py:79-104).
// Chunks are whole code frames, so every layer sees a fixed number of rows per step and no module ever has to hold back a partial stride.
struct StateBuf {
  float* p = nullptr;
  int S = 0, n = 0, C = 0, maxB = 0;
  int nmax = 0;  // rows of the stream's LARGEST step: the per-item pitch is (S + nmax) * C whatever this step's n is
  size_t floats() const { return (size_t)maxB * (S + nmax) * C; }
  long long pitch() const { return (long long)(S + nmax) * C; }
  Act all() const {
    Act t;
    t.p = p; t.rows = S + n; t.C = C; t.ld = C; t.dtype = KK_F32;
    t.bstride = pitch();
    return t;
  }
  Act fresh() const {
    Act t;
    t.p = p + (size_t)S * C; t.rows = n; t.C = C; t.ld = C; t.dtype = KK_F32;
    t.bstride = pitch();
    return t;
  }
};

// rows [n, n + S) -> rows [0, S) of every item; the ranges overlap when n < S, so a block reads everything before it writes
constexpr int SHIFT_PER_THREAD = 32;
__global__ __launch_bounds__(256) void state_shift_kernel(float* p, int S, int n, int C, long long pitch) {
  float* q = p + (long long)blockIdx.x * pitch;
  const int total = S * C;
  float v[SHIFT_PER_THREAD];
#pragma unroll
  for (int k = 0; k < SHIFT_PER_THREAD; ++k) {
    const int e = threadIdx.x + k * 256;
    v[k] = e < total ? q[(long long)n * C + e] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SHIFT_PER_THREAD; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < total) q[e] = v[k];
  }
}
// 'edge' left padding of a fresh resampler state (conv.py:265-281 with pad_mode "edge"): the carried rows repeat the first new row
__global__ __launch_bounds__(256) void state_edge_fill_kernel(float* p, int S, int n, int C, long long pitch) {
  float* q = p + (long long)blockIdx.x * pitch;
  for (int e = threadIdx.x; e < S * C; e += 256) q[e] = q[(long long)S * C + e % C];
}

int state_shift(const StateBuf& b, int B, hipStream_t st) {
  if (b.S == 0) return 0;
  if (b.S * b.C > 256 * SHIFT_PER_THREAD) return kk_fail("mimi stream: carried state larger than the shift kernel takes");
  hipLaunchKernelGGL(state_shift_kernel, dim3(B), dim3(256), 0, st, b.p, b.S, b.n, b.C, b.pitch());
  KK_CHECK_LAUNCH();
  return 0;
}

}  // namespace

struct kk_mimi_stream {
  kk_mimi* m = nullptr;
  int max_batch = 0, max_pos = 0, context = 250;
  bool encoder = false;
  int chunk = 1;           // code frames of the NEXT step (kk_mimi_stream_set_chunk; the state carries over)
  int max_chunk = 1;       // largest step the state buffers and the workspace are sized for
  float* kc = nullptr;     // [layers][maxB][max_pos][dim]
  float* vc = nullptr;
  float* rope = nullptr;   // [max_pos][hd/2][2]
  float* pool = nullptr;   // every StateBuf below
  size_t pool_floats = 0;
  // decode: resampler (previous quantised frame), init conv, per layer [transposed conv | residual block], last conv
  // encode: init conv, per layer [residual block | strided conv], last conv, resampler
  StateBuf first, last, resample;
  std::vector<StateBuf> up, blk;
  int frames = 0, pos = 0, B = 0;
  bool fresh = true;
};

namespace {

// one step of a transformer stack over the stream's KV caches: x [B][T rows][D] in place (transformer.py:79-104,137-177)
int run_transformer_step(Run& r, kk_mimi_stream* s, const std::vector<MimiLayer>& layers, Act& x, Act& n, Act& qkv, Act& att, Act& hbuf) {
  const kk_mimi_config& c = r.m->cfg;
  const int B = r.B, D = c.dim, H = c.num_heads, hd = D / H, T = x.rows;
  for (size_t l = 0; l < layers.size(); ++l) {
    const MimiLayer& L = layers[l];
    float* kcl = s->kc + (size_t)l * s->max_batch * s->max_pos * D;
    float* vcl = s->vc + (size_t)l * s->max_batch * s->max_pos * D;
    MM_TRY(r.layernorm(x, n, L.n1w.p, L.n1b.p));
    MM_TRY(r.conv(L.in_proj, n, qkv, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    if (!r.dry) {
      MM_TRY(kk_launch_rope_append((float*)qkv.p, T, H, H, hd, s->rope, s->pos, kcl, vcl, s->max_pos, B, r.st));
      MM_TRY(kk_launch_attn_cache((const float*)qkv.p, T, H, H, hd, s->pos, kcl, vcl, s->max_pos, 1.0f / sqrtf((float)hd), (float*)att.p, 0, s->context, B, r.st));
    }
    MM_TRY(r.conv(L.out_proj, att, x, 0, 1, false, 1, 0, KK_ACT_NONE, &x, 0));
    MM_TRY(r.layernorm(x, n, L.n2w.p, L.n2b.p));
    MM_TRY(r.conv(L.lin1, n, hbuf, 0, 1, false, 1, 0, KK_ACT_GELU_TANH, nullptr, 0));
    MM_TRY(r.conv(L.lin2, hbuf, x, 0, 1, false, 1, 0, KK_ACT_NONE, &x, 0));
  }
  return 0;
}

int copy_rows(Run& r, const Act& src, int src_row0, const Act& dst, int rows) {
  if (r.dry) return 0;
  const KKLen len{nullptr, 0, rows};
  return kk_launch_copy_slice((const float*)src.p + (size_t)src_row0 * src.ld, src.bs(), src.ld, dst.p, dst.bs(), dst.ld, 0, dst.C, rows, len, r.B, KK_F32, r.st);
}

// lays the stream's state buffers out for its largest step (sizes only when pool == nullptr); returns the floats needed.  The per-item
// pitch of every buffer is fixed by max_chunk, so the carried rows stay where they are when the step size changes (stream_set_rows)
size_t stream_layout(kk_mimi_stream* s, float* pool) {
  const kk_mimi* m = s->m;
  const kk_mimi_config& c = m->cfg;
  const int us = c.upsample_stride, D = c.dim, mb = s->max_batch;
  size_t off = 0;
  auto place = [&](StateBuf& b, int S, int n, int C) {
    b.S = S; b.n = b.nmax = n; b.C = C; b.maxB = mb;
    b.p = pool ? pool + off : nullptr;
    off += (b.floats() + 63) & ~(size_t)63;
  };
  const int F = s->max_chunk;
  s->up.assign(c.n_ratios, StateBuf());
  s->blk.assign(c.n_ratios, StateBuf());
  if (!s->encoder) {
    place(s->resample, 1, F, D);                       // quantised frames: [previous | new]
    int rows = F * us;
    place(s->first, c.ksize - 1, rows, D);
    for (int l = 0; l < c.n_ratios; ++l) {
      const SeaLayer& L = m->sea[l];
      place(s->up[l], 1, rows, L.up.Cin);            // [previous input row | new rows] of the transposed conv
      rows *= L.ratio;
      place(s->blk[l], c.residual_ksize - 1, rows, L.up.Cout);
    }
    place(s->last, c.last_ksize - 1, rows, m->final_conv.Cin);
  } else {
    long long rows = (long long)F * kk_mimi_samples_per_frame(m);
    place(s->first, c.ksize - 1, (int)rows, 1);
    for (int l = 0; l < c.n_ratios; ++l) {
      const SeaLayer& L = m->enc_sea[l];
      place(s->blk[l], c.residual_ksize - 1, (int)rows, L.b0.Cin);
      place(s->up[l], L.up.K - L.ratio, (int)rows, L.up.Cin);  // the strided conv carries k - stride rows
      rows /= L.ratio;
    }
    place(s->last, c.last_ksize - 1, (int)rows, m->enc_final.Cin);
    place(s->resample, us, (int)rows, D);              // conv k = 2 us, stride us: carries us rows
  }
  return off;
}

// rows every buffer takes in a step of F code frames (the same recurrences as stream_layout); pointers and pitches do not move
void stream_set_rows(kk_mimi_stream* s, int F) {
  const kk_mimi* m = s->m;
  const kk_mimi_config& c = m->cfg;
  const int us = c.upsample_stride;
  s->chunk = F;
  if (!s->encoder) {
    s->resample.n = F;
    int rows = F * us;
    s->first.n = rows;
    for (int l = 0; l < c.n_ratios; ++l) {
      s->up[l].n = rows;
      rows *= m->sea[l].ratio;
      s->blk[l].n = rows;
    }
    s->last.n = rows;
  } else {
    long long rows = (long long)F * kk_mimi_samples_per_frame(m);
    s->first.n = (int)rows;
    for (int l = 0; l < c.n_ratios; ++l) {
      s->blk[l].n = (int)rows;
      s->up[l].n = (int)rows;
      rows /= m->enc_sea[l].ratio;
    }
    s->last.n = (int)rows;
    s->resample.n = (int)rows;
  }
}

int stream_begin(Run& r, kk_mimi_stream* s) {  // zero state on the first step after create / reset
  if (r.dry || !s->fresh) return 0;
  if (hipMemsetAsync(s->pool, 0, s->pool_floats * 4, r.st) != hipSuccess) return kk_fail("mimi stream: state reset failed");
  return 0;
}

// dense scratch for the ELU'd input of the few-rows kernel: the largest [S + n][C] block among the stream's state buffers (and the
// residual blocks' hidden rows, which are never wider than their input)
void stream_elu_scratch(Run& r, kk_mimi_stream* s) {
  size_t mx = 0;
  auto upd = [&](const StateBuf& b) { mx = std::max(mx, (size_t)(b.S + b.n) * b.C); };
  upd(s->first); upd(s->last); upd(s->resample);
  for (const auto& b : s->up) upd(b);
  for (const auto& b : s->blk) upd(b);
  r.elu_floats = (size_t)r.B * mx;
  r.elu_tmp = (float*)r.raw(r.elu_floats * 4);
}

int run_decode_step(Run& r, kk_mimi_stream* s, const int* codes, float* pcm_out) {
  kk_mimi* m = r.m;
  const kk_mimi_config& c = m->cfg;
  const int B = r.B, D = c.dim, Q = c.qdim, us = c.upsample_stride, F = s->chunk;
  Act q1 = r.act(F, Q), q2 = r.act(F, Q), xu = r.act((F + 1) * us, D), x = r.act(F * us, D), xup = r.act(F * us, D);
  Act n = r.act(F * us, D), qkv = r.act(F * us, 3 * D), att = r.act(F * us, D), hbuf = r.act(F * us, c.dim_feedforward);
  stream_elu_scratch(r, s);
  if (r.oom) return kk_fail("kk_mimi_decode_step: workspace too small");
  if (!r.dry && s->pos + F * us > s->max_pos) return kk_fail("kk_mimi_decode_step: the stream is longer than max_frames (kk_mimi_stream_create)");
  MM_TRY(stream_begin(r, s));
  // ---- quantizer.decode of the new frames, straight behind the previous one
  Act xq = s->resample.fresh();
  if (!r.dry) {
    hipLaunchKernelGGL(rvq_sum_kernel<float>, dim3(F, B), dim3(256), 0, r.st, codes, m->codebooks.p, c.nq, c.bins, Q, F, q1.ld, (float*)q1.p, (float*)q2.p);
    KK_CHECK_LAUNCH();
  }
  MM_TRY(r.conv(m->proj_first, q1, xq, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  if (c.nq > 1) MM_TRY(r.conv(m->proj_rest, q2, xq, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 1));
  // ---- upsample.step: the depthwise transposed conv over [previous | new]; its first `us` rows repeat the last step's
  if (!r.dry) {
    const Act w = s->resample.all();
    hipLaunchKernelGGL(upsample_dw_kernel<float>, dim3((F + 1) * us, B), dim3(256), 0, r.st, (const float*)w.p, w.bs(), m->up_w.p, D, w.ld, F + 1, us, (float*)xu.p);
    KK_CHECK_LAUNCH();
  }
  MM_TRY(copy_rows(r, xu, us, x, F * us));
  MM_TRY(copy_rows(r, x, 0, xup, F * us));  // debug hook: the transformer updates x in place
  if (!r.dry) MM_TRY(state_shift(s->resample, B, r.st));
  r.note("upsampled", xup);
  // ---- decoder_transformer with the KV caches
  MM_TRY(run_transformer_step(r, s, m->layers, x, n, qkv, att, hbuf));
  r.note("transformer", x);
  // ---- decoder.step (seanet.py:228-283 through each module's step)
  MM_TRY(copy_rows(r, x, 0, s->first.fresh(), F * us));
  MM_TRY(r.conv(m->init_conv, s->first.all(), s->up[0].fresh(), 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  if (!r.dry) MM_TRY(state_shift(s->first, B, r.st));
  static const char* lname[8] = {"layer0", "layer1", "layer2", "layer3", "layer4", "layer5", "layer6", "layer7"};
  for (size_t l = 0; l < m->sea.size(); ++l) {
    const SeaLayer& S = m->sea[l];
    const StateBuf& U = s->up[l];
    const StateBuf& Bk = s->blk[l];
    const StateBuf& nextb = l + 1 < m->sea.size() ? s->up[l + 1] : s->last;
    Act full = r.act((U.n + 1) * S.ratio, S.up.Cout), hb = r.act(Bk.n, S.b0.Cout);
    if (r.oom) return kk_fail("kk_mimi_decode_step: workspace too small");
    MM_TRY(r.conv(S.up, U.all(), full, 0, 1, true, S.ratio, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    MM_TRY(copy_rows(r, full, S.ratio, Bk.fresh(), Bk.n));
    if (!r.dry) MM_TRY(state_shift(U, B, r.st));
    MM_TRY(r.conv(S.b0, Bk.all(), hb, 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    const Act skip = Bk.fresh();
    MM_TRY(r.conv(S.b1, hb, nextb.fresh(), 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, &skip, 0));
    if (!r.dry) MM_TRY(state_shift(Bk, B, r.st));
    r.note(l < 8 ? lname[l] : "layerN", nextb.fresh());
  }
  Act out;
  out.p = pcm_out; out.rows = s->last.n; out.C = 1; out.ld = 1; out.dtype = KK_F32;
  MM_TRY(r.conv(m->final_conv, s->last.all(), out, 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
  if (!r.dry) {
    MM_TRY(state_shift(s->last, B, r.st));
    s->pos += F * us;
    s->frames += F;
    s->fresh = false;
  }
  return 0;
}

// Mimi.encode_step (mimi.py:156-161): pcm [B][chunk * samples_per_frame] -> codes [B][nq][chunk]
int run_encode_step(Run& r, kk_mimi_stream* s, const float* pcm, int* codes) {
  kk_mimi* m = r.m;
  const kk_mimi_config& c = m->cfg;
  const int B = r.B, D = c.dim, Q = c.qdim, us = c.upsample_stride, F = s->chunk, T = F * us;
  const int N = s->first.n;
  if (!r.dry && s->pos + T > s->max_pos) return kk_fail("kk_mimi_encode_step: the stream is longer than max_frames (kk_mimi_encode_stream_create)");
  stream_elu_scratch(r, s);
  if (r.oom) return kk_fail("kk_mimi_encode_step: workspace too small");
  MM_TRY(stream_begin(r, s));
  Act in;
  in.p = const_cast<float*>(pcm); in.rows = N; in.C = 1; in.ld = 1; in.dtype = KK_F32;
  MM_TRY(copy_rows(r, in, 0, s->first.fresh(), N));
  MM_TRY(r.conv(m->enc_init, s->first.all(), s->blk[0].fresh(), 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  if (!r.dry) MM_TRY(state_shift(s->first, B, r.st));
  for (size_t l = 0; l < m->enc_sea.size(); ++l) {
    const SeaLayer& S = m->enc_sea[l];
    const StateBuf& Bk = s->blk[l];
    const StateBuf& Dn = s->up[l];
    const StateBuf& nextb = l + 1 < m->enc_sea.size() ? s->blk[l + 1] : s->last;
    Act hb = r.act(Bk.n, S.b0.Cout);
    if (r.oom) return kk_fail("kk_mimi_encode_step: workspace too small");
    MM_TRY(r.conv(S.b0, Bk.all(), hb, 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    const Act skip = Bk.fresh();
    MM_TRY(r.conv(S.b1, hb, Dn.fresh(), 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, &skip, 0));
    if (!r.dry) MM_TRY(state_shift(Bk, B, r.st));
    MM_TRY(r.conv(S.up, Dn.all(), nextb.fresh(), 0, 1, false, S.ratio, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    if (!r.dry) MM_TRY(state_shift(Dn, B, r.st));
  }
  Act x = r.act(T, D), n = r.act(T, D), qkv = r.act(T, 3 * D), att = r.act(T, D), hbuf = r.act(T, c.dim_feedforward);
  Act xd = r.act(F, D), res = r.act(F, Q), dots = r.act(F, c.bins), xs = r.act(T, D);
  if (r.oom) return kk_fail("kk_mimi_encode_step: workspace too small");
  MM_TRY(r.conv(m->enc_final, s->last.all(), xs, 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
  if (!r.dry) MM_TRY(state_shift(s->last, B, r.st));
  r.note("seanet", xs);
  MM_TRY(copy_rows(r, xs, 0, x, T));  // (the transformer updates x in place; xs stays for the debug hook)
  MM_TRY(run_transformer_step(r, s, m->enc_layers, x, n, qkv, att, hbuf));
  r.note("transformer", x);
  // downsample.step: conv k = 2 us, stride us, 'edge' left padding on the first step
  MM_TRY(copy_rows(r, x, 0, s->resample.fresh(), T));
  if (!r.dry && s->fresh) {
    hipLaunchKernelGGL(state_edge_fill_kernel, dim3(B), dim3(256), 0, r.st, s->resample.p, s->resample.S, s->resample.n, s->resample.C, s->resample.pitch());
    KK_CHECK_LAUNCH();
  }
  MM_TRY(r.conv(m->enc_down, s->resample.all(), xd, 0, 1, false, us, 0, KK_ACT_NONE, nullptr, 0));
  if (!r.dry) MM_TRY(state_shift(s->resample, B, r.st));
  r.note("downsampled", xd);
  for (int i = 0; i < c.nq; ++i) {  // split RVQ search, as in run_encode
    if (i == 0) MM_TRY(r.conv(m->inproj_first, xd, res, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    if (i == 1) MM_TRY(r.conv(m->inproj_rest, xd, res, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    MM_TRY(r.conv(m->cb_dot[i], res, dots, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    if (!r.dry) {
      hipLaunchKernelGGL(rvq_argmin_kernel, dim3(F, B), dim3(256), 0, r.st, (const float*)dots.p, m->c2.p + (size_t)i * c.bins,
                         m->codebooks.p + (size_t)i * c.bins * Q, c.bins, Q, F, c.nq, i, (float*)res.p, codes);
      KK_CHECK_LAUNCH();
    }
  }
  if (!r.dry) {
    s->pos += T;
    s->frames += F;
    s->fresh = false;
  }
  return 0;
}

int mimi_encode_frames(const kk_mimi_config& c, int N) {
  long long L = N;
  for (int i = c.n_ratios - 1; i >= 0; --i) L = (L + c.ratios[i] - 1) / c.ratios[i];
  return (int)((L + c.upsample_stride - 1) / c.upsample_stride);
}

// Mimi.encode (mimi.py:138-145), always fp32: the code-book search is an argmin, so the arithmetic stays on the parity path
int run_encode(Run& r, int N, const float* pcm, int* codes) {
  kk_mimi* m = r.m;
  const kk_mimi_config& c = m->cfg;
  const int B = r.B, D = c.dim, Q = c.qdim;
  r.adt = KK_F32;
  Act x;
  x.p = const_cast<float*>(pcm); x.rows = N; x.C = 1; x.ld = 1; x.dtype = KK_F32;
  Act y = r.act(N, m->enc_init.Cout);
  if (r.oom) return kk_fail("kk_mimi_encode: workspace too small");
  MM_TRY(r.conv(m->enc_init, x, y, c.ksize - 1, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  for (size_t l = 0; l < m->enc_sea.size(); ++l) {
    const SeaLayer& S = m->enc_sea[l];
    Act hb = r.act(y.rows, S.b0.Cout), o = r.act(y.rows, y.C);
    const int Lo = kk_cdiv(y.rows, S.ratio);
    Act d = r.act(Lo, S.up.Cout);
    if (r.oom) return kk_fail("kk_mimi_encode: workspace too small");
    MM_TRY(r.conv(S.b0, y, hb, c.residual_ksize - 1, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    MM_TRY(r.conv(S.b1, hb, o, 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, &y, 0));
    // causal strided conv: left pad k - stride, the right "extra padding" (conv.py:200-209) is the kernel's implicit zero rows
    MM_TRY(r.conv(S.up, o, d, S.up.K - S.ratio, 1, false, S.ratio, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    y = d;
  }
  const int T = y.rows;
  Act xe = r.act(T, D), n = r.act(T, D), qkv = r.act(T, 3 * D), att = r.act(T, D), hbuf = r.act(T, c.dim_feedforward);
  if (r.oom) return kk_fail("kk_mimi_encode: workspace too small");
  MM_TRY(r.conv(m->enc_final, y, xe, c.last_ksize - 1, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
  r.note("seanet", xe);
  Act xt = r.act(T, D);
  if (r.oom) return kk_fail("kk_mimi_encode: workspace too small");
  if (!r.dry && hipMemcpyAsync(xt.p, xe.p, (size_t)B * T * D * 4, hipMemcpyDeviceToDevice, r.st) != hipSuccess) return kk_fail("kk_mimi_encode: copy failed");
  MM_TRY(run_transformer(r, m->enc_layers, xt, n, qkv, att, hbuf));
  r.note("transformer", xt);
  // resampler: conv k = 2 s, stride s, 'edge' padding on both sides (ConvDownsample1d, conv.py:350-367)
  const int s = c.upsample_stride, k = 2 * s, Nf = kk_cdiv(T, s);
  const int left = k - s, Lp = (Nf - 1) * s + k;  // = left + T + extra
  Act xp = r.act(Lp, D), xd = r.act(Nf, D), res = r.act(Nf, Q), dots = r.act(Nf, c.bins);
  if (r.oom) return kk_fail("kk_mimi_encode: workspace too small");
  if (!r.dry) {
    hipLaunchKernelGGL(edge_pad_kernel, dim3(Lp, B), dim3(256), 0, r.st, (const float*)xt.p, T, D, left, Lp, (float*)xp.p);
    KK_CHECK_LAUNCH();
  }
  MM_TRY(r.conv(m->enc_down, xp, xd, 0, 1, false, s, 0, KK_ACT_NONE, nullptr, 0));
  r.note("downsampled", xd);
  // split RVQ search (quantization.py:128-133,170-176): first code book on its own projection, the other nq-1 on the second
  for (int i = 0; i < c.nq; ++i) {
    if (i == 0) MM_TRY(r.conv(m->inproj_first, xd, res, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    if (i == 1) MM_TRY(r.conv(m->inproj_rest, xd, res, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    MM_TRY(r.conv(m->cb_dot[i], res, dots, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    if (!r.dry) {
      hipLaunchKernelGGL(rvq_argmin_kernel, dim3(Nf, B), dim3(256), 0, r.st, (const float*)dots.p, m->c2.p + (size_t)i * c.bins,
                         m->codebooks.p + (size_t)i * c.bins * Q, c.bins, Q, Nf, c.nq, i, (float*)res.p, codes);
      KK_CHECK_LAUNCH();
    }
  }
  return 0;
}

int check_cfg(const kk_mimi_config& c) {
  if (c.dim <= 0 || c.dim % c.num_heads != 0 || c.dim / c.num_heads != 64) return kk_fail("kk_mimi_create: head size must be 64");
  if (c.nq < 1 || c.bins < 1 || c.qdim < 1 || c.num_layers < 0 || c.n_ratios < 1 || c.n_ratios > 8) return kk_fail("kk_mimi_create: bad configuration");
  if (c.upsample_stride < 1 || c.ksize < 1 || c.residual_ksize < 1 || c.last_ksize < 1 || c.compress < 1) return kk_fail("kk_mimi_create: bad configuration");
  for (int i = 0; i < c.n_ratios; ++i)
    if (c.ratios[i] < 1) return kk_fail("kk_mimi_create: bad ratio");
  return 0;
}

}  // namespace

extern "C" int kk_mimi_create(const kk_mimi_config* cfg, kk_mimi** out) {
  if (!cfg || !out) return kk_fail("kk_mimi_create: null argument");
  MM_TRY(check_cfg(*cfg));
  kk_mimi* m = new kk_mimi();
  m->cfg = *cfg;
  m->adt = cfg->compute_dtype == KK_BF16 ? KK_BF16 : KK_F32;
  *out = m;
  return 0;
}

extern "C" void kk_mimi_destroy(kk_mimi* m) {
  if (!m) return;
  if (m->dev) (void)hipFree(m->dev);
  delete m;
}

extern "C" int kk_mimi_load_tensor(kk_mimi* m, const char* name, const int64_t* shape, int ndim, const float* data) {
  if (!m || !name || !shape || !data || ndim < 1) return kk_fail("kk_mimi_load_tensor: bad argument");
  if (m->finalized) return kk_fail("kk_mimi_load_tensor: model already finalized");
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  m->host[name].assign(data, data + n);
  return 0;
}

extern "C" int kk_mimi_finalize(kk_mimi* m, void* stream) {
  if (!m) return kk_fail("kk_mimi_finalize: null model");
  if (m->finalized) return kk_fail("kk_mimi_finalize: already finalized");
  const kk_mimi_config& c = m->cfg;
  Packer P{m, ""};
  const int D = c.dim, Q = c.qdim;
  // code books: embedding = embedding_sum / max(cluster_usage, 1e-5), first | rest
  {
    m->codebooks.n = c.nq * c.bins * Q;
    m->codebooks.off = P.alloc((size_t)m->codebooks.n);
    for (int i = 0; i < c.nq; ++i) {
      const std::string p = std::string("quantizer.") + (i == 0 ? "rvq_first" : "rvq_rest") + ".vq.layers." + std::to_string(i == 0 ? 0 : i - 1) + ".codebook";
      const std::vector<float>* es = P.get(p + ".embedding_sum", (size_t)c.bins * Q);
      const std::vector<float>* cu = P.get(p + ".cluster_usage", (size_t)c.bins);
      if (!es || !cu) break;
      float* dst = &m->pack[m->codebooks.off + (size_t)i * c.bins * Q];
      for (int r = 0; r < c.bins; ++r) {
        const float u = (*cu)[r] > 1e-5f ? (*cu)[r] : 1e-5f;
        for (int q = 0; q < Q; ++q) dst[(size_t)r * Q + q] = (*es)[(size_t)r * Q + q] / u;
      }
    }
  }
  m->proj_first = P.conv("quantizer.rvq_first.output_proj.weight", "", D, 1, Q);
  if (c.nq > 1) m->proj_rest = P.conv("quantizer.rvq_rest.output_proj.weight", "", D, 1, Q);
  m->up_w = P.vec("upsample.convtr.convtr.convtr.weight", (size_t)2 * c.upsample_stride * D);  // [1][2s][D] == [2s][D]
  {
    const int half = 32;
    std::vector<float> f(half);
    for (int i = 0; i < half; ++i) f[i] = (float)pow((double)c.rope_base, -(double)i / (double)half);
    m->inv_freq.n = half;
    m->inv_freq.off = P.alloc(half);
    memcpy(&m->pack[m->inv_freq.off], f.data(), half * 4);
  }
  auto pack_transformer = [&](const std::string& stack, std::vector<MimiLayer>& layers) {
    layers.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
      const std::string p = stack + ".transformer.layers." + std::to_string(l);
      MimiLayer& L = layers[l];
      L.n1w = P.vec(p + ".norm1.weight", D); L.n1b = P.vec(p + ".norm1.bias", D);
      L.n2w = P.vec(p + ".norm2.weight", D); L.n2b = P.vec(p + ".norm2.bias", D);
      L.in_proj = P.conv(p + ".self_attn.in_proj.weight", "", 3 * D, 1, D);
      const std::vector<float>* s1 = P.get(p + ".layer_scale_1.scale", D);
      const std::vector<float>* s2 = P.get(p + ".layer_scale_2.scale", D);
      L.out_proj = P.conv(p + ".self_attn.out_proj.weight", "", D, 1, D, s1);
      L.lin1 = P.conv(p + ".gating.linear1.weight", "", c.dim_feedforward, 1, D);
      L.lin2 = P.conv(p + ".gating.linear2.weight", "", D, 1, c.dim_feedforward, s2);
    }
  };
  pack_transformer("decoder_transformer", m->layers);
  // ---- encode side, when the checkpoint has it
  m->has_encoder = m->host.count("encoder.init_conv1d.conv.conv.weight") != 0;
  if (m->has_encoder) {
    const int saved = m->adt;
    m->adt = KK_F32;  // the encoder runs on the fp32 kernels only: no fragment packs
    int em = 1;
    m->enc_init = P.conv("encoder.init_conv1d.conv.conv.weight", "encoder.init_conv1d.conv.conv.bias", c.nfilters, c.ksize, 1);
    m->enc_sea.resize(c.n_ratios);
    for (int l = 0; l < c.n_ratios; ++l) {
      const std::string p = "encoder.layers." + std::to_string(l);
      const int dim = em * c.nfilters, hid = dim / c.compress;
      SeaLayer& S = m->enc_sea[l];
      S.ratio = c.ratios[c.n_ratios - 1 - l];  // reversed(cfg.ratios), seanet.py:187
      S.b0 = P.conv(p + ".residuals.0.block.0.conv.conv.weight", p + ".residuals.0.block.0.conv.conv.bias", hid, c.residual_ksize, dim);
      S.b1 = P.conv(p + ".residuals.0.block.1.conv.conv.weight", p + ".residuals.0.block.1.conv.conv.bias", dim, 1, hid);
      S.up = P.conv(p + ".downsample.conv.conv.weight", p + ".downsample.conv.conv.bias", 2 * dim, 2 * S.ratio, dim);
      em *= 2;
    }
    m->enc_final = P.conv("encoder.final_conv1d.conv.conv.weight", "encoder.final_conv1d.conv.conv.bias", D, c.last_ksize, em * c.nfilters);
    m->enc_down = P.conv("downsample.conv.conv.conv.weight", "", D, 2 * c.upsample_stride, D);
    m->inproj_first = P.conv("quantizer.rvq_first.input_proj.weight", "", Q, 1, D);
    if (c.nq > 1) m->inproj_rest = P.conv("quantizer.rvq_rest.input_proj.weight", "", Q, 1, D);
    pack_transformer("encoder_transformer", m->enc_layers);
    // distance tables: c2 = |e|^2 / 2 and E^T as a 1x1 conv (quantization.py:27-28,35-39)
    m->c2.n = c.nq * c.bins;
    m->c2.off = P.alloc((size_t)m->c2.n);
    m->cb_dot.resize(c.nq);
    for (int i = 0; i < c.nq; ++i) {
      PackedConv& cd = m->cb_dot[i];
      cd.Cin = Q; cd.Cout = c.bins; cd.K = 1; cd.ldw = rup(c.bins, 64);
      cd.w_off = P.alloc((size_t)Q * cd.ldw);
      const float* E = &m->pack[m->codebooks.off + (size_t)i * c.bins * Q];
      float* dst = &m->pack[cd.w_off];
      float* c2 = &m->pack[m->c2.off + (size_t)i * c.bins];
      for (int r = 0; r < c.bins; ++r) {
        float ss = 0.f;
        for (int q = 0; q < Q; ++q) {
          const float e = E[(size_t)r * Q + q];
          dst[(size_t)q * cd.ldw + r] = e;
          ss += e * e;
        }
        c2[r] = ss / 2.0f;
      }
    }
    m->adt = saved;
  }
  int mult = 1 << c.n_ratios;
  m->init_conv = P.conv("decoder.init_conv1d.conv.conv.weight", "decoder.init_conv1d.conv.conv.bias", mult * c.nfilters, c.ksize, D);
  m->sea.resize(c.n_ratios);
  for (int l = 0; l < c.n_ratios; ++l) {
    const std::string p = "decoder.layers." + std::to_string(l);
    const int cin = mult * c.nfilters, cout = cin / 2, hid = cout / c.compress;
    SeaLayer& S = m->sea[l];
    S.ratio = c.ratios[l];
    S.up = P.conv(p + ".upsample.convtr.convtr.weight", p + ".upsample.convtr.convtr.bias", cout, 2 * S.ratio, cin);
    S.b0 = P.conv(p + ".residuals.0.block.0.conv.conv.weight", p + ".residuals.0.block.0.conv.conv.bias", hid, c.residual_ksize, cout);
    S.b1 = P.conv(p + ".residuals.0.block.1.conv.conv.weight", p + ".residuals.0.block.1.conv.conv.bias", cout, 1, hid);
    mult /= 2;
  }
  m->final_conv = P.conv("decoder.final_conv1d.conv.conv.weight", "decoder.final_conv1d.conv.conv.bias", 1, c.last_ksize, c.nfilters);
  if (!P.err.empty()) return kk_fail(("kk_mimi_finalize: " + P.err).c_str());
  if (hipMalloc((void**)&m->dev, m->pack.size() * sizeof(float)) != hipSuccess) return kk_fail("kk_mimi_finalize: hipMalloc failed");
  if (hipMemcpyAsync(m->dev, m->pack.data(), m->pack.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return kk_fail("kk_mimi_finalize: upload failed");
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return kk_fail("kk_mimi_finalize: stream sync failed");
  resolve(m, m->codebooks); resolve(m, m->inv_freq); resolve(m, m->up_w);
  resolve(m, m->proj_first); resolve(m, m->proj_rest); resolve(m, m->init_conv); resolve(m, m->final_conv);
  for (auto& L : m->layers) {
    resolve(m, L.n1w); resolve(m, L.n1b); resolve(m, L.n2w); resolve(m, L.n2b);
    resolve(m, L.in_proj); resolve(m, L.out_proj); resolve(m, L.lin1); resolve(m, L.lin2);
  }
  for (auto& S : m->sea) { resolve(m, S.up); resolve(m, S.b0); resolve(m, S.b1); }
  if (m->has_encoder) {
    resolve(m, m->enc_init); resolve(m, m->enc_final); resolve(m, m->enc_down); resolve(m, m->inproj_first); resolve(m, m->inproj_rest);
    resolve(m, m->c2);
    for (auto& S : m->enc_sea) { resolve(m, S.up); resolve(m, S.b0); resolve(m, S.b1); }
    for (auto& L : m->enc_layers) {
      resolve(m, L.n1w); resolve(m, L.n1b); resolve(m, L.n2w); resolve(m, L.n2b);
      resolve(m, L.in_proj); resolve(m, L.out_proj); resolve(m, L.lin1); resolve(m, L.lin2);
    }
    for (auto& cd : m->cb_dot) resolve(m, cd);
  }
  m->host.clear();
  std::vector<float>().swap(m->pack);
  m->finalized = true;
  return 0;
}

extern "C" int64_t kk_mimi_samples_per_frame(const kk_mimi* m) {
  if (!m) return 0;
  int64_t s = m->cfg.upsample_stride;
  for (int i = 0; i < m->cfg.n_ratios; ++i) s *= m->cfg.ratios[i];
  return s;
}

extern "C" size_t kk_mimi_workspace_bytes(kk_mimi* m, int B, int Nf) {
  if (!m || !m->finalized || B <= 0 || Nf <= 0) return 0;
  Run r{m, nullptr, B, nullptr, 0, 0, true, false};
  if (run_decode(r, Nf, nullptr, nullptr) != 0) return 0;
  return r.used + 256;
}

extern "C" int kk_mimi_decode(kk_mimi* m, void* stream, int B, int Nf, const int32_t* codes, void* workspace, size_t workspace_bytes,
                              float* pcm_out) {
  if (!m || !m->finalized) return kk_fail("kk_mimi_decode: model not finalized");
  if (B <= 0 || Nf <= 0 || !codes || !workspace || !pcm_out) return kk_fail("kk_mimi_decode: bad argument");
  if (workspace_bytes < kk_mimi_workspace_bytes(m, B, Nf)) return kk_fail("kk_mimi_decode: workspace too small");
  { std::lock_guard<std::mutex> lk(m->dbg_mu); m->dbg.clear(); }
  Run r{m, (hipStream_t)stream, B, (char*)workspace, workspace_bytes, 0, false, false};
  return run_decode(r, Nf, codes, pcm_out);
}

// ---- streaming (Mimi.decode_step / encode_step / MimiStreamingDecoder, mimi.py:156-168,264-306)
static int kk_fail(const char* who, const char* what) {
  std::string msg = std::string(who) + what;
  return kk_fail(msg.c_str());
}
static int stream_create(kk_mimi* m, bool encoder, int max_batch, int max_frames, int chunk_frames, kk_mimi_stream** out, const char* who) {
  if (!m || !m->finalized || !out || max_batch < 1 || max_frames < 1 || chunk_frames < 1 || chunk_frames > max_frames) return kk_fail(who, ": bad argument");
  if (encoder && !m->has_encoder) return kk_fail(who, ": the checkpoint held no encoder.* parameters");
  const kk_mimi_config& c = m->cfg;
  const int hd = c.dim / c.num_heads;
  if (hd != 64 && hd != 128) return kk_fail(who, ": head size must be 64 or 128");
  for (int l = 0; l < c.n_ratios; ++l) {
    const SeaLayer& L = encoder ? m->enc_sea[l] : m->sea[l];
    if (L.up.K != 2 * L.ratio) return kk_fail(who, ": (transposed) conv kernel must be twice its stride");
  }
  kk_mimi_stream* s = new (std::nothrow) kk_mimi_stream();
  if (!s) return kk_fail(who, ": out of memory");
  s->m = m; s->max_batch = max_batch; s->max_pos = max_frames * c.upsample_stride; s->encoder = encoder; s->chunk = s->max_chunk = chunk_frames;
  const size_t D = c.dim, nl = encoder ? m->enc_layers.size() : m->layers.size(), kvn = nl * max_batch * s->max_pos * D * 4;
  std::vector<float> tab((size_t)s->max_pos * (hd / 2) * 2);
  for (int p = 0; p < s->max_pos; ++p)
    for (int i = 0; i < hd / 2; ++i) {
      const float ang = (float)p * (float)pow((double)c.rope_base, -(double)i / (double)(hd / 2));  // the offline rope_kernel's angle
      tab[((size_t)p * (hd / 2) + i) * 2] = cosf(ang);
      tab[((size_t)p * (hd / 2) + i) * 2 + 1] = sinf(ang);
    }
  s->pool_floats = stream_layout(s, nullptr);
  if (hipMalloc((void**)&s->pool, s->pool_floats * 4) != hipSuccess || hipMalloc((void**)&s->kc, kvn ? kvn : 4) != hipSuccess ||
      hipMalloc((void**)&s->vc, kvn ? kvn : 4) != hipSuccess || hipMalloc((void**)&s->rope, tab.size() * 4) != hipSuccess ||
      hipMemcpy(s->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    kk_mimi_stream_destroy(s);
    return kk_fail(who, ": hipMalloc failed");
  }
  (void)stream_layout(s, s->pool);
  *out = s;
  return 0;
}
extern "C" int kk_mimi_stream_create(kk_mimi* m, int max_batch, int max_frames, kk_mimi_stream** out) {
  return stream_create(m, false, max_batch, max_frames, 1, out, "kk_mimi_stream_create");
}
// chunk_frames = the LARGEST number of code frames one kk_mimi_decode_step / kk_mimi_encode_step call will carry (and the size of the first
// ones); kk_mimi_stream_set_chunk changes the size of the following steps, the state carries over
extern "C" int kk_mimi_stream_create_chunked(kk_mimi* m, int encoder, int max_batch, int max_frames, int chunk_frames, kk_mimi_stream** out) {
  return stream_create(m, encoder != 0, max_batch, max_frames, chunk_frames, out, "kk_mimi_stream_create_chunked");
}
// The reference's step functions take any number of frames per call and CONTINUE the conv / KV state (mimi.py:156-168; conv.py:265-351):
// the next steps carry `chunk_frames` code frames (1 .. the stream's largest).  The positions of one step see each other in the
// transformer (no mask, transformer.py:79-104), so the step size is part of the result, as in the reference.
extern "C" int kk_mimi_stream_set_chunk(kk_mimi_stream* s, int chunk_frames) {
  if (!s || chunk_frames < 1) return kk_fail("kk_mimi_stream_set_chunk: bad argument");
  if (chunk_frames > s->max_chunk) return kk_fail("kk_mimi_stream_set_chunk: larger than the chunk_frames the stream was created for");
  stream_set_rows(s, chunk_frames);
  return 0;
}
extern "C" int kk_mimi_stream_max_chunk_frames(const kk_mimi_stream* s) { return s ? s->max_chunk : -1; }
extern "C" void kk_mimi_stream_destroy(kk_mimi_stream* s) {
  if (!s) return;
  for (float* p : {s->pool, s->kc, s->vc, s->rope})
    if (p) (void)hipFree(p);
  delete s;
}
extern "C" int kk_mimi_stream_reset(kk_mimi_stream* s) {  // MimiStreamingDecoder.reset / Mimi.reset_state (mimi.py:131-137,274-279)
  if (!s) return kk_fail("kk_mimi_stream_reset: null stream");
  s->frames = s->pos = 0;
  s->B = 0;
  s->fresh = true;  // the state buffers are zeroed on the next step's stream
  return 0;
}
extern "C" int kk_mimi_stream_frames(const kk_mimi_stream* s) { return s ? s->frames : -1; }
extern "C" int kk_mimi_stream_chunk_frames(const kk_mimi_stream* s) { return s ? s->chunk : -1; }
// TransformerConfig.context (mimi.py:55-77: 250 for mimi_202407): cached positions a step may look back on.  Only between resets.
extern "C" int kk_mimi_stream_set_context(kk_mimi_stream* s, int context) {
  if (!s || context < 0) return kk_fail("kk_mimi_stream_set_context: bad argument");
  if (s->frames != 0) return kk_fail("kk_mimi_stream_set_context: only on a fresh or reset stream");
  s->context = context;
  return 0;
}
extern "C" size_t kk_mimi_stream_workspace_bytes(kk_mimi_stream* s, int B) {
  if (!s || B < 1 || B > s->max_batch) return 0;
  Run r{s->m, nullptr, B, nullptr, 0, 0, true, false};
  r.adt = KK_F32;
  const int cur = s->chunk;
  stream_set_rows(s, s->max_chunk);  // sized for the largest step: one workspace serves every step size
  const int rc = s->encoder ? run_encode_step(r, s, nullptr, nullptr) : run_decode_step(r, s, nullptr, nullptr);
  stream_set_rows(s, cur);
  if (rc != 0) return 0;
  return r.used + 256;
}
static int step_check(kk_mimi_stream* s, bool encoder, int B, const void* in, void* workspace, size_t workspace_bytes, void* out, const char* who) {
  if (!s || !in || !workspace || !out || B < 1 || B > s->max_batch) return kk_fail(who, ": bad argument");
  if (s->encoder != encoder) return kk_fail(who, ": the stream was created for the other direction");
  if (s->frames > 0 && B != s->B) return kk_fail(who, ": the batch size of a stream is fixed until it is reset");
  if (workspace_bytes < kk_mimi_stream_workspace_bytes(s, B)) return kk_fail(who, ": workspace too small");
  return 0;
}
// chunk frames of codes [B][nq][chunk] int32 -> pcm [B][chunk * samples_per_frame] float32; the stream's state lives in `s`
// (library-owned device memory).  B is fixed by the first step after create / reset.  fp32 kernels.
extern "C" int kk_mimi_decode_step(kk_mimi_stream* s, void* stream, int B, const int32_t* codes, void* workspace, size_t workspace_bytes, float* pcm_out) {
  MM_TRY(step_check(s, false, B, codes, workspace, workspace_bytes, pcm_out, "kk_mimi_decode_step"));
  s->B = B;
  { std::lock_guard<std::mutex> lk(s->m->dbg_mu); s->m->dbg.clear(); }
  Run r{s->m, (hipStream_t)stream, B, (char*)workspace, workspace_bytes, 0, false, false};
  r.adt = KK_F32;
  return run_decode_step(r, s, codes, pcm_out);
}
// Mimi.encode_step (mimi.py:156-161): pcm [B][chunk * samples_per_frame] float32 -> codes [B][nq][chunk] int32
extern "C" int kk_mimi_encode_step(kk_mimi_stream* s, void* stream, int B, const float* pcm, void* workspace, size_t workspace_bytes, int32_t* codes_out) {
  MM_TRY(step_check(s, true, B, pcm, workspace, workspace_bytes, codes_out, "kk_mimi_encode_step"));
  s->B = B;
  { std::lock_guard<std::mutex> lk(s->m->dbg_mu); s->m->dbg.clear(); }
  Run r{s->m, (hipStream_t)stream, B, (char*)workspace, workspace_bytes, 0, false, false};
  r.adt = KK_F32;
  return run_encode_step(r, s, pcm, codes_out);
}

extern "C" int kk_mimi_encode_frames(const kk_mimi* m, int N) { return (m && N > 0) ? mimi_encode_frames(m->cfg, N) : 0; }

extern "C" size_t kk_mimi_encode_workspace_bytes(kk_mimi* m, int B, int N) {
  if (!m || !m->finalized || !m->has_encoder || B <= 0 || N <= 0) return 0;
  Run r{m, nullptr, B, nullptr, 0, 0, true, false};
  if (run_encode(r, N, nullptr, nullptr) != 0) return 0;
  return r.used + 256;
}

extern "C" int kk_mimi_encode(kk_mimi* m, void* stream, int B, int N, const float* pcm, void* workspace, size_t workspace_bytes,
                              int32_t* codes_out) {
  if (!m || !m->finalized) return kk_fail("kk_mimi_encode: model not finalized");
  if (!m->has_encoder) return kk_fail("kk_mimi_encode: the checkpoint held no encoder.* parameters");
  if (B <= 0 || N <= 0 || !pcm || !workspace || !codes_out) return kk_fail("kk_mimi_encode: bad argument");
  if (workspace_bytes < kk_mimi_encode_workspace_bytes(m, B, N)) return kk_fail("kk_mimi_encode: workspace too small");
  { std::lock_guard<std::mutex> lk(m->dbg_mu); m->dbg.clear(); }
  Run r{m, (hipStream_t)stream, B, (char*)workspace, workspace_bytes, 0, false, false};
  return run_encode(r, N, pcm, codes_out);
}

// named intermediates of the LAST decode call (tests): "quantized", "upsampled", "transformer", "layer0".."layer3"
extern "C" int kk_mimi_debug_info(kk_mimi* m, const char* name, int64_t* rows, int64_t* channels) {
  if (!m || !name) return kk_fail("kk_mimi_debug_info: bad argument");
  std::lock_guard<std::mutex> lk(m->dbg_mu);
  auto it = m->dbg.find(name);
  if (it == m->dbg.end()) return kk_fail("kk_mimi_debug_info: unknown stage");
  if (rows) *rows = it->second.rows;
  if (channels) *channels = it->second.C;
  return 0;
}
extern "C" int kk_mimi_debug_fetch(kk_mimi* m, void* stream, const char* name, float* dst) {
  if (!m || !name || !dst) return kk_fail("kk_mimi_debug_fetch: bad argument");
  std::lock_guard<std::mutex> lk(m->dbg_mu);
  auto it = m->dbg.find(name);
  if (it == m->dbg.end()) return kk_fail("kk_mimi_debug_fetch: unknown stage");
  const DebugBuf& d = it->second;
  // any dtype / pitch -> dense fp32 [B][rows][C]
  if (kk_launch_convert(d.p, d.dtype, d.bs, d.ld, dst, KK_F32, (long long)d.rows * d.C, d.C, d.C, d.rows, d.B, (hipStream_t)stream) != 0) return -1;
  return 0;
}
