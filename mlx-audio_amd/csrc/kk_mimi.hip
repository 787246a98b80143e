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

#include <map>
#include <string>
#include <vector>

#include "../../include/kokoro_hip.h"
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

struct PackedConv {  // generic-kernel pack [K][Cin][ldw] fp32 (+ bias)
  size_t w_off = 0, b_off = 0;
  bool has_bias = false;
  int Cin = 0, Cout = 0, K = 0, ldw = 0;
  const float* w = nullptr;
  const float* b = nullptr;
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
  const float* p;
  int rows, C, ld;
  long long bs;
  int B;
};

}  // namespace

struct kk_mimi {
  kk_mimi_config cfg;
  std::map<std::string, std::vector<float>> host;
  std::vector<float> pack;
  float* dev = nullptr;
  bool finalized = false;
  PackedVec codebooks;  // [nq][bins][qdim]
  PackedVec inv_freq;   // [32]
  PackedVec up_w;       // [2*stride][dim]
  PackedConv proj_first, proj_rest, init_conv, final_conv;
  std::vector<MimiLayer> layers;
  std::vector<SeaLayer> sea;
  std::map<std::string, DebugBuf> dbg;
};

namespace {

// ------------------------------------------------------------------------------------------------------------- kernels
// one workgroup per (frame, item): column c of the first code book's row, and of the sum of the other rows in ascending
// code-book order (the reference's accumulation order, quantization.py:97-101)
__global__ __launch_bounds__(256) void rvq_sum_kernel(const int* codes, const float* cb, int nq, int bins, int qdim, int Nf, float* q_first,
                                                      float* q_rest) {
  const int t = blockIdx.x, b = blockIdx.y;
  for (int c = threadIdx.x; c < qdim; c += blockDim.x) {
    const int* cd = codes + (long long)b * nq * Nf + t;
    int id = cd[0];
    id = id < 0 ? 0 : (id >= bins ? bins - 1 : id);
    q_first[((long long)b * Nf + t) * qdim + c] = cb[(long long)id * qdim + c];
    float s = 0.f;
    for (int i = 1; i < nq; ++i) {
      int idi = cd[(long long)i * Nf];
      idi = idi < 0 ? 0 : (idi >= bins ? bins - 1 : idi);
      const float e = cb[((long long)i * bins + idi) * qdim + c];
      s = i == 1 ? e : s + e;
    }
    q_rest[((long long)b * Nf + t) * qdim + c] = s;
  }
}

// depth-wise transposed conv, kernel 2*s, stride s, causal (last s outputs dropped): output row p = s*t + j gets
// x[t] w[j] + x[t-1] w[j + s]
__global__ __launch_bounds__(256) void upsample_dw_kernel(const float* x, const float* w, int C, int Lin, int s, float* out) {
  const int p = blockIdx.x, b = blockIdx.y;
  const int tt = p / s, j = p - tt * s;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = x[((long long)b * Lin + tt) * C + c] * w[(long long)j * C + c];
    if (tt > 0) v += x[((long long)b * Lin + tt - 1) * C + c] * w[(long long)(j + s) * C + c];
    out[((long long)b * Lin * s + p) * C + c] = v;
  }
}

// nn.RoPE(traditional): pairs (2i, 2i+1) of every q and k head rotated by pos * inv_freq[i]; qkv [B][T][3*D] in place
__global__ __launch_bounds__(256) void rope_kernel(float* qkv, const float* inv_freq, int T, int D, int hd) {
  const int t = blockIdx.x, b = blockIdx.y;
  float* row = qkv + ((long long)b * T + t) * 3 * D;
  const int half = hd / 2, npairs = D / 2;
  for (int e = threadIdx.x; e < 2 * npairs; e += blockDim.x) {
    const int which = e / npairs, pr = e - which * npairs;  // q (0) or k (1)
    const int h = pr / half, i = pr - h * half;
    const float ang = (float)t * inv_freq[i];
    const float c = cosf(ang), s = sinf(ang);
    float* p2 = row + which * D + h * hd + 2 * i;
    const float x0 = p2[0], x1 = p2[1];
    p2[0] = x0 * c - x1 * s;
    p2[1] = x0 * s + x1 * c;
  }
}

// ------------------------------------------------------------------------------------------------------------- host
int rup(int v, int m) { return (v + m - 1) / m * m; }

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
    if (!bname.empty()) {
      const std::vector<float>* b = get(bname, (size_t)O);
      if (!b) return c;
      c.has_bias = true;
      c.b_off = alloc(O);
      memcpy(&m->pack[c.b_off], b->data(), (size_t)O * 4);
    }
    return c;
  }
};

void resolve(kk_mimi* m, PackedConv& c) {
  c.w = m->dev + c.w_off;
  c.b = c.has_bias ? m->dev + c.b_off : nullptr;
}
void resolve(kk_mimi* m, PackedVec& v) { v.p = v.n ? m->dev + v.off : nullptr; }

struct Run {
  kk_mimi* m;
  hipStream_t st;
  int B;
  char* base;
  size_t cap, used;
  bool dry;
  float* f32(size_t n) {
    const size_t off = (used + 255) & ~(size_t)255;
    used = off + n * 4;
    if (dry) return nullptr;
    return used <= cap ? (float*)(base + off) : nullptr;
  }
  // conv / transposed conv / linear on [B][L][C] fp32 buffers (pitch = channel count)
  int conv(const PackedConv& w, const float* x, int Lin, float* out, int Lout, int pad, int dil, bool transposed, int stride, int in_act,
           int act, const float* res, int accumulate) {
    if (dry) return 0;
    KKConvArgs a;
    memset(&a, 0, sizeof a);
    a.x = x; a.xbs = (long long)Lin * w.Cin; a.ldx = w.Cin;
    a.w = w.w; a.ldw = w.ldw; a.bias = w.b;
    a.out = out; a.obs = (long long)Lout * w.Cout; a.ldo = w.Cout;
    if (res) { a.res = res; a.rbs = a.obs; a.ldr = w.Cout; }
    a.Cin = w.Cin; a.Cout = w.Cout; a.Kw = w.K;
    a.mode = transposed ? KK_CONVT : KK_CONV; a.stride = stride; a.pad = pad; a.dil = dil;
    a.Q = transposed ? kk_cdiv(Lout, stride) : Lout; a.Lo_rows = Lout;
    a.lin = KKLen{nullptr, 0, Lin}; a.lout = KKLen{nullptr, 0, Lout};
    a.in_slope = 1.f; a.scale = 1.f; a.accumulate = accumulate; a.act = act; a.in_act = in_act;
    return kk_launch_conv_generic(a, B, KK_F32, KK_F32, st);
  }
  int layernorm(const float* x, float* out, int C, int L, const float* w, const float* b) {
    if (dry) return 0;
    KKLnArgs a;
    memset(&a, 0, sizeof a);
    a.x = x; a.xbs = (long long)L * C; a.ldx = C; a.out = out; a.obs = a.xbs; a.ldo = C; a.C = C; a.Lmax = L;
    a.len = KKLen{nullptr, 0, L}; a.w = w; a.bias = b; a.eps = 1e-5f; a.act = KK_ACT_NONE;
    return kk_launch_layernorm(a, B, KK_F32, st);
  }
  void note(const char* name, const float* p, int rows, int C) {
    if (!dry) m->dbg[name] = DebugBuf{p, rows, C, C, (long long)rows * C, B};
  }
};

#define MM_TRY(x)        \
  do {                   \
    const int rc__ = (x); \
    if (rc__ != 0) return rc__; \
  } while (0)

int run_decode(Run& r, int Nf, const int* codes, float* pcm) {
  kk_mimi* m = r.m;
  const kk_mimi_config& c = m->cfg;
  const int B = r.B, D = c.dim, Q = c.qdim, T = Nf * c.upsample_stride;
  // ---- split RVQ decode
  float* q1 = r.f32((size_t)B * Nf * Q);
  float* q2 = r.f32((size_t)B * Nf * Q);
  float* x0 = r.f32((size_t)B * Nf * D);
  float* x = r.f32((size_t)B * T * D);
  if (!r.dry && (!q1 || !q2 || !x0 || !x)) return kk_fail("kk_mimi_decode: workspace too small");
  if (!r.dry) {
    hipLaunchKernelGGL(rvq_sum_kernel, dim3(Nf, B), dim3(256), 0, r.st, codes, m->codebooks.p, c.nq, c.bins, Q, Nf, q1, q2);
    KK_CHECK_LAUNCH();
  }
  MM_TRY(r.conv(m->proj_first, q1, Nf, x0, Nf, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  if (c.nq > 1) MM_TRY(r.conv(m->proj_rest, q2, Nf, x0, Nf, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 1));
  r.note("quantized", x0, Nf, D);
  float* xu = r.f32((size_t)B * T * D);  // kept for the debug hook: the transformer updates x in place
  if (!r.dry) {
    if (!xu) return kk_fail("kk_mimi_decode: workspace too small");
    hipLaunchKernelGGL(upsample_dw_kernel, dim3(T, B), dim3(256), 0, r.st, x0, m->up_w.p, D, Nf, c.upsample_stride, xu);
    KK_CHECK_LAUNCH();
    if (hipMemcpyAsync(x, xu, (size_t)B * T * D * 4, hipMemcpyDeviceToDevice, r.st) != hipSuccess) return kk_fail("kk_mimi_decode: copy failed");
  }
  r.note("upsampled", xu, T, D);
  // ---- transformer
  float* n = r.f32((size_t)B * T * D);
  float* qkv = r.f32((size_t)B * T * 3 * D);
  float* att = r.f32((size_t)B * T * D);
  float* hbuf = r.f32((size_t)B * T * c.dim_feedforward);
  if (!r.dry && (!n || !qkv || !att || !hbuf)) return kk_fail("kk_mimi_decode: workspace too small");
  for (int l = 0; l < c.num_layers; ++l) {
    const MimiLayer& L = m->layers[l];
    MM_TRY(r.layernorm(x, n, D, T, L.n1w.p, L.n1b.p));
    MM_TRY(r.conv(L.in_proj, n, T, qkv, T, 0, 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
    if (!r.dry) {
      hipLaunchKernelGGL(rope_kernel, dim3(T, B), dim3(256), 0, r.st, qkv, m->inv_freq.p, T, D, D / c.num_heads);
      KK_CHECK_LAUNCH();
      KKAttnArgs a;
      memset(&a, 0, sizeof a);
      a.qkv = qkv; a.bs = (long long)T * 3 * D; a.ld = 3 * D; a.out = att; a.obs = (long long)T * D; a.ldo = D;
      a.heads = c.num_heads; a.hs = D; a.Tmax = T; a.len = KKLen{nullptr, 0, T}; a.scale = 1.0f / sqrtf((float)(D / c.num_heads));
      MM_TRY(kk_launch_attention(a, B, KK_F32, r.st));
    }
    MM_TRY(r.conv(L.out_proj, att, T, x, T, 0, 1, false, 1, 0, KK_ACT_NONE, x, 0));  // x += ls1 * (W att)
    MM_TRY(r.layernorm(x, n, D, T, L.n2w.p, L.n2b.p));
    MM_TRY(r.conv(L.lin1, n, T, hbuf, T, 0, 1, false, 1, 0, KK_ACT_GELU_TANH, nullptr, 0));
    MM_TRY(r.conv(L.lin2, hbuf, T, x, T, 0, 1, false, 1, 0, KK_ACT_NONE, x, 0));     // x += ls2 * (W2 gelu(W1 n))
  }
  r.note("transformer", x, T, D);
  // ---- SEANet decoder
  int Lc = T, Cc = m->init_conv.Cout;
  float* y = r.f32((size_t)B * Lc * Cc);
  if (!r.dry && !y) return kk_fail("kk_mimi_decode: workspace too small");
  MM_TRY(r.conv(m->init_conv, x, T, y, T, (c.ksize - 1), 1, false, 1, 0, KK_ACT_NONE, nullptr, 0));
  static const char* lname[8] = {"layer0", "layer1", "layer2", "layer3", "layer4", "layer5", "layer6", "layer7"};
  for (size_t l = 0; l < m->sea.size(); ++l) {
    const SeaLayer& S = m->sea[l];
    const int Lo = Lc * S.ratio, Co = S.up.Cout;
    float* u = r.f32((size_t)B * Lo * Co);
    float* hb = r.f32((size_t)B * Lo * S.b0.Cout);
    float* o = r.f32((size_t)B * Lo * Co);
    if (!r.dry && (!u || !hb || !o)) return kk_fail("kk_mimi_decode: workspace too small");
    MM_TRY(r.conv(S.up, y, Lc, u, Lo, 0, 1, true, S.ratio, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    MM_TRY(r.conv(S.b0, u, Lo, hb, Lo, (c.residual_ksize - 1), 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
    MM_TRY(r.conv(S.b1, hb, Lo, o, Lo, 0, 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, u, 0));
    r.note(l < 8 ? lname[l] : "layerN", o, Lo, Co);
    y = o; Lc = Lo; Cc = Co;
  }
  MM_TRY(r.conv(m->final_conv, y, Lc, pcm, Lc, (c.last_ksize - 1), 1, false, 1, KK_ACT_ELU, KK_ACT_NONE, nullptr, 0));
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
  m->layers.resize(c.num_layers);
  for (int l = 0; l < c.num_layers; ++l) {
    const std::string p = "decoder_transformer.transformer.layers." + std::to_string(l);
    MimiLayer& L = m->layers[l];
    L.n1w = P.vec(p + ".norm1.weight", D); L.n1b = P.vec(p + ".norm1.bias", D);
    L.n2w = P.vec(p + ".norm2.weight", D); L.n2b = P.vec(p + ".norm2.bias", D);
    L.in_proj = P.conv(p + ".self_attn.in_proj.weight", "", 3 * D, 1, D);
    const std::vector<float>* s1 = P.get(p + ".layer_scale_1.scale", D);
    const std::vector<float>* s2 = P.get(p + ".layer_scale_2.scale", D);
    L.out_proj = P.conv(p + ".self_attn.out_proj.weight", "", D, 1, D, s1);
    L.lin1 = P.conv(p + ".gating.linear1.weight", "", c.dim_feedforward, 1, D);
    L.lin2 = P.conv(p + ".gating.linear2.weight", "", D, 1, c.dim_feedforward, s2);
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
  Run r{m, nullptr, B, nullptr, 0, 0, true};
  if (run_decode(r, Nf, nullptr, nullptr) != 0) return 0;
  return r.used + 256;
}

extern "C" int kk_mimi_decode(kk_mimi* m, void* stream, int B, int Nf, const int32_t* codes, void* workspace, size_t workspace_bytes,
                              float* pcm_out) {
  if (!m || !m->finalized) return kk_fail("kk_mimi_decode: model not finalized");
  if (B <= 0 || Nf <= 0 || !codes || !workspace || !pcm_out) return kk_fail("kk_mimi_decode: bad argument");
  if (workspace_bytes < kk_mimi_workspace_bytes(m, B, Nf)) return kk_fail("kk_mimi_decode: workspace too small");
  m->dbg.clear();
  Run r{m, (hipStream_t)stream, B, (char*)workspace, workspace_bytes, 0, false};
  return run_decode(r, Nf, codes, pcm_out);
}

// named intermediates of the LAST decode call (tests): "quantized", "upsampled", "transformer", "layer0".."layer3"
extern "C" int kk_mimi_debug_info(kk_mimi* m, const char* name, int64_t* rows, int64_t* channels) {
  if (!m || !name) return kk_fail("kk_mimi_debug_info: bad argument");
  auto it = m->dbg.find(name);
  if (it == m->dbg.end()) return kk_fail("kk_mimi_debug_info: unknown stage");
  if (rows) *rows = it->second.rows;
  if (channels) *channels = it->second.C;
  return 0;
}
extern "C" int kk_mimi_debug_fetch(kk_mimi* m, void* stream, const char* name, float* dst) {
  if (!m || !name || !dst) return kk_fail("kk_mimi_debug_fetch: bad argument");
  auto it = m->dbg.find(name);
  if (it == m->dbg.end()) return kk_fail("kk_mimi_debug_fetch: unknown stage");
  const DebugBuf& d = it->second;
  if (hipMemcpyAsync(dst, d.p, (size_t)d.B * d.rows * d.C * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
    return kk_fail("kk_mimi_debug_fetch: copy failed");
  return 0;
}
