// Host side of libkokoro_hip.so: weight intake, weight-norm folding / packing, the forward
// orchestration of kokoro.py:120-170 as a sequence of kernel launches on the caller's stream,
// and the C ABI declared in include/kokoro_hip.h.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/kokoro_hip.h"
#include "kk_common.h"
#include "kk_kernels.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static const void* g_op_wfrag = nullptr;  // kk_debug_set_op_wfrag: fragment-order weights for the single-kernel conv entry points
static int g_op_variant = 4;              // kk_debug_set_op_variant: which fragment-order kernel they use (4 or 5)

int kk_fail(const char* msg) {
  g_err = msg ? msg : "unknown error";
  return -1;
}
static int failf(const char* fmt, const std::string& a) {
  char buf[512];
  snprintf(buf, sizeof buf, fmt, a.c_str());
  return kk_fail(buf);
}
extern "C" const char* kk_last_error(void) { return g_err.c_str(); }
extern "C" int kk_abi_version(void) { return KK_ABI_VERSION; }

#define KK_TRY(x)            \
  do {                       \
    int rc__ = (x);          \
    if (rc__ != 0) return rc__; \
  } while (0)

// ------------------------------------------------------------------------------------------------
// model structures
// ------------------------------------------------------------------------------------------------
namespace {

struct HostTensor {
  std::vector<float> d;
  std::vector<int64_t> shape;
};

struct ConvW {  // packed [Kw][Cin][ldw] fp32 + bias
  size_t w_off = 0, b_off = 0;
  bool has_bias = false;
  int Cin = 0, Cout = 0, Kw = 0, ldw = 0;
  const float* w = nullptr;
  const float* b = nullptr;
  // bf16 MFMA pack [Kw][CoutP][CinP] (bf16 mode, eligible layers only); bias is then padded to CoutP
  bool mfma = false;
  size_t wb_off = 0, wf_off = 0;
  int CinP = 0, CoutP = 0, Cout8 = 0;
  const bf16_t* wb = nullptr;
  const bf16_t* wf = nullptr;  // the same weights in MFMA fragment order (variant-4 kernel, kk_mfma4_pack_index)
  size_t wl_off = 0;           // k = 1, Cin % 32 == 0: the weights in the streaming Linear kernel's fragment order (kk_linear_rows.hip), 0 = none
  const bf16_t* wl = nullptr;
  // MX-fp8 pack (kk_set_quantization, bf16 mode, the reference's quantised layer set only): e4m3 fragments + E8M0 scale bytes
  bool fp8 = false;
  size_t q8_off = 0, s8_off = 0;
  const uint4* q8 = nullptr;
  const unsigned char* s8 = nullptr;
};
struct VecW {
  size_t off = 0;
  int n = 0;
  const float* p = nullptr;
};
struct LstmW {
  ConvW in;  // [1][I][8H], bias = b_ih + b_hh, columns dir*4H + gate row
  VecW whT;  // [2][H][4H]
  int H = 0;
  // bf16 mode, H = 256: Wh as bf16 [2][4H][H] for the on-chip recurrence kernel
  bool has_whb = false;
  size_t whb_off = 0;
  const void* whb = nullptr;
};
struct AdainRef {
  size_t off = 0;  // offset of gamma in the style vector of its half; beta at off + C
  int C = 0;
};
struct ResBlk1d {  // AdainResBlk1d, istftnet.py:825-899
  int Cin = 0, Cout = 0;
  bool up = false, learned = false;
  AdainRef n1, n2;
  ConvW conv1, conv2, sc;
  VecW pool_w, pool_b;
};
struct ResBlock1 {  // AdaINResBlock1, istftnet.py:341-396
  int C = 0, k = 0;
  int dil[3] = {1, 3, 5};
  ConvW c1[3], c2[3];
  AdainRef a1[3], a2[3];
  VecW al1[3], al2[3];
};

struct DebugEntry {
  void* p;
  int ld;
  long long bs;
  int rows, C, dtype, B;
};

}  // namespace

struct kk_model {
  kk_config cfg;
  std::map<std::string, HostTensor> host;
  bool finalized = false;
  std::vector<float> pack;  // host staging of all packed fp32 parameters
  float* dev = nullptr;     // device copy of `pack`
  int adt = KK_F32;         // activation dtype

  // Albert
  VecW emb_word, emb_pos, emb_type, emb_ln_w, emb_ln_b;
  ConvW map_in, qkv, att_dense, ffn, ffn_out, bert_encoder;
  VecW att_ln_w, att_ln_b, full_ln_w, full_ln_b;
  // predictor
  std::vector<LstmW> dur_lstms;
  std::vector<AdainRef> dur_adaln;
  LstmW pred_lstm, shared_lstm, text_lstm;
  VecW dur_W, dur_b;
  ResBlk1d f0blk[3], nblk[3];
  ConvW f0_proj, n_proj;
  // text encoder
  VecW te_emb;
  std::vector<ConvW> te_cnn;
  std::vector<VecW> te_ln_w, te_ln_b;
  // decoder
  ResBlk1d enc, dec[4];
  ConvW f0_conv, n_conv, asr_res;
  // generator
  VecW lin_w;
  float lin_b = 0.f;
  ConvW noise_conv[4], ups[4], conv_post;
  ConvW noise_conv_rows[4];  // bf16 mode: the strided noise convs re-expressed as stride-1 convs over row groups (Packer::strided_rows)
  int noise_rows_pad[4] = {0, 0, 0, 0};
  ResBlock1 noise_res[4];
  std::vector<ResBlock1> resblocks;
  // style projections (all AdaIN / AdaLN fc's), one matrix per style half
  VecW sp_wT, sp_b, sd_wT, sd_b;  // prosody half (ref_s[128:]) / decoder half (ref_s[:128])
  int Np = 0, Nd = 0;

  size_t head_wf_off = 0;       // conv_post in the fused head's fragment order (bf16), 0 = not eligible
  const bf16_t* head_wf = nullptr;
  int q_group = 0;      // kk_set_quantization: group size of the MLX affine quantisation the checkpoint went through (0 = none)
};

// Everything a forward MUTATES lives here, not in the model: the cache of captured graphs, the side stream and its fork / join events, the debug
// hooks and switches, the profile brackets.  A kk_model is immutable after kk_finalize and is shared by any number of contexts; one context serves
// one stream / thread at a time ("a kk_model is immutable after finalize and may be shared by threads using distinct streams + workspaces").
struct kk_context {
  kk_model* m = nullptr;
  std::map<std::string, DebugEntry> dbg;
  std::map<std::string, const float*> dbg_over;

  // optional per-kernel-class timing (kk_profile_*): HIP events around every launch of a class
  bool force_generic = false;  // tests: run the bf16 mode without the MFMA kernel
  bool no_fusion = false;      // tests: MFMA convs but stand-alone statistics / AdaIN kernels
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;  // pairs
  // graph replay of kk_forward (kk_set_graph_mode): one instantiated hipGraph per distinct argument tuple
  struct GraphEntry {
    std::vector<unsigned long long> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int seen = 0;  // 1 = ran eagerly once (first-launch attribute calls are done), 2 = captured
  };
  bool graph_mode = false;
  std::vector<GraphEntry> graphs;
  unsigned long long* seed_dev = nullptr;  // the Philox seed of a replayed forward
  bool capturing = false;
  bool no_v4 = false;
  bool no_fp8 = false;  // tests: quantised model, but the Q1 layer set runs on the bf16 kernel with the same (dequantised) weights
  int v5_mode = 0;              // conv variant 5 (wave-specialised, persistent): 0 = the layers where it wins (>= 9 taps: +1.1 % on the step), 1 = wherever eligible
                                // (tests / A-B), 2 = never
  bool no_head_fusion = false;  // tests / A-B: stand-alone conv_post + iSTFT head kernels instead of the fused head (kk_head.hip)
  bool keep_debug = false;      // tests: also materialise the tensors fused kernels skip (conv_post)
  hipStream_t cap_stream = nullptr;
  // side stream of a forward: branches that do not depend on each other (TextEncoder beside Albert / the duration stack; the harmonic source
  // beside the decoder) run concurrently -- the B = 1 latency is a chain of small kernels.  Fork / join through events, also inside a capture.
  hipStream_t side_stream = nullptr;
  hipEvent_t side_fork[2] = {nullptr, nullptr}, side_join[2] = {nullptr, nullptr};
  bool no_side = false;  // debug bit 8 of kk_debug_force_generic: everything on the caller's stream
  int linrows_mode = 0;  // debug bits 9 / 10: the streaming Linear kernel never / at every size (default: up to KK_LINROWS_MAX rows in flight)
  struct ProfRec { int cls; double flops; double bytes; };
  std::vector<ProfRec> prof_rec;
};

// ------------------------------------------------------------------------------------------------
// create / load
// ------------------------------------------------------------------------------------------------
extern "C" int kk_create(const kk_config* cfg, kk_model** out) {
  if (!cfg || !out) return kk_fail("kk_create: null argument");
  if (cfg->n_upsamples != 2 || cfg->n_resblock_kernels < 1 || cfg->n_resblock_kernels > 4)
    return kk_fail("kk_create: unsupported generator geometry");
  if (cfg->plbert_hidden != cfg->plbert_heads * 64) return kk_fail("kk_create: Albert head size must be 64");
  if (cfg->style_dim != 128) return kk_fail("kk_create: style_dim must be 128 (kokoro.py:145,165 split ref_s at 128)");
  if (cfg->hidden_dim % 32 != 0 || cfg->hidden_dim / 2 > 256) return kk_fail("kk_create: hidden_dim must be a multiple of 32, <= 512");
  if (cfg->gen_istft_n_fft != 20 || cfg->gen_istft_hop_size != 5) return kk_fail("kk_create: iSTFT head is built for n_fft 20 / hop 5");
  if (cfg->compute_dtype != KK_DTYPE_F32 && cfg->compute_dtype != KK_DTYPE_BF16) return kk_fail("kk_create: compute_dtype");
  kk_model* m = new (std::nothrow) kk_model();
  if (!m) return kk_fail("kk_create: out of memory");
  m->cfg = *cfg;
  m->adt = cfg->compute_dtype == KK_DTYPE_BF16 ? KK_BF16 : KK_F32;
  *out = m;
  return 0;
}

extern "C" void kk_destroy(kk_model* m) {
  if (!m) return;
  if (m->dev) (void)hipFree(m->dev);
  delete m;
}

// A context on a finalized model: the side stream + fork / join events of a forward (Ctx::begin_side) are created here, never inside a capture.
extern "C" int kk_context_create(kk_model* m, kk_context** out) {
  if (!m || !out) return kk_fail("kk_context_create: null argument");
  if (!m->finalized) return kk_fail("kk_context_create: kk_finalize has not been called");
  kk_context* cx = new (std::nothrow) kk_context();
  if (!cx) return kk_fail("kk_context_create: out of memory");
  cx->m = m;
  bool ok = hipStreamCreateWithFlags(&cx->side_stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i)
    ok = hipEventCreateWithFlags(&cx->side_fork[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&cx->side_join[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    kk_context_destroy(cx);
    return kk_fail("kk_context_create: hipStreamCreate / hipEventCreate failed");
  }
  if (getenv("KK_NO_SIDE")) cx->no_side = true;  // (A/B timing / debugging: no side stream)
  *out = cx;
  return 0;
}

extern "C" void kk_context_destroy(kk_context* cx) {
  if (!cx) return;
  for (hipEvent_t e : cx->prof_ev) (void)hipEventDestroy(e);
  for (auto& g : cx->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  if (cx->seed_dev) (void)hipFree(cx->seed_dev);
  if (cx->cap_stream) (void)hipStreamDestroy(cx->cap_stream);
  if (cx->side_stream) (void)hipStreamDestroy(cx->side_stream);
  for (int i = 0; i < 2; ++i) {
    if (cx->side_fork[i]) (void)hipEventDestroy(cx->side_fork[i]);
    if (cx->side_join[i]) (void)hipEventDestroy(cx->side_join[i]);
  }
  delete cx;
}

extern "C" kk_model* kk_context_model(kk_context* cx) { return cx ? cx->m : nullptr; }

static float bf16_to_f32(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float f16_to_f32(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, f = h & 1023;
  uint32_t u;
  if (e == 0) {
    if (f == 0) u = s << 31;
    else {
      int ee = -1;
      uint32_t ff = f;
      while (!(ff & 1024)) { ff <<= 1; ++ee; }
      u = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((ff & 1023) << 13);
    }
  } else if (e == 31) u = (s << 31) | 0x7F800000u | (f << 13);
  else u = (s << 31) | ((e - 15 + 127) << 23) | (f << 13);
  float r;
  memcpy(&r, &u, 4);
  return r;
}

static uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static std::string normalise_name(const std::string& in) {
  // PyTorch-side names -> MLX-side names (kokoro.py:24-44, 189-196)
  static const std::pair<const char*, const char*> lstm_map[] = {
      {"weight_ih_l0_reverse", "Wx_backward"}, {"weight_hh_l0_reverse", "Wh_backward"}, {"bias_ih_l0_reverse", "bias_ih_backward"},
      {"bias_hh_l0_reverse", "bias_hh_backward"}, {"weight_ih_l0", "Wx_forward"},       {"weight_hh_l0", "Wh_forward"},
      {"bias_ih_l0", "bias_ih_forward"},          {"bias_hh_l0", "bias_hh_forward"}};
  const size_t dot = in.rfind('.');
  if (dot == std::string::npos) return in;
  const std::string base = in.substr(0, dot), leaf = in.substr(dot + 1);
  for (auto& kv : lstm_map)
    if (leaf == kv.first) return base + "." + kv.second;
  if (in.rfind("text_encoder.", 0) == 0) {
    if (leaf == "gamma") return base + ".weight";
    if (leaf == "beta") return base + ".bias";
  }
  return in;
}

extern "C" int kk_load_tensor(kk_model* m, const char* name, int dtype, const int64_t* shape, int ndim, const void* data) {
  if (!m || !name || !data || ndim < 0 || ndim > 4) return kk_fail("kk_load_tensor: bad argument");
  if (m->finalized) return kk_fail("kk_load_tensor: model already finalized");
  std::string nm = normalise_name(name);
  if (nm.find("position_ids") != std::string::npos) return 0;  // dropped by sanitize (kokoro.py:177-179)
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return kk_fail("kk_load_tensor: bad shape");
    t.shape.push_back(shape[i]);
    n *= (size_t)shape[i];
  }
  t.d.resize(n);
  if (dtype == KK_DTYPE_F32) memcpy(t.d.data(), data, n * 4);
  else if (dtype == KK_DTYPE_BF16) for (size_t i = 0; i < n; ++i) t.d[i] = bf16_to_f32(((const uint16_t*)data)[i]);
  else if (dtype == KK_DTYPE_F16) for (size_t i = 0; i < n; ++i) t.d[i] = f16_to_f32(((const uint16_t*)data)[i]);
  else return kk_fail("kk_load_tensor: dtype must be F32, BF16 or F16");
  m->host[nm] = std::move(t);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// finalize: fold weight-norm, pack, upload
// ------------------------------------------------------------------------------------------------
namespace {

struct Packer {
  kk_model* m;
  std::string err;
  bool ok() const { return err.empty(); }
  size_t alloc(size_t n) {
    size_t off = (m->pack.size() + 63) & ~(size_t)63;  // 256-byte alignment
    m->pack.resize(off + n, 0.f);
    return off;
  }
  const HostTensor* get(const std::string& name) {
    auto it = m->host.find(name);
    if (it == m->host.end()) {
      if (err.empty()) err = "missing parameter: " + name;
      return nullptr;
    }
    return &it->second;
  }
  // 3-D conv weight in MLX layout [A][K][C]; accepts the PyTorch layout [A][C][K] as well.
  bool conv3(const std::string& name, int A, int K, int C, std::vector<float>& out) {
    const HostTensor* t = get(name);
    if (!t) return false;
    out.assign((size_t)A * K * C, 0.f);
    if (t->shape.size() == 3 && t->shape[0] == A && t->shape[1] == K && t->shape[2] == C) {
      out = t->d;
      return true;
    }
    if (t->shape.size() == 3 && t->shape[0] == A && t->shape[1] == C && t->shape[2] == K) {
      for (int a = 0; a < A; ++a)
        for (int c = 0; c < C; ++c)
          for (int k = 0; k < K; ++k) out[((size_t)a * K + k) * C + c] = t->d[((size_t)a * C + c) * K + k];
      return true;
    }
    if (err.empty()) err = "unexpected shape for " + name;
    return false;
  }
  bool vec(const std::string& name, size_t n, std::vector<float>& out) {
    const HostTensor* t = get(name);
    if (!t) return false;
    if (t->d.size() != n) {
      if (err.empty()) err = "unexpected size for " + name;
      return false;
    }
    out = t->d;
    return true;
  }
  VecW put(const std::vector<float>& v) {
    VecW r;
    r.n = (int)v.size();
    r.off = alloc(v.size());
    memcpy(&m->pack[r.off], v.data(), v.size() * 4);
    return r;
  }
  VecW put_named(const std::string& name, size_t n) {
    std::vector<float> v;
    if (!vec(name, n, v)) return VecW();
    return put(v);
  }
  // weight_norm (istftnet.py:53-93 with dim=0): per leading index a, w = g[a] * v[a] / (||v[a]||_2 + 1e-7)
  bool folded(const std::string& prefix, int A, int K, int C, std::vector<float>& w) {
    std::vector<float> g;
    if (!conv3(prefix + ".weight_v", A, K, C, w)) return false;
    if (!vec(prefix + ".weight_g", (size_t)A, g)) return false;
    for (int a = 0; a < A; ++a) {
      float ss = 0.f;
      float* row = &w[(size_t)a * K * C];
      for (int i = 0; i < K * C; ++i) ss += row[i] * row[i];
      const float nrm = sqrtf(ss) + 1e-7f;
      for (int i = 0; i < K * C; ++i) row[i] = row[i] / nrm * g[a];
    }
    return true;
  }
  // bf16 MFMA pack of wsrc[o][k][i] -> [k][CoutP][CinP] (two bf16 per float slot of the staging vector)
  void pack_mfma(ConvW& c, const std::vector<float>& wsrc, int O, int K, int I) {
    // Cout is rounded up to 8 for the kernel (zero weights / bias): only used when the destination has room (Ctx::can_mfma)
    if (m->adt != KK_BF16 || O < 16 || I < 16) return;
    c.mfma = true;
    c.Cout8 = kk_cdiv(O, 8) * 8;
    c.CinP = kk_cdiv(I, 64) * 64;
    c.CoutP = kk_cdiv(O, 128) * 128;
    const size_t nel = (size_t)K * c.CoutP * c.CinP;
    c.wb_off = alloc((nel + 1) / 2);  // resize() zero-fills
    c.wf_off = alloc((nel + 1) / 2);  // (may move the staging vector: take the pointers afterwards)
    uint16_t* dst = (uint16_t*)&m->pack[c.wb_off];
    uint16_t* dfr = (uint16_t*)&m->pack[c.wf_off];
    for (int o = 0; o < O; ++o)
      for (int k = 0; k < K; ++k)
        for (int i = 0; i < I; ++i) {
          const uint16_t v = f32_to_bf16_rne(wsrc[((size_t)o * K + k) * I + i]);
          dst[((size_t)k * c.CoutP + o) * c.CinP + i] = v;
          dfr[kk_mfma4_pack_index(k, o, i, c.CoutP, c.CinP)] = v;
        }
    if (K == 1 && I % 32 == 0) {  // Linear layers: the streaming kernel's pack (columns padded to 16 with zeros)
      const size_t nl = (size_t)kk_cdiv(O, 16) * 16 * I;
      c.wl_off = alloc((nl + 1) / 2);
      uint16_t* dl = (uint16_t*)&m->pack[c.wl_off];
      for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i) dl[kk_linear_pack_index(o, i, I)] = f32_to_bf16_rne(wsrc[(size_t)o * I + i]);
    }
  }
  // MX-fp8 pack of a Linear weight wsrc[o][i] (the layer set of the reference's quantisation predicate, tts/utils.py:241-260):
  // e4m3 with one power-of-two scale per `q_group` inputs, in MFMA fragment order (kk_mxfp8.hip)
  void pack_fp8(ConvW& c, const std::vector<float>& wsrc, int O, int I) {
    if (!m->q_group || m->adt != KK_BF16 || !kk_mxfp8_eligible(I, O) || I % m->q_group) return;
    const size_t qb = kk_mxfp8_q_bytes(O, I), sb = kk_mxfp8_s_bytes(O, I);
    c.q8_off = alloc((qb + 3) / 4);
    c.s8_off = alloc((sb + 3) / 4);
    if (kk_mxfp8_pack_weight_host(wsrc.data(), O, I, m->q_group, (unsigned char*)&m->pack[c.q8_off], (unsigned char*)&m->pack[c.s8_off]) == 0)
      c.fp8 = true;
  }
  // pack a conv weight given as wsrc[o][k][i] into [k][i][ldw]
  ConvW pack_oki(const std::vector<float>& wsrc, int O, int K, int I, const std::vector<float>* bias) {
    ConvW c;
    c.Cin = I; c.Cout = O; c.Kw = K; c.ldw = kk_cdiv(O, 64) * 64;
    c.w_off = alloc((size_t)K * I * c.ldw);
    float* dst = &m->pack[c.w_off];
    for (int o = 0; o < O; ++o)
      for (int k = 0; k < K; ++k)
        for (int i = 0; i < I; ++i) dst[((size_t)k * I + i) * c.ldw + o] = wsrc[((size_t)o * K + k) * I + i];
    pack_mfma(c, wsrc, O, K, I);
    if (bias) {
      c.has_bias = true;
      const int nb = c.mfma ? c.CoutP : O;
      c.b_off = alloc(nb);
      memcpy(&m->pack[c.b_off], bias->data(), (size_t)O * 4);
    }
    return c;
  }
  // ConvWeighted used as conv1d: weight_v [O][K][I]
  ConvW convw(const std::string& prefix, int O, int K, int I, bool bias) {
    std::vector<float> w, b;
    if (!folded(prefix, O, K, I, w)) return ConvW();
    if (bias && !vec(prefix + ".bias", (size_t)O, b)) return ConvW();
    return pack_oki(w, O, K, I, bias ? &b : nullptr);
  }
  // ConvWeighted used as conv_transpose1d (Generator.ups, istftnet.py:725-734,161-166): weight_v [Cin][K][Cout], norm over
  // each Cin slice, bias [Cout]
  ConvW convw_t(const std::string& prefix, int Cin, int K, int Cout) {
    std::vector<float> w, b;
    if (!folded(prefix, Cin, K, Cout, w)) return ConvW();
    if (!vec(prefix + ".bias", (size_t)Cout, b)) return ConvW();
    // re-index [i][k][o] as [o][k][i]: the packed forms ([k][i][o] fp32, [k][o][i] bf16) are the same for both kinds of conv
    std::vector<float> woki((size_t)Cout * K * Cin);
    for (int i = 0; i < Cin; ++i)
      for (int k = 0; k < K; ++k)
        for (int o = 0; o < Cout; ++o) woki[((size_t)o * K + k) * Cin + i] = w[((size_t)i * K + k) * Cout + o];
    return pack_oki(woki, Cout, K, Cin, &b);
  }
  // nn.Conv1d: weight [O][K][I] (+ PyTorch [O][I][K]), bias [O]
  ConvW conv_plain(const std::string& prefix, int O, int K, int I) {
    std::vector<float> w, b;
    if (!conv3(prefix + ".weight", O, K, I, w)) return ConvW();
    if (!vec(prefix + ".bias", (size_t)O, b)) return ConvW();
    return pack_oki(w, O, K, I, &b);
  }
  // A strided conv (stride s, kernel K, padding pad) over rows of pitch `ld` reads, for output row q, the input rows
  // s*q - pad .. s*q - pad + K - 1: whole or partial GROUPS of s consecutive rows.  Viewing the input as rows of s*ld
  // channels turns it into a stride-1 conv with K2 = tau_max - tau_min + 1 taps (tau = floor((t - pad) / s)) whose weights are
  // the original taps scattered into [tau][co][j*ld + c] (zero elsewhere) -- which the MFMA kernel can run (noise_convs,
  // istftnet.py:744-752: Conv1d(22, 256, k=12, stride=6, padding=3) becomes 3 taps x 384 channels).  bf16 pack only.
  ConvW strided_rows(const std::string& prefix, int O, int K, int I, int s, int pad, int ld, int* pad2) {
    ConvW c;
    if (m->adt != KK_BF16) return c;
    std::vector<float> w, b;
    if (!conv3(prefix + ".weight", O, K, I, w)) return ConvW();
    if (!vec(prefix + ".bias", (size_t)O, b)) return ConvW();
    auto fdiv = [](int a, int d) { return a >= 0 ? a / d : -((-a + d - 1) / d); };
    const int tmin = fdiv(-pad, s), tmax = fdiv(K - 1 - pad, s);
    const int K2 = tmax - tmin + 1, I2 = s * ld;
    *pad2 = -tmin;
    c.Cin = I2; c.Cout = O; c.Kw = K2; c.ldw = 0;
    c.mfma = true;
    c.Cout8 = kk_cdiv(O, 8) * 8;
    c.CinP = kk_cdiv(I2, 64) * 64;
    c.CoutP = kk_cdiv(O, 128) * 128;
    const size_t nel = (size_t)K2 * c.CoutP * c.CinP;
    c.wb_off = alloc((nel + 1) / 2);
    c.wf_off = alloc((nel + 1) / 2);
    uint16_t* dst = (uint16_t*)&m->pack[c.wb_off];
    uint16_t* dfr = (uint16_t*)&m->pack[c.wf_off];
    for (int o = 0; o < O; ++o)
      for (int t = 0; t < K; ++t) {
        const int tau = fdiv(t - pad, s), j = (t - pad) - tau * s;
        for (int i = 0; i < I; ++i) {
          const uint16_t v = f32_to_bf16_rne(w[((size_t)o * K + t) * I + i]);
          dst[((size_t)(tau - tmin) * c.CoutP + o) * c.CinP + j * ld + i] = v;
          dfr[kk_mfma4_pack_index(tau - tmin, o, j * ld + i, c.CoutP, c.CinP)] = v;
        }
      }
    c.has_bias = true;
    c.b_off = alloc(c.CoutP);
    memcpy(&m->pack[c.b_off], b.data(), (size_t)O * 4);
    return c;
  }
  // nn.Linear: weight [O][I], bias [O]
  ConvW linear(const std::string& prefix, int O, int I, bool quantised = false) {
    std::vector<float> w, b;
    if (!vec(prefix + ".weight", (size_t)O * I, w)) return ConvW();
    if (!vec(prefix + ".bias", (size_t)O, b)) return ConvW();
    ConvW c = pack_oki(w, O, 1, I, &b);
    if (quantised) pack_fp8(c, w, O, I);
    return c;
  }
  LstmW lstm(const std::string& prefix, int I, int H) {
    LstmW l;
    l.H = H;
    const int G = 4 * H;
    std::vector<float> w((size_t)2 * G * I), b((size_t)2 * G), whT((size_t)2 * H * G);
    const char* dirs[2] = {"forward", "backward"};
    for (int d = 0; d < 2; ++d) {
      std::vector<float> wx, wh, bi, bh;
      if (!vec(prefix + ".Wx_" + dirs[d], (size_t)G * I, wx)) return l;
      if (!vec(prefix + ".Wh_" + dirs[d], (size_t)G * H, wh)) return l;
      if (!vec(prefix + ".bias_ih_" + dirs[d], (size_t)G, bi)) return l;
      if (!vec(prefix + ".bias_hh_" + dirs[d], (size_t)G, bh)) return l;
      memcpy(&w[(size_t)d * G * I], wx.data(), wx.size() * 4);
      for (int g = 0; g < G; ++g) b[(size_t)d * G + g] = bi[g] + bh[g];  // mx.addmm(b_ih + b_hh, ...) modules.py:156-158
      for (int g = 0; g < G; ++g)
        for (int j = 0; j < H; ++j) whT[((size_t)d * H + j) * G + g] = wh[(size_t)g * H + j];
    }
    l.in = pack_oki(w, 2 * G, 1, I, &b);
    l.whT = put(whT);
    if (m->adt == KK_BF16 && H == 256) {
      l.has_whb = true;
      l.whb_off = alloc(((size_t)2 * G * H + 1) / 2);
      uint16_t* dst = (uint16_t*)&m->pack[l.whb_off];
      for (int d = 0; d < 2; ++d)
        for (int g = 0; g < G; ++g)
          for (int j = 0; j < H; ++j) dst[((size_t)d * G + g) * H + j] = f32_to_bf16_rne(whT[((size_t)d * H + j) * G + g]);
    }
    return l;
  }
};

struct StyleBuilder {  // concatenates every `fc` of one style half into wT [128][N]
  std::vector<std::pair<std::string, int>> items;  // (prefix, C) : fc.weight [2C][128]
  AdainRef add(const std::string& prefix, int C) {
    AdainRef r;
    r.C = C;
    size_t off = 0;
    for (auto& it : items) off += 2 * (size_t)it.second;
    r.off = off;
    items.emplace_back(prefix, C);
    return r;
  }
  bool build(Packer& P, VecW& wT, VecW& bias, int& N) {
    size_t n = 0;
    for (auto& it : items) n += 2 * (size_t)it.second;
    N = (int)n;
    std::vector<float> W((size_t)128 * n), Bv(n);
    size_t off = 0;
    for (auto& it : items) {
      const int C2 = 2 * it.second;
      std::vector<float> w, b;
      if (!P.vec(it.first + ".fc.weight", (size_t)C2 * 128, w)) return false;
      if (!P.vec(it.first + ".fc.bias", (size_t)C2, b)) return false;
      for (int o = 0; o < C2; ++o) {
        Bv[off + o] = b[o];
        for (int j = 0; j < 128; ++j) W[(size_t)j * n + off + o] = w[(size_t)o * 128 + j];
      }
      off += C2;
    }
    wT = P.put(W);
    bias = P.put(Bv);
    return true;
  }
};

ResBlk1d build_resblk1d(Packer& P, StyleBuilder& S, const std::string& p, int Cin, int Cout, bool up) {
  ResBlk1d r;
  r.Cin = Cin; r.Cout = Cout; r.up = up; r.learned = Cin != Cout;
  r.n1 = S.add(p + ".norm1", Cin);
  r.n2 = S.add(p + ".norm2", Cout);
  r.conv1 = P.convw(p + ".conv1", Cout, 3, Cin, true);
  r.conv2 = P.convw(p + ".conv2", Cout, 3, Cout, true);
  if (r.learned) r.sc = P.convw(p + ".conv1x1", Cout, 1, Cin, false);
  if (up) {
    std::vector<float> w, b;
    if (P.folded(p + ".pool", Cin, 3, 1, w) && P.vec(p + ".pool.bias", (size_t)Cin, b)) {
      r.pool_w = P.put(w);  // [C][3]
      r.pool_b = P.put(b);
    }
  }
  return r;
}

ResBlock1 build_resblock1(Packer& P, StyleBuilder& S, const std::string& p, int C, int k, const int* dil) {
  ResBlock1 r;
  r.C = C; r.k = k;
  for (int j = 0; j < 3; ++j) {
    r.dil[j] = dil[j];
    const std::string js = std::to_string(j);
    r.c1[j] = P.convw(p + ".convs1." + js, C, k, C, true);
    r.c2[j] = P.convw(p + ".convs2." + js, C, k, C, true);
    r.a1[j] = S.add(p + ".adain1." + js, C);
    r.a2[j] = S.add(p + ".adain2." + js, C);
    r.al1[j] = P.put_named(p + ".alpha1." + js, (size_t)C);
    r.al2[j] = P.put_named(p + ".alpha2." + js, (size_t)C);
  }
  return r;
}

void resolve(kk_model* m, ConvW& c) {
  c.w = m->dev + c.w_off;
  c.b = c.has_bias ? m->dev + c.b_off : nullptr;
  c.wb = c.mfma ? (const bf16_t*)(m->dev + c.wb_off) : nullptr;
  c.wf = c.mfma ? (const bf16_t*)(m->dev + c.wf_off) : nullptr;
  c.wl = (c.mfma && c.wl_off) ? (const bf16_t*)(m->dev + c.wl_off) : nullptr;
  c.q8 = c.fp8 ? (const uint4*)(m->dev + c.q8_off) : nullptr;
  c.s8 = c.fp8 ? (const unsigned char*)(m->dev + c.s8_off) : nullptr;
}
void resolve(kk_model* m, VecW& v) { v.p = v.n ? m->dev + v.off : nullptr; }
void resolve(kk_model* m, LstmW& l) {
  resolve(m, l.in);
  resolve(m, l.whT);
  l.whb = l.has_whb ? (const void*)(m->dev + l.whb_off) : nullptr;
}
void resolve(kk_model* m, ResBlk1d& r) {
  resolve(m, r.conv1); resolve(m, r.conv2);
  if (r.learned) resolve(m, r.sc);
  resolve(m, r.pool_w); resolve(m, r.pool_b);
}
void resolve(kk_model* m, ResBlock1& r) {
  for (int j = 0; j < 3; ++j) {
    resolve(m, r.c1[j]); resolve(m, r.c2[j]); resolve(m, r.al1[j]); resolve(m, r.al2[j]);
  }
}

}  // namespace

extern "C" int kk_finalize(kk_model* m, void* stream) {
  if (!m) return kk_fail("kk_finalize: null model");
  if (m->finalized) return 0;
  const kk_config& c = m->cfg;
  Packer P{m, ""};
  StyleBuilder SP, SD;
  const int H = c.hidden_dim, S = c.style_dim, hs = c.plbert_hidden, E = c.plbert_embedding, DH = c.decoder_hidden;
  // ---- Albert (modules.py:438-649)
  m->emb_word = P.put_named("bert.embeddings.word_embeddings.weight", (size_t)c.n_token * E);
  m->emb_pos = P.put_named("bert.embeddings.position_embeddings.weight", (size_t)c.plbert_max_pos * E);
  m->emb_type = P.put_named("bert.embeddings.token_type_embeddings.weight", (size_t)2 * E);
  m->emb_ln_w = P.put_named("bert.embeddings.LayerNorm.weight", E);
  m->emb_ln_b = P.put_named("bert.embeddings.LayerNorm.bias", E);
  m->map_in = P.linear("bert.encoder.embedding_hidden_mapping_in", hs, E, true);
  const std::string lp = "bert.encoder.albert_layer_groups.0.albert_layers.0.";
  {
    std::vector<float> w((size_t)3 * hs * hs), b((size_t)3 * hs);
    const char* nm[3] = {"query", "key", "value"};
    for (int i = 0; i < 3; ++i) {
      std::vector<float> wi, bi;
      if (P.vec(lp + "attention." + nm[i] + ".weight", (size_t)hs * hs, wi) && P.vec(lp + "attention." + nm[i] + ".bias", hs, bi)) {
        memcpy(&w[(size_t)i * hs * hs], wi.data(), wi.size() * 4);
        memcpy(&b[(size_t)i * hs], bi.data(), bi.size() * 4);
      }
    }
    m->qkv = P.pack_oki(w, 3 * hs, 1, hs, &b);
    P.pack_fp8(m->qkv, w, 3 * hs, hs);
  }
  m->att_dense = P.linear(lp + "attention.dense", hs, hs, true);
  m->att_ln_w = P.put_named(lp + "attention.LayerNorm.weight", hs);
  m->att_ln_b = P.put_named(lp + "attention.LayerNorm.bias", hs);
  m->full_ln_w = P.put_named(lp + "full_layer_layer_norm.weight", hs);
  m->full_ln_b = P.put_named(lp + "full_layer_layer_norm.bias", hs);
  m->ffn = P.linear(lp + "ffn", c.plbert_intermediate, hs, true);
  m->ffn_out = P.linear(lp + "ffn_output", hs, c.plbert_intermediate, true);
  m->bert_encoder = P.linear("bert_encoder", H, hs, true);
  // ---- prosody predictor (modules.py:288-342,380-387)
  for (int i = 0; i < c.n_layer; ++i) {
    m->dur_lstms.push_back(P.lstm("predictor.text_encoder.lstms." + std::to_string(2 * i), H + S, H / 2));
    m->dur_adaln.push_back(SP.add("predictor.text_encoder.lstms." + std::to_string(2 * i + 1), H));
  }
  m->pred_lstm = P.lstm("predictor.lstm", H + S, H / 2);
  m->shared_lstm = P.lstm("predictor.shared", H + S, H / 2);
  {  // stored transposed [H][max_dur]: the duration kernel's lanes (one per output bin) then read consecutive addresses
    std::vector<float> wd, wt((size_t)c.max_dur * H);
    if (P.vec("predictor.duration_proj.linear_layer.weight", (size_t)c.max_dur * H, wd)) {
      for (int o = 0; o < c.max_dur; ++o)
        for (int i = 0; i < H; ++i) wt[(size_t)i * c.max_dur + o] = wd[(size_t)o * H + i];
      m->dur_W = P.put(wt);
    }
  }
  m->dur_b = P.put_named("predictor.duration_proj.linear_layer.bias", c.max_dur);
  for (int which = 0; which < 2; ++which) {
    const std::string nm = which == 0 ? "predictor.F0" : "predictor.N";
    ResBlk1d* blk = which == 0 ? m->f0blk : m->nblk;
    blk[0] = build_resblk1d(P, SP, nm + ".0", H, H, false);
    blk[1] = build_resblk1d(P, SP, nm + ".1", H, H / 2, true);
    blk[2] = build_resblk1d(P, SP, nm + ".2", H / 2, H / 2, false);
    (which == 0 ? m->f0_proj : m->n_proj) = P.conv_plain(nm + "_proj", 1, 1, H / 2);
  }
  // ---- text encoder (modules.py:21-39)
  m->te_emb = P.put_named("text_encoder.embedding.weight", (size_t)c.n_token * H);
  for (int i = 0; i < c.n_layer; ++i) {
    const std::string is = std::to_string(i);
    m->te_cnn.push_back(P.convw("text_encoder.cnn." + is + ".0", H, c.text_encoder_kernel_size, H, true));
    m->te_ln_w.push_back(P.put_named("text_encoder.cnn." + is + ".1.weight", H));
    m->te_ln_b.push_back(P.put_named("text_encoder.cnn." + is + ".1.bias", H));
  }
  m->text_lstm = P.lstm("text_encoder.lstm", H, H / 2);
  // ---- decoder (istftnet.py:902-945)
  m->enc = build_resblk1d(P, SD, "decoder.encode", H + 2, DH, false);
  for (int i = 0; i < 3; ++i) m->dec[i] = build_resblk1d(P, SD, "decoder.decode." + std::to_string(i), DH + 2 + 64, DH, false);
  m->dec[3] = build_resblk1d(P, SD, "decoder.decode.3", DH + 2 + 64, H, true);
  m->f0_conv = P.convw("decoder.F0_conv", 1, 3, 1, true);
  m->n_conv = P.convw("decoder.N_conv", 1, 3, 1, true);
  m->asr_res = P.convw("decoder.asr_res.0", 64, 1, H, true);
  // ---- generator (istftnet.py:696-767)
  const std::string g = "decoder.generator.";
  {
    std::vector<float> lw, lb;
    if (P.vec(g + "m_source.l_linear.weight", 9, lw) && P.vec(g + "m_source.l_linear.bias", 1, lb)) {
      m->lin_w = P.put(lw);
      m->lin_b = lb[0];
    }
  }
  const int C0 = c.upsample_initial_channel, nk = c.n_resblock_kernels;
  const int d135[3] = {1, 3, 5};
  for (int i = 0; i < c.n_upsamples; ++i) {
    const int cin = C0 >> i, cout = C0 >> (i + 1);
    const std::string is = std::to_string(i);
    m->ups[i] = P.convw_t(g + "ups." + is, cin, c.upsample_kernel_sizes[i], cout);
    if (i + 1 < c.n_upsamples) {
      int sf0 = 1;
      for (int j = i + 1; j < c.n_upsamples; ++j) sf0 *= c.upsample_rates[j];
      m->noise_conv[i] = P.conv_plain(g + "noise_convs." + is, cout, sf0 * 2, c.gen_istft_n_fft + 2);
      m->noise_conv_rows[i] = P.strided_rows(g + "noise_convs." + is, cout, sf0 * 2, c.gen_istft_n_fft + 2, sf0, (sf0 + 1) / 2, 64, &m->noise_rows_pad[i]);
      m->noise_res[i] = build_resblock1(P, SD, g + "noise_res." + is, cout, 7, d135);
    } else {
      m->noise_conv[i] = P.conv_plain(g + "noise_convs." + is, cout, 1, c.gen_istft_n_fft + 2);
      m->noise_res[i] = build_resblock1(P, SD, g + "noise_res." + is, cout, 11, d135);
    }
    for (int j = 0; j < nk; ++j)
      m->resblocks.push_back(
          build_resblock1(P, SD, g + "resblocks." + std::to_string(i * nk + j), cout, c.resblock_kernel_sizes[j], c.resblock_dilations[j]));
  }
  m->conv_post = P.convw(g + "conv_post", c.gen_istft_n_fft + 2, 7, C0 >> c.n_upsamples, true);
  if (P.ok() && m->adt == KK_BF16 && m->conv_post.mfma &&
      kk_head_eligible(m->conv_post.Cin, m->conv_post.Cout, m->conv_post.Kw, c.gen_istft_n_fft, c.gen_istft_hop_size)) {
    // the fused head (kk_head.hip) reads conv_post in its own fragment order: the same bf16 values as the MFMA pack
    const ConvW& cp = m->conv_post;
    m->head_wf_off = P.alloc((kk_head_pack_elems() + 1) / 2);  // zero-filled: output columns 22..31 are padding
    const uint16_t* src = (const uint16_t*)&m->pack[cp.wb_off];
    uint16_t* dst = (uint16_t*)&m->pack[m->head_wf_off];
    for (int t = 0; t < cp.Kw; ++t)
      for (int o = 0; o < cp.Cout; ++o)
        for (int i = 0; i < cp.Cin; ++i) dst[kk_head_pack_index(t, o, i)] = src[((size_t)t * cp.CoutP + o) * cp.CinP + i];
  }
  if (P.ok()) {
    SP.build(P, m->sp_wT, m->sp_b, m->Np);
    SD.build(P, m->sd_wT, m->sd_b, m->Nd);
  }
  if (!P.ok()) return failf("kk_finalize: %s", P.err);

  // ---- upload and resolve device pointers
  if (hipMalloc((void**)&m->dev, m->pack.size() * sizeof(float)) != hipSuccess) return kk_fail("kk_finalize: hipMalloc failed");
  if (hipMemcpyAsync(m->dev, m->pack.data(), m->pack.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return kk_fail("kk_finalize: upload failed");
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return kk_fail("kk_finalize: stream sync failed");
  VecW* vecs[] = {&m->emb_word, &m->emb_pos, &m->emb_type, &m->emb_ln_w, &m->emb_ln_b, &m->att_ln_w, &m->att_ln_b, &m->full_ln_w,
                  &m->full_ln_b, &m->dur_W, &m->dur_b, &m->te_emb, &m->lin_w, &m->sp_wT, &m->sp_b, &m->sd_wT, &m->sd_b};
  for (VecW* v : vecs) resolve(m, *v);
  ConvW* convs[] = {&m->map_in, &m->qkv, &m->att_dense, &m->ffn, &m->ffn_out, &m->bert_encoder, &m->f0_proj, &m->n_proj,
                    &m->f0_conv, &m->n_conv, &m->asr_res, &m->conv_post};
  for (ConvW* cw : convs) resolve(m, *cw);
  m->head_wf = m->head_wf_off ? (const bf16_t*)(m->dev + m->head_wf_off) : nullptr;
  for (auto& l : m->dur_lstms) resolve(m, l);
  resolve(m, m->pred_lstm); resolve(m, m->shared_lstm); resolve(m, m->text_lstm);
  for (int i = 0; i < 3; ++i) { resolve(m, m->f0blk[i]); resolve(m, m->nblk[i]); }
  for (auto& cw : m->te_cnn) resolve(m, cw);
  for (auto& v : m->te_ln_w) resolve(m, v);
  for (auto& v : m->te_ln_b) resolve(m, v);
  resolve(m, m->enc);
  for (int i = 0; i < 4; ++i) resolve(m, m->dec[i]);
  for (int i = 0; i < c.n_upsamples; ++i) { resolve(m, m->noise_conv[i]); resolve(m, m->noise_conv_rows[i]); resolve(m, m->ups[i]); resolve(m, m->noise_res[i]); }
  for (auto& r : m->resblocks) resolve(m, r);
  m->host.clear();
  m->pack.clear();
  m->pack.shrink_to_fit();
  m->finalized = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
namespace {

struct Buf {
  void* p = nullptr;
  int ld = 0;
  long long bs = 0;
  int rows = 0;
  int dtype = KK_F32;
  Buf slice(int coff) const {
    Buf r = *this;
    r.p = p ? (char*)p + (size_t)coff * (dtype == KK_F32 ? 4 : 2) : nullptr;
    return r;
  }
  Buf row_offset(int r0) const {
    Buf r = *this;
    r.p = p ? (char*)p + (size_t)r0 * ld * (dtype == KK_F32 ? 4 : 2) : nullptr;
    r.rows = rows - r0;
    return r;
  }
};

struct ConvOpt {
  int mode = KK_CONV, stride = 1, pad = 0, dil = 1, in_shift = 0;
  float in_slope = 1.f;
  int act = KK_ACT_NONE;
  float act_slope = 0.f;
  float scale = 1.f;
  int accumulate = 0;
  const Buf* res = nullptr;
  // MFMA-only fusions (bf16 mode): AdaIN + activation applied while the input is staged; statistics of the output
  const float* nrm_a = nullptr;
  const float* nrm_b = nullptr;
  int nrm_act = KK_ACT_NONE;
  float nrm_slope = 0.f;
  const float* nrm_alpha = nullptr;
  int nrm_C = 0;
  bool want_stats = false;
  bool pad_out_ok = false;  // the destination may receive up to 7 extra zero channels (Cout rounded up to 8)
  double alg_taps_cin = 0;  // algorithmic taps*Cin of the op this launch implements, when the packed form carries structural zeros
  float post_slope = 1.f;   // MFMA variants 4 / 5: LeakyReLU(post_slope) of the FINAL stored value (after residual, scale, accumulate); 1 = none
};

struct Ctx {
  kk_model* m;
  kk_context* cx = nullptr;  // the switches it carries shape the allocation plan too, so even a dry run has one
  hipStream_t st;
  hipStream_t main_st = nullptr;  // the caller's (or the capture) stream while a branch runs on the side stream
  bool dry;
  // side branch k: fork_point(k) marks where its inputs are ready on the main stream; begin_side(k) ... end_side(k) brackets its launches
  // (they go to the model's side stream); join_side(k) makes the main stream wait for it.  All no-ops in a dry run / with the debug switch.
  // Not on the legacy NULL stream: an event recorded on / waited for by stream 0 did not order it against a non-blocking stream here (measured:
  // the branch read its inputs early), and a blocking side stream would serialise with stream 0 anyway.  A capture runs on the model's own stream,
  // so a replayed graph has the branches whatever stream it is launched on.
  bool side_on() const { return !dry && !cx->no_side && cx->side_stream && st != nullptr; }
  static int side_mask() {  // (debugging: KK_SIDE_MASK bit k enables branch k; default both)
    static int v = -1;
    if (v < 0) {
      const char* e = getenv("KK_SIDE_MASK");
      v = e ? atoi(e) : 3;
    }
    return v;
  }
  int fork_point(int k) {
    if (!side_on() || !(side_mask() & (1 << k))) return 0;
    return hipEventRecord(cx->side_fork[k], st) == hipSuccess ? 0 : kk_fail("kk_forward: hipEventRecord failed");
  }
  int begin_side(int k) {
    if (!side_on() || !(side_mask() & (1 << k))) return 0;
    if (hipStreamWaitEvent(cx->side_stream, cx->side_fork[k], 0) != hipSuccess) return kk_fail("kk_forward: hipStreamWaitEvent failed");
    main_st = st;
    st = cx->side_stream;
    return 0;
  }
  int end_side(int k) {
    if (!side_on() || !main_st) return 0;
    const hipError_t e = hipEventRecord(cx->side_join[k], st);
    st = main_st;
    main_st = nullptr;
    return e == hipSuccess ? 0 : kk_fail("kk_forward: hipEventRecord failed");
  }
  int join_side(int k) {
    if (!side_on() || !(side_mask() & (1 << k))) return 0;
    return hipStreamWaitEvent(st, cx->side_join[k], 0) == hipSuccess ? 0 : kk_fail("kk_forward: hipStreamWaitEvent failed");
  }
  char* base;
  size_t cap, used = 0;
  int B;
  int adt;

  void* raw(size_t bytes) {
    const size_t off = (used + 255) & ~(size_t)255;
    used = off + bytes;
    return base ? base + off : nullptr;
  }
  Buf act(int rows, int ld, int dtype = -1) {
    Buf b;
    b.dtype = dtype < 0 ? adt : dtype;
    b.ld = ld;
    b.rows = rows;
    b.bs = (long long)rows * ld;
    b.p = raw((size_t)B * rows * ld * (b.dtype == KK_F32 ? 4 : 2));
    return b;
  }
  float* f32(size_t n) { return (float*)raw(n * 4); }
  int* i32(size_t n) { return (int*)raw(n * 4); }

  // ---- profiling: bracket a launch with events (only when kk_profile_begin was called)
  void prof_start() {
    if (dry || !cx->prof_on) return;
    const size_t i = cx->prof_rec.size();
    if (2 * i + 1 >= cx->prof_ev.size()) return;
    (void)hipEventRecord(cx->prof_ev[2 * i], st);
  }
  void prof_stop(int cls, double flops, double bytes) {
    if (dry || !cx->prof_on) return;
    const size_t i = cx->prof_rec.size();
    if (2 * i + 1 >= cx->prof_ev.size()) return;
    (void)hipEventRecord(cx->prof_ev[2 * i + 1], st);
    cx->prof_rec.push_back({cls, flops, bytes});
  }
  static double esz(int dt) { return dt == KK_F32 ? 4.0 : 2.0; }

  int dbg(const char* name, const Buf& b, int C) {
    if (dry) return 0;
    auto it = cx->dbg_over.find(name);
    if (it != cx->dbg_over.end())
      KK_TRY(kk_launch_convert(it->second, KK_F32, (long long)b.rows * C, C, b.p, b.dtype, b.bs, b.ld, C, b.rows, B, st));
    cx->dbg[name] = DebugEntry{b.p, b.ld, b.bs, b.rows, C, b.dtype, B};
    return 0;
  }

  int conv(const ConvW& w, const Buf& x, KKLen lin, const Buf& out, KKLen lout, int Q, const ConvOpt& o) {
    if (dry) return 0;
    KKConvArgs a;
    memset(&a, 0, sizeof a);
    a.x = x.p; a.xbs = x.bs; a.ldx = x.ld;
    a.w = w.w; a.ldw = w.ldw; a.bias = w.b;
    a.out = out.p; a.obs = out.bs; a.ldo = out.ld;
    if (o.res) { a.res = o.res->p; a.rbs = o.res->bs; a.ldr = o.res->ld; }
    a.Cin = w.Cin; a.Cout = w.Cout; a.Kw = w.Kw;
    a.mode = o.mode; a.stride = o.stride; a.pad = o.pad; a.dil = o.dil; a.in_shift = o.in_shift;
    a.Q = Q; a.Lo_rows = out.rows;
    a.lin = lin; a.lout = lout;
    a.in_slope = o.in_slope; a.scale = o.scale; a.accumulate = o.accumulate; a.act = o.act; a.act_slope = o.act_slope;
    // algorithmic work of this launch at full length: every output row sums Kw*Cin (conv) or Kw/stride*Cin (convT) products
    const double rows_out = o.mode == KK_CONVT ? (double)Q * o.stride : (double)Q;
    const double taps = o.mode == KK_CONVT ? (double)w.Kw / o.stride : (double)w.Kw;
    const double flops = 2.0 * B * rows_out * w.Cout * (o.alg_taps_cin > 0 ? o.alg_taps_cin : w.Cin * taps);
    const double bytes = B * (rows_out * w.Cout * esz(out.dtype) * (o.res ? 2.0 : 1.0) + (double)Q * (o.mode == KK_CONVT ? 1 : o.stride) * w.Cin * esz(x.dtype)) +
                         (double)w.Kw * w.Cin * w.Cout * 4.0;
    if (can_fp8(w, x, lin, out, lout, Q, o)) {
      // the reference's quantised layer set (tts/utils.py:241-260) on the block-scaled fp8 matrix instruction: activation pre-pass, then the product
      KKFp8Args f;
      memset(&f, 0, sizeof f);
      f.aq = (const uint4*)q8_aq; f.as = (const unsigned char*)q8_as; f.wq = w.q8; f.ws = w.s8;
      f.M = B * x.rows; f.N = w.Cout; f.K = w.Cin; f.bias = w.b; f.out = (bf16_t*)out.p; f.ldo = out.ld; f.rows_per_item = x.rows;
      f.lout = lout; f.act = o.act;
      prof_start();
      int rc = kk_launch_mxfp8_quant_rows(x.p, x.ld, f.M, f.K, q8_aq, q8_as, st);
      if (rc == 0) rc = kk_launch_linear_mxfp8(f, st);
      prof_stop(10, flops, B * (double)Q * (w.Cin * 2.0 + w.Cin * 1.03 * 2.0 + w.Cout * 2.0) + (double)w.Cin * w.Cout * 1.03);
      return rc;
    }
    // Linear layers (k = 1, no fused transform, bias / GELU only): the streaming matrix-core kernel (kk_linear_rows.hip) while few rows are in flight -- the
    // latency-bound case (B = 1: 37 -> ~10 us per launch); from ~1000 rows on the tiled kernel's operand reuse wins (B = 32: the streaming form costs 4 % of the
    // step).  Both kernels feed the SAME matrix instruction the same operands in the same K order and share the epilogue arithmetic, so their results are
    // bit-identical (tests/test_gpu_forward.py::test_streaming_linear_equals_tiled_bitexact) and the choice by size does not touch batch invariance.
    static int linrows_max = -1;
    if (linrows_max < 0) { const char* e = getenv("KK_LINROWS_MAX"); linrows_max = e ? atoi(e) : 1024; }
    const bool lr_size_ok = cx->linrows_mode == 1 || (cx->linrows_mode == 0 && (long long)B * Q <= linrows_max);
    if (lr_size_ok && w.wl && w.Kw == 1 && can_mfma(w, x, out, o) && out.dtype == KK_BF16 && o.mode == KK_CONV && o.stride == 1 && o.pad == 0 && o.dil == 1 &&
        o.in_shift == 0 && !o.nrm_a && !o.want_stats && !o.res && !o.accumulate && o.in_slope == 1.f && o.scale == 1.f && o.post_slope == 1.f &&
        (o.act == KK_ACT_NONE || o.act == KK_ACT_GELU) && Q == x.rows && Q == out.rows && lin.len == lout.len && lin.mul == lout.mul && lin.add == lout.add &&
        !cx->no_v4) {
      KKLinMfmaArgs f;
      memset(&f, 0, sizeof f);
      f.x = (const bf16_t*)x.p; f.xbs = x.bs; f.ldx = x.ld; f.wl = w.wl; f.bias = w.b; f.Nb = w.CoutP;
      f.out = (bf16_t*)out.p; f.obs = out.bs; f.ldo = out.ld; f.K = w.Cin; f.N = w.Cout8; f.rows = Q; f.items = B;
      f.flat = (x.bs == (long long)x.rows * x.ld && out.bs == (long long)out.rows * out.ld) ? 1 : 0;
      f.len = lout; f.act = o.act;
      prof_start();
      const int rc = kk_launch_linear_rows_mfma(f, st);
      prof_stop(1, flops, bytes);
      return rc;
    }
    if (can_mfma(w, x, out, o)) {
      KKMfmaArgs g;
      memset(&g, 0, sizeof g);
      g.nrm_a = o.nrm_a; g.nrm_b = o.nrm_b; g.nrm_stride = fz_stride; g.nrm_act = o.nrm_act; g.nrm_slope = o.nrm_slope;
      g.nrm_alpha = o.nrm_alpha; g.nrm_C = o.nrm_C;
      g.x = (const bf16_t*)x.p; g.xbs = x.bs; g.ldx = x.ld; g.w = w.wb; g.CinP = w.CinP; g.Cin = w.Cin; g.CoutP = w.CoutP; g.bias = w.b;
      g.out = out.p; g.obs = out.bs; g.ldo = out.ld;
      if (o.res) { g.res = o.res->p; g.rbs = o.res->bs; g.ldr = o.res->ld; }
      g.Cout = w.Cout8; g.Kw = w.Kw; g.mode = o.mode; g.stride = o.stride; g.pad = o.pad; g.dil = o.dil; g.in_shift = o.in_shift;
      g.Q = Q; g.Lo_rows = out.rows; g.lin = lin; g.lout = lout; g.in_slope = o.in_slope; g.scale = o.scale; g.accumulate = o.accumulate;
      g.act = o.act; g.act_slope = o.act_slope; g.post_slope = o.post_slope;
      if (o.want_stats) {
        last_ntiles = kk_cdiv(Q, kk_mfma_stat_tile_rows(g, out.dtype)) * (o.mode == KK_CONVT ? o.stride : 1);
        if ((size_t)B * last_ntiles * 2 * w.Cout > fz_part_floats) return kk_fail("internal: statistics scratch too small");
        g.stat_part = fz_part;
        g.stat_ntiles = last_ntiles;
      }
      // variant 4 (W fragments straight into registers): bf16 outputs at the default 192-row tile
      const bool v4 = w.wf && out.dtype == KK_BF16 && !cx->no_v4;
      g.wf = v4 ? w.wf : nullptr;
      // variant 5 (wave-specialised, persistent) takes the stride-1 convolutions; the polyphase transposed ones stay on variant 4
      // measured per shape: 3-9 % faster than variant 4 on the 11-tap layers, level on 7 taps, 10-25 % slower on 3 taps; round 3 (16x16x32
      // MFMA in both): also 2 % faster on the 7-tap layers of stage 0 (256 channels = 4 slabs per tile: more periods to spread the service work over)
      static int v5_min_taps = -1;
      if (v5_min_taps < 0) {
        const char* e = getenv("KK_V5_MIN_TAPS");  // (experiments)
        v5_min_taps = e ? atoi(e) : 9;
      }
      const bool v5 = v4 && cx->v5_mode != 2 && (cx->v5_mode == 1 || g.Kw >= v5_min_taps || (g.Kw >= v5_min_taps - 2 && g.CinP >= 256)) && B <= 256 && kk_mfma_tile_rows(Q) == 192 &&
                      kk_mfma5_eligible(g, out.dtype);
      // Linear layers over short utterances (Albert, T = 130 rows per item): the rows of a dense [B][T][C] tensor as ONE flat item, so that the
      // 192-row tiles run across utterance boundaries (32 x 130 rows = 22 tiles instead of 32).  k = 1, so rows do not interact; input rows past
      // an utterance's length are zeros already and the epilogue stores zeros there (KKMfmaArgs::flat_T).
      int Bl = B;
      static int no_flat = -1;
      if (no_flat < 0) no_flat = getenv("KK_NO_FLAT") ? 1 : 0;  // (A/B timing)
      if (!no_flat && !v5 && B > 1 && w.Kw == 1 && o.mode == KK_CONV && o.stride == 1 && o.pad == 0 && o.dil == 1 && o.in_shift == 0 && !o.nrm_a && !o.want_stats &&
          Q == x.rows && Q == out.rows && Q % 192 != 0 && x.bs == (long long)x.rows * x.ld && out.bs == (long long)out.rows * out.ld &&
          (!o.res || (o.res->rows == out.rows && o.res->bs == (long long)o.res->rows * o.res->ld)) && (long long)B * Q < (1ll << 30)) {
        g.flat_T = Q;
        g.flat_len = lout;
        g.Q = B * Q;
        g.Lo_rows = B * out.rows;
        g.lin = KKLen{nullptr, 0, B * Q};
        g.lout = KKLen{nullptr, 0, B * Q};
        Bl = 1;
      }
      prof_start();
      const int rc = v5 ? kk_launch_conv_mfma5(g, Bl, out.dtype, st) : v4 ? kk_launch_conv_mfma4(g, Bl, out.dtype, st) : kk_launch_conv_mfma(g, Bl, out.dtype, st);
      prof_stop(1, flops, bytes);
      return rc;
    }
    if (o.nrm_a || o.want_stats) return kk_fail("internal: fused norm requested on a conv that is not MFMA eligible");
    prof_start();
    const int rc = kk_launch_conv_generic(a, B, x.dtype, out.dtype, st);
    prof_stop(0, flops, bytes);
    return rc;
  }

  // MX-fp8 linear: a plain Linear (k = 1, bias, optional exact GELU) between two whole bf16 buffers of the same row count
  void* q8_aq = nullptr;  // activation fragments / scale bytes of the launch in flight (run_text allocates them)
  void* q8_as = nullptr;
  size_t q8_rows = 0, q8_K = 0;
  bool can_fp8(const ConvW& w, const Buf& x, KKLen lin, const Buf& out, KKLen lout, int Q, const ConvOpt& o) const {
    return w.fp8 && !cx->no_fp8 && !cx->force_generic && q8_aq && x.dtype == KK_BF16 && out.dtype == KK_BF16 && w.Kw == 1 && o.mode == KK_CONV &&
           o.stride == 1 && o.pad == 0 && o.dil == 1 && o.in_shift == 0 && o.in_slope == 1.f && !o.res && o.scale == 1.f && !o.accumulate &&
           (o.act == KK_ACT_NONE || o.act == KK_ACT_GELU) && !o.nrm_a && !o.want_stats && x.rows == out.rows && Q == x.rows &&
           x.bs == (long long)x.rows * x.ld && out.bs == (long long)out.rows * out.ld && x.ld % 8 == 0 && !((uintptr_t)x.p & 15) &&
           lin.len == lout.len && lin.mul == lout.mul && lin.add == lout.add && (size_t)B * x.rows <= q8_rows && (size_t)w.Cin <= q8_K;
  }

  bool can_mfma(const ConvW& w, const Buf& x, const Buf& out, const ConvOpt& o) const {
    const bool al16 = !(((uintptr_t)x.p | (uintptr_t)out.p | (uintptr_t)(o.res ? o.res->p : nullptr)) & 15);
    // a Cout that is not a multiple of 8 is rounded up: legal only for a whole-buffer destination with spare pitch, no residual
    const bool cout_ok = w.Cout == w.Cout8 || (out.ld >= w.Cout8 && !o.res && !o.accumulate && o.pad_out_ok);
    return w.mfma && cout_ok && x.dtype == KK_BF16 && (out.dtype == KK_BF16 || out.dtype == KK_F32) && (!o.res || o.res->dtype == out.dtype) &&
           kk_mfma_eligible(w.Cin, w.Cout8, w.Kw, o.mode, o.stride, o.dil) && x.ld >= w.CinP && x.ld % 8 == 0 && out.ld % 8 == 0 &&
           (!o.res || o.res->ld % 8 == 0) && al16 && !cx->force_generic;
  }

  // ---- fused-norm plumbing (bf16 MFMA path) ------------------------------------------------------------------
  float* fz_part = nullptr;      // per-tile column sums written by conv epilogues
  size_t fz_part_floats = 0;
  float* fz_pa[2] = {nullptr, nullptr};  // ping-pong folded AdaIN parameters [B][fz_stride]
  float* fz_pb[2] = {nullptr, nullptr};
  int fz_stride = 0;
  int fz_flip = 0;
  int last_ntiles = 0;
  // partial sums of the last want_stats conv -> mean/rstd (optional) and folded parameters for the AdaIN `gb`
  int finalize(int C, KKLen len, const float* gb, int gbs, float* mean, float* rstd, float** pa, float** pb) {
    if (dry) return 0;
    KKStatsArgs a;
    memset(&a, 0, sizeof a);
    a.C = C; a.len = len; a.partial = fz_part; a.nchunk = last_ntiles; a.mean = mean; a.rstd = rstd; a.eps = 1e-5f; a.fused = 1;
    if (gb) {
      fz_flip ^= 1;
      a.gb = gb; a.gbs = gbs; a.pa = fz_pa[fz_flip]; a.pb = fz_pb[fz_flip]; a.pstride = fz_stride; a.Cp = rup64(C);
      *pa = a.pa; *pb = a.pb;
    }
    return kk_launch_norm_finalize(a, B, st);
  }
  // known mean/rstd (standalone statistics kernel or a kept copy) -> folded parameters
  int fold(int C, KKLen len, const float* mean, const float* rstd, const float* gb, int gbs, float** pa, float** pb) {
    if (dry) return 0;
    KKStatsArgs a;
    memset(&a, 0, sizeof a);
    fz_flip ^= 1;
    a.C = C; a.len = len; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.fused = 2; a.gb = gb; a.gbs = gbs;
    a.pa = fz_pa[fz_flip]; a.pb = fz_pb[fz_flip]; a.pstride = fz_stride; a.Cp = rup64(C);
    *pa = a.pa; *pb = a.pb;
    return kk_launch_norm_finalize(a, B, st);
  }
  static int rup64(int v) { return (v + 63) / 64 * 64; }

  // scratch for instance-norm statistics, sized for the largest request seen in the dry run
  float* st_partial = nullptr;
  float* st_mean = nullptr;
  float* st_rstd = nullptr;
  // rows one workgroup of the stand-alone statistics pass walks: short tensors (decoder, F rows) need more workgroups
  // a CONSTANT: the grouping of the partial sums must not depend on the longest utterance of the batch, or an utterance's statistics (and
  // every sample after them) change in the last bit with its neighbours (found by test_full_config_batch_invariance_bitexact)
  // -- so it is chosen by the tensor's length DOMAIN (rows per predicted frame: 1 / 2 for the decoder and F0 / N stacks, 20 / 120 for the
  // generator stages), a static property of the layer: short tensors get 64-row chunks (enough workgroups), long ones 512
  static int rows_per_chunk(int rows_per_frame) { return rows_per_frame >= 20 ? 512 : 64; }
  int Fcap = 0;  // frame capacity of this call (run_audio): Lmax / Fcap = the tensor's rows per frame

  int stats(const Buf& x, int C, int Lmax, KKLen len) {
    if (dry) return 0;
    KKStatsArgs a;
    memset(&a, 0, sizeof a);
    a.x = x.p; a.xbs = x.bs; a.ldx = x.ld; a.C = C; a.Lmax = Lmax; a.len = len;
    a.partial = st_partial; a.rows_per_chunk = rows_per_chunk(Fcap > 0 ? Lmax / Fcap : 1); a.mean = st_mean; a.rstd = st_rstd; a.eps = 1e-5f;
    prof_start();
    const int rc = kk_launch_instnorm_stats(a, B, x.dtype, st);
    prof_stop(2, 3.0 * B * Lmax * C, (double)B * Lmax * C * esz(x.dtype));
    return rc;
  }
  int adain(const Buf& x, int C, KKLen len_in, const Buf& out, int Cpad, int Lmax_out, const float* gb, int gbs, int act, float slope,
            const float* alpha, int pool, const float* pool_w, const float* pool_b) {
    if (dry) return 0;
    KKAdainArgs a;
    memset(&a, 0, sizeof a);
    a.x = x.p; a.xbs = x.bs; a.ldx = x.ld; a.out = out.p; a.obs = out.bs; a.ldo = out.ld;
    a.C = C; a.Cpad = Cpad; a.Lmax_out = Lmax_out; a.len_in = len_in;
    a.mean = st_mean; a.rstd = st_rstd; a.gb = gb; a.gbs = gbs; a.act = act; a.slope = slope; a.alpha = alpha;
    a.pool = pool; a.pool_w = pool_w; a.pool_b = pool_b; a.fast = adt == KK_BF16;
    prof_start();
    const int rc = kk_launch_adain_act(a, B, x.dtype, st);
    prof_stop(3, 8.0 * B * Lmax_out * C, (double)B * Lmax_out * C * esz(x.dtype) * (pool ? 1.5 : 2.0));
    return rc;
  }
  int layernorm(const Buf& x, const Buf* res, const Buf& out, int C, int Lmax, KKLen len, const float* w, const float* b, const float* gb,
                int gbs, float eps, int act, float slope) {
    if (dry) return 0;
    KKLnArgs a;
    memset(&a, 0, sizeof a);
    a.x = x.p; a.xbs = x.bs; a.ldx = x.ld;
    if (res) { a.res = res->p; a.rbs = res->bs; a.ldr = res->ld; }
    a.out = out.p; a.obs = out.bs; a.ldo = out.ld; a.C = C; a.Lmax = Lmax; a.len = len; a.w = w; a.bias = b; a.gb = gb; a.gbs = gbs;
    a.eps = eps; a.act = act; a.slope = slope;
    prof_start();
    const int rc = kk_launch_layernorm(a, B, x.dtype, st);
    prof_stop(6, 8.0 * B * Lmax * C, (double)B * Lmax * C * esz(x.dtype) * (res ? 3.0 : 2.0));
    return rc;
  }
  int lstm(const LstmW& l, const Buf& x, int Cin_ld_unused, float* xproj, const Buf& out, int Lmax, KKLen len) {
    (void)Cin_ld_unused;
    if (dry) return 0;
    Buf xp;
    xp.p = xproj; xp.ld = 8 * l.H; xp.bs = (long long)Lmax * 8 * l.H; xp.rows = Lmax; xp.dtype = KK_F32;
    ConvOpt o;
    KK_TRY(conv(l.in, x, len, xp, len, Lmax, o));
    KKLstmArgs a;
    memset(&a, 0, sizeof a);
    a.xproj = xproj; a.whT = l.whT.p; a.out = out.p; a.obs = out.bs; a.ldo = out.ld; a.H = l.H; a.Lmax = Lmax; a.len = len;
    prof_start();
    const int rc = (l.whb && adt == KK_BF16 && !cx->force_generic) ? kk_launch_lstm_h256_bf16(a, l.whb, B, out.dtype, st)
                                                                    : kk_launch_lstm(a, B, out.dtype, st);
    prof_stop(4, 2.0 * B * Lmax * 2 * 4 * l.H * l.H, 2.0 * Lmax * 4 * l.H * l.H * 4.0 * B);
    return rc;
  }
};

static int rup(int v, int m) { return (v + m - 1) / m * m; }

// AdainResBlk1d (istftnet.py:825-899).  x: [B][Lmax_in][>=Cin]; out: channel slice receiving Cout channels at Lout rows.
int run_resblk1d(Ctx& c, const ResBlk1d& r, const Buf& x, KKLen lin, int Lmax_in, const Buf& out, const float* style, int gbs,
                 Buf& bufA, Buf& bufB, Buf& bufC) {
  const int Lmax_out = r.up ? 2 * Lmax_in : Lmax_in;
  KKLen lout = lin;
  if (r.up) { lout.mul = lin.mul * 2; lout.add = lin.add * 2; }
  {
    // bf16 MFMA path: AdaIN + LeakyReLU ride in the convs' input staging, statistics in their epilogues
    ConvOpt p1, p2;
    p1.pad = p2.pad = 1;
    p2.res = r.learned ? &out : &x;
    const Buf& c1in = r.up ? bufA : x;
    if (c.adt == KK_BF16 && c.fz_part && !c.cx->no_fusion && c.can_mfma(r.conv1, c1in, bufB, p1) && c.can_mfma(r.conv2, bufB, out, p2) &&
        (!r.learned || c.can_mfma(r.sc, x, out, ConvOpt()))) {
      float *pa = nullptr, *pb = nullptr;
      KK_TRY(c.stats(x, r.Cin, Lmax_in, lin));
      ConvOpt o1;
      o1.pad = 1;
      o1.want_stats = true;
      if (r.up) {
        KK_TRY(c.adain(x, r.Cin, lin, bufA, rup(r.Cin, 64) <= bufA.ld ? rup(r.Cin, 64) : r.Cin, Lmax_out, style + r.n1.off, gbs,
                       KK_ACT_LRELU, 0.2f, nullptr, 1, r.pool_w.p, r.pool_b.p));
      } else {
        KK_TRY(c.fold(r.Cin, lin, c.st_mean, c.st_rstd, style + r.n1.off, gbs, &pa, &pb));
        o1.nrm_a = pa; o1.nrm_b = pb; o1.nrm_act = KK_ACT_LRELU; o1.nrm_slope = 0.2f; o1.nrm_C = r.Cin;
      }
      KK_TRY(c.conv(r.conv1, c1in, lout, bufB, lout, Lmax_out, o1));
      KK_TRY(c.finalize(r.Cout, lout, style + r.n2.off, gbs, nullptr, nullptr, &pa, &pb));
      ConvOpt o2;
      o2.pad = 1;
      o2.scale = 0.70710678118654752440f;
      o2.nrm_a = pa; o2.nrm_b = pb; o2.nrm_act = KK_ACT_LRELU; o2.nrm_slope = 0.2f; o2.nrm_C = r.Cout;
      if (r.learned) {
        ConvOpt os;
        os.in_shift = r.up ? 1 : 0;
        KK_TRY(c.conv(r.sc, x, lin, out, lout, Lmax_out, os));
        o2.res = &out;
      } else {
        o2.res = &x;
      }
      return c.conv(r.conv2, bufB, lout, out, lout, Lmax_out, o2);
    }
  }
  KK_TRY(c.stats(x, r.Cin, Lmax_in, lin));
  KK_TRY(c.adain(x, r.Cin, lin, bufA, rup(r.Cin, 64) <= bufA.ld ? rup(r.Cin, 64) : r.Cin, Lmax_out, style + r.n1.off, gbs, KK_ACT_LRELU,
                 0.2f, nullptr, r.up ? 1 : 0, r.pool_w.p, r.pool_b.p));
  ConvOpt o1;
  o1.pad = 1;
  KK_TRY(c.conv(r.conv1, bufA, lout, bufB, lout, Lmax_out, o1));
  KK_TRY(c.stats(bufB, r.Cout, Lmax_out, lout));
  KK_TRY(c.adain(bufB, r.Cout, lout, bufC, r.Cout, Lmax_out, style + r.n2.off, gbs, KK_ACT_LRELU, 0.2f, nullptr, 0, nullptr, nullptr));
  ConvOpt o2;
  o2.pad = 1;
  o2.scale = 0.70710678118654752440f;  // / sqrt(2), istftnet.py:898
  if (r.learned) {
    ConvOpt os;
    os.in_shift = r.up ? 1 : 0;  // nearest x2 up-sampling of the shortcut input (istftnet.py:863-866)
    KK_TRY(c.conv(r.sc, x, lin, out, lout, Lmax_out, os));
    o2.res = &out;
  } else {
    o2.res = &x;
  }
  KK_TRY(c.conv(r.conv2, bufC, lout, out, lout, Lmax_out, o2));
  return 0;
}

// AdaINResBlock1 (istftnet.py:377-396).  Iteration 0 reads x_in; iterations keep their running value in y.
// If acc != null the last iteration writes acc (+)= (conv + y) * acc_scale instead of y.
int run_resblock1(Ctx& c, const ResBlock1& r, const Buf& x_in, const Buf& y, Buf& t1, Buf& t2, int Lmax, KKLen len, const float* style,
                  int gbs, const Buf* acc, float acc_scale, int acc_accumulate, bool x_stats_ready = false, float post_slope = 1.f) {
  {
    ConvOpt p1, p2;
    p1.dil = r.dil[2]; p1.pad = (r.k * r.dil[2] - r.dil[2]) / 2;
    p2.pad = (r.k - 1) / 2; p2.res = &x_in;
    if (c.adt == KK_BF16 && c.fz_part && !c.cx->no_fusion && c.can_mfma(r.c1[0], x_in, t2, p1) && c.can_mfma(r.c2[0], t2, y, p2) &&
        (!acc || c.can_mfma(r.c2[2], t2, *acc, p2))) {
      // bf16 MFMA path: 2 convs + 2 tiny parameter folds per iteration, 5 tensor passes instead of 12
      float *pa = nullptr, *pb = nullptr;
      if (!x_stats_ready) KK_TRY(c.stats(x_in, r.C, Lmax, len));
      KK_TRY(c.fold(r.C, len, c.st_mean, c.st_rstd, style + r.a1[0].off, gbs, &pa, &pb));
      for (int j = 0; j < 3; ++j) {
        const Buf& src = j == 0 ? x_in : y;
        ConvOpt o1;
        o1.dil = r.dil[j];
        o1.pad = (r.k * r.dil[j] - r.dil[j]) / 2;
        o1.nrm_a = pa; o1.nrm_b = pb; o1.nrm_act = KK_ACT_SNAKE; o1.nrm_alpha = r.al1[j].p; o1.nrm_C = r.C;
        o1.want_stats = true;
        KK_TRY(c.conv(r.c1[j], src, len, t2, len, Lmax, o1));
        KK_TRY(c.finalize(r.C, len, style + r.a2[j].off, gbs, nullptr, nullptr, &pa, &pb));
        ConvOpt o2;
        o2.pad = (r.k - 1) / 2;
        o2.res = &src;
        o2.nrm_a = pa; o2.nrm_b = pb; o2.nrm_act = KK_ACT_SNAKE; o2.nrm_alpha = r.al2[j].p; o2.nrm_C = r.C;
        if (j == 2 && acc) {
          o2.scale = acc_scale;
          o2.accumulate = acc_accumulate;
          o2.post_slope = post_slope;
          KK_TRY(c.conv(r.c2[j], t2, len, *acc, len, Lmax, o2));
        } else {
          o2.want_stats = j < 2;
          KK_TRY(c.conv(r.c2[j], t2, len, y, len, Lmax, o2));
          if (j < 2) KK_TRY(c.finalize(r.C, len, style + r.a1[j + 1].off, gbs, nullptr, nullptr, &pa, &pb));
        }
      }
      return 0;
    }
  }
  for (int j = 0; j < 3; ++j) {
    const Buf& src = j == 0 ? x_in : y;
    KK_TRY(c.stats(src, r.C, Lmax, len));
    KK_TRY(c.adain(src, r.C, len, t1, r.C, Lmax, style + r.a1[j].off, gbs, KK_ACT_SNAKE, 0.f, r.al1[j].p, 0, nullptr, nullptr));
    ConvOpt o1;
    o1.dil = r.dil[j];
    o1.pad = (r.k * r.dil[j] - r.dil[j]) / 2;
    KK_TRY(c.conv(r.c1[j], t1, len, t2, len, Lmax, o1));
    KK_TRY(c.stats(t2, r.C, Lmax, len));
    KK_TRY(c.adain(t2, r.C, len, t1, r.C, Lmax, style + r.a2[j].off, gbs, KK_ACT_SNAKE, 0.f, r.al2[j].p, 0, nullptr, nullptr));
    ConvOpt o2;
    o2.pad = (r.k - 1) / 2;
    o2.res = &src;
    if (j == 2 && acc) {
      o2.scale = acc_scale;
      o2.accumulate = acc_accumulate;
      KK_TRY(c.conv(r.c2[j], t1, len, *acc, len, Lmax, o2));
    } else {
      KK_TRY(c.conv(r.c2[j], t1, len, y, len, Lmax, o2));
    }
  }
  return 0;
}

struct TextState {  // results of the text stage that the audio stage consumes
  Buf d, t_en;
  int* pred_dur = nullptr;
};

int run_text(Ctx& c, int Tmax, const int* ids, const int* lens, const float* ref_s, const float* speed, TextState& ts, int* pred_dur_out) {
  kk_model* m = c.m;
  kk_context* cx = c.cx;
  (void)cx;
  const kk_config& cf = m->cfg;
  const int H = cf.hidden_dim, hs = cf.plbert_hidden, E = cf.plbert_embedding, B = c.B;
  const KKLen lT{lens, 1, 0};
  if (m->q_group && m->adt == KK_BF16) {  // MX-fp8 activation scratch of the quantised linears (largest K of the layer set)
    c.q8_rows = (size_t)B * Tmax;
    c.q8_K = (size_t)std::max(std::max(hs, E), (int)cf.plbert_intermediate);
    c.q8_aq = c.raw(kk_mxfp8_q_bytes((int)c.q8_rows, (int)c.q8_K));
    c.q8_as = c.raw(kk_mxfp8_s_bytes((int)c.q8_rows, (int)c.q8_K));
    if (c.dry) c.q8_aq = nullptr;
  }
  KK_TRY(c.fork_point(0));  // TextEncoder (below) needs the ids only
  // ---- Albert
  Buf e = c.act(Tmax, E), x = c.act(Tmax, hs), qkv = c.act(Tmax, 3 * hs), ctxb = c.act(Tmax, hs), att = c.act(Tmax, hs),
      ff = c.act(Tmax, cf.plbert_intermediate), tmp = c.act(Tmax, hs);
  if (!c.dry) {
    KKEmbedArgs ea;
    memset(&ea, 0, sizeof ea);
    ea.ids = ids; ea.word = m->emb_word.p; ea.pos = m->emb_pos.p; ea.type = m->emb_type.p; ea.ln_w = m->emb_ln_w.p; ea.ln_b = m->emb_ln_b.p;
    ea.out = e.p; ea.obs = e.bs; ea.ldo = e.ld; ea.E = E; ea.Tmax = Tmax; ea.len = lT; ea.eps = 1e-12f;
    KK_TRY(kk_launch_albert_embed(ea, B, e.dtype, c.st));
  }
  ConvOpt plain;
  KK_TRY(c.conv(m->map_in, e, lT, x, lT, Tmax, plain));
  for (int layer = 0; layer < cf.plbert_layers; ++layer) {
    KK_TRY(c.conv(m->qkv, x, lT, qkv, lT, Tmax, plain));
    if (!c.dry) {
      KKAttnArgs aa;
      memset(&aa, 0, sizeof aa);
      aa.qkv = qkv.p; aa.bs = qkv.bs; aa.ld = qkv.ld; aa.out = ctxb.p; aa.obs = ctxb.bs; aa.ldo = ctxb.ld;
      aa.heads = cf.plbert_heads; aa.hs = hs; aa.Tmax = Tmax; aa.len = lT; aa.scale = 0.125f;
      c.prof_start();
      KK_TRY(kk_launch_attention(aa, B, qkv.dtype, c.st));
      c.prof_stop(7, 4.0 * B * cf.plbert_heads * (double)Tmax * Tmax * 64, (double)B * Tmax * 4 * hs * Ctx::esz(qkv.dtype));
    }
    KK_TRY(c.conv(m->att_dense, ctxb, lT, tmp, lT, Tmax, plain));
    KK_TRY(c.layernorm(tmp, &x, att, hs, Tmax, lT, m->att_ln_w.p, m->att_ln_b.p, nullptr, 0, 1e-12f, KK_ACT_NONE, 0.f));
    ConvOpt gelu;
    gelu.act = KK_ACT_GELU;
    KK_TRY(c.conv(m->ffn, att, lT, ff, lT, Tmax, gelu));
    KK_TRY(c.conv(m->ffn_out, ff, lT, tmp, lT, Tmax, plain));
    KK_TRY(c.layernorm(tmp, &att, x, hs, Tmax, lT, m->full_ln_w.p, m->full_ln_b.p, nullptr, 0, 1e-12f, KK_ACT_NONE, 0.f));
  }
  KK_TRY(c.dbg("bert_dur", x, hs));
  // ---- bert_encoder + DurationEncoder (kokoro.py:143-146, modules.py:392-411)
  const int S = cf.style_dim;
  Buf cat = c.act(Tmax, H + S);
  KK_TRY(c.conv(m->bert_encoder, x, lT, cat, lT, Tmax, plain));
  float* style_p = c.f32((size_t)B * m->Np);
  if (!c.dry) {
    KK_TRY(kk_launch_fill_style(ref_s, 128, cat.p, cat.bs, cat.ld, H, S, Tmax, lT, B, cat.dtype, c.st));
    KK_TRY(kk_launch_style_fc(ref_s, 128, m->sp_wT.p, m->sp_b.p, style_p, m->Np, B, c.st));
  }
  float* xproj = c.f32((size_t)B * Tmax * 4 * H);  // 8 * (H/2)
  Buf h = c.act(Tmax, H);
  for (int i = 0; i < cf.n_layer; ++i) {
    KK_TRY(c.lstm(m->dur_lstms[i], cat, 0, xproj, h, Tmax, lT));
    KK_TRY(c.layernorm(h, nullptr, cat, H, Tmax, lT, nullptr, nullptr, style_p + m->dur_adaln[i].off, m->Np, 1e-5f, KK_ACT_NONE, 0.f));
  }
  KK_TRY(c.dbg("d", cat, H + S));
  // ---- duration head (kokoro.py:147-150)
  KK_TRY(c.lstm(m->pred_lstm, cat, 0, xproj, h, Tmax, lT));
  int* pred_dur = c.i32((size_t)B * Tmax);
  float* dur_f = c.f32((size_t)B * Tmax);
  if (!c.dry) {
    KK_TRY(kk_launch_duration(h.p, h.bs, h.ld, m->dur_W.p, m->dur_b.p, H, cf.max_dur, speed, pred_dur, dur_f, Tmax, lT, B, h.dtype, c.st));
    Buf df;
    df.p = dur_f; df.ld = 1; df.bs = Tmax; df.rows = Tmax; df.dtype = KK_F32;
    KK_TRY(c.dbg("duration", df, 1));
    if (pred_dur_out &&
        hipMemcpyAsync(pred_dur_out, pred_dur, (size_t)B * Tmax * 4, hipMemcpyDeviceToDevice, c.st) != hipSuccess)
      return kk_fail("kk_forward_text: copy of pred_dur failed");
  }
  // ---- TextEncoder (modules.py:41-68): independent of everything above -> the side stream (its own input-projection scratch)
  Buf te = c.act(Tmax, H), te2 = c.act(Tmax, H), t_en = c.act(Tmax, H);
  float* xproj_te = c.f32((size_t)B * Tmax * 8 * (H / 2));
  KK_TRY(c.begin_side(0));
  if (!c.dry) KK_TRY(kk_launch_embedding(ids, m->te_emb.p, te.p, te.bs, te.ld, H, Tmax, lT, B, te.dtype, c.st));
  for (int i = 0; i < cf.n_layer; ++i) {
    ConvOpt o;
    o.pad = (cf.text_encoder_kernel_size - 1) / 2;
    KK_TRY(c.conv(m->te_cnn[i], te, lT, te2, lT, Tmax, o));
    KK_TRY(c.layernorm(te2, nullptr, te, H, Tmax, lT, m->te_ln_w[i].p, m->te_ln_b[i].p, nullptr, 0, 1e-5f, KK_ACT_LRELU, 0.2f));
  }
  KK_TRY(c.lstm(m->text_lstm, te, 0, xproj_te, t_en, Tmax, lT));
  KK_TRY(c.end_side(0));
  KK_TRY(c.join_side(0));
  KK_TRY(c.dbg("t_en", t_en, H));
  ts.d = cat;
  ts.t_en = t_en;
  ts.pred_dur = pred_dur;
  return 0;
}

int run_audio(Ctx& c, int Tmax, const int* lens, const float* ref_s, const int* dur, int Fmax, int noise_mode, const float* noise,
              uint64_t seed, const TextState& ts, float* wav_out, int* nframes_out) {
  kk_model* m = c.m;
  kk_context* cx = c.cx;
  (void)cx;
  const kk_config& cf = m->cfg;
  const int H = cf.hidden_dim, S = cf.style_dim, DH = cf.decoder_hidden, B = c.B;
  const int u0 = cf.upsample_rates[0], u1 = cf.upsample_rates[1], hop = cf.gen_istft_hop_size;
  const int L2 = 2 * Fmax, L20 = L2 * u0, Tf = L20 * u1 + 1, Nw = L20 * u1 * hop;
  const int C0 = cf.upsample_initial_channel, nk = cf.n_resblock_kernels;
  c.Fcap = Fmax;
  // ---- alignment + length regulation (kokoro.py:151-157,162)
  int* frame_idx = c.i32((size_t)B * Fmax);
  int* lenF = c.i32(B);
  int* lens4 = c.i32((size_t)4 * B);
  if (!c.dry) {
    KK_TRY(kk_launch_alignment(dur, Tmax, lens, frame_idx, lenF, Fmax, B, c.st));
    KK_TRY(kk_launch_lens(lenF, lens4, B, u0, u1, hop, c.st));
    if (nframes_out && hipMemcpyAsync(nframes_out, lenF, (size_t)B * 4, hipMemcpyDeviceToDevice, c.st) != hipSuccess)
      return kk_fail("kk_forward_audio: copy of nframes failed");
  }
  const KKLen lF{lenF, 1, 0}, l2{lenF, 2, 0}, l20{lenF, 2 * u0, 0}, lTf{lens4 + 2 * B, 1, 0}, lTfm1{lenF, 2 * u0 * u1, 0};
  Buf en = c.act(Fmax, H + S);
  const int ld514 = rup(H + 2, 64), ldcat = rup(DH + 2 + 64, 64);
  Buf cat514 = c.act(Fmax, ld514);
  if (!c.dry && hipMemsetAsync(cat514.p, 0, (size_t)B * cat514.bs * (c.adt == KK_F32 ? 4 : 2), c.st) != hipSuccess)
    return kk_fail("kk_forward_audio: memset failed");  // pad channels feed zero weights: they must be finite
  if (!c.dry) {
    KK_TRY(kk_launch_gather_rows(ts.d.p, ts.d.bs, ts.d.ld, frame_idx, Fmax, lenF, en.p, en.bs, en.ld, 0, H + S, B, en.dtype, c.st));
    KK_TRY(kk_launch_gather_rows(ts.t_en.p, ts.t_en.bs, ts.t_en.ld, frame_idx, Fmax, lenF, cat514.p, cat514.bs, cat514.ld, 0, H, B,
                                 cat514.dtype, c.st));
  }
  KK_TRY(c.dbg("en", en, H + S));
  KK_TRY(c.dbg("asr", cat514, H));
  float* style_p = c.f32((size_t)B * m->Np);
  float* style_d = c.f32((size_t)B * m->Nd);
  if (!c.dry) {
    KK_TRY(kk_launch_style_fc(ref_s, 128, m->sp_wT.p, m->sp_b.p, style_p, m->Np, B, c.st));
    KK_TRY(kk_launch_style_fc(ref_s, 0, m->sd_wT.p, m->sd_b.p, style_d, m->Nd, B, c.st));
  }
  // statistics scratch, sized for the largest (rows, channels) pair normalised below
  {
    // upper bounds: partial = B * chunks(L) * 2 * C for every (L, C) pair used below
    size_t pmax = 0, cmax = 0;
    auto upd = [&](int L, int C) {
      pmax = std::max(pmax, kk_stats_partial_floats(B, C, L, 64));  // the smallest chunk any domain uses
      cmax = std::max(cmax, (size_t)B * C);
    };
    upd(Fmax, H); upd(L2, H); upd(L2, H / 2); upd(Fmax, H + 2); upd(Fmax, DH + 2 + 64); upd(Fmax, DH); upd(L2, DH + 2 + 64);
    upd(L2, H); upd(L20, C0 / 2); upd(Tf, C0 / 4);
    c.st_partial = c.f32(pmax);
    c.st_mean = c.f32(cmax);
    c.st_rstd = c.f32(cmax);
    // fused-norm scratch of the bf16 MFMA path
    c.fz_stride = rup(std::max(std::max(DH + 2 + 64, H + S), C0), 64);
    // (+ 16 tiles: a polyphase transposed conv has ceil(Q / rows) tiles PER PHASE, up to stride - 1 more than ceil(L / rows) in total)
    const size_t tiles = std::max(std::max((size_t)(kk_cdiv(Tf, 128) + 16) * (C0 / 4), (size_t)(kk_cdiv(L20, 128) + 16) * (C0 / 2)),
                                  (size_t)(kk_cdiv(L2, 128) + 16) * std::max(DH, H));
    c.fz_part_floats = (size_t)B * tiles * 2;
    float* fp = c.f32(c.fz_part_floats);
    float* p0 = c.f32((size_t)B * c.fz_stride);
    float* p1 = c.f32((size_t)B * c.fz_stride);
    float* p2 = c.f32((size_t)B * c.fz_stride);
    float* p3 = c.f32((size_t)B * c.fz_stride);
    if (c.adt == KK_BF16) { c.fz_part = fp; c.fz_pa[0] = p0; c.fz_pb[0] = p1; c.fz_pa[1] = p2; c.fz_pb[1] = p3; }
  }
  // ---- F0Ntrain (modules.py:355-377)
  float* xprojF = c.f32((size_t)B * Fmax * 4 * H);
  Buf xs = c.act(Fmax, H);
  KK_TRY(c.lstm(m->shared_lstm, en, 0, xprojF, xs, Fmax, lF));
  Buf pA = c.act(L2, H), pB = c.act(L2, H), pC = c.act(L2, H), y0 = c.act(Fmax, H), y1 = c.act(L2, H / 2), y2 = c.act(L2, H / 2);
  Buf f0n[2];
  for (int which = 0; which < 2; ++which) {
    const ResBlk1d* blk = which == 0 ? m->f0blk : m->nblk;
    KK_TRY(run_resblk1d(c, blk[0], xs, lF, Fmax, y0, style_p, m->Np, pA, pB, pC));
    KK_TRY(run_resblk1d(c, blk[1], y0, lF, Fmax, y1, style_p, m->Np, pA, pB, pC));
    KK_TRY(run_resblk1d(c, blk[2], y1, l2, L2, y2, style_p, m->Np, pA, pB, pC));
    f0n[which] = c.act(L2, 1, KK_F32);  // the F0 / N curves stay fp32 in every mode (phase accuracy)
    ConvOpt o;
    KK_TRY(c.conv(which == 0 ? m->f0_proj : m->n_proj, y2, l2, f0n[which], l2, L2, o));
  }
  KK_TRY(c.dbg("F0_pred", f0n[0], 1));
  KK_TRY(c.dbg("N_pred", f0n[1], 1));
  KK_TRY(c.fork_point(1));  // the harmonic source (below, after the decoder) needs the F0 curve only
  // ---- Decoder (istftnet.py:947-963)
  Buf catA = c.act(Fmax, ldcat), catB = c.act(Fmax, ldcat);
  if (!c.dry) {  // pad channels of the concat buffers feed zero weights: they must be finite
    const size_t es = c.adt == KK_F32 ? 4 : 2;
    if (hipMemsetAsync(catA.p, 0, (size_t)B * catA.bs * es, c.st) != hipSuccess || hipMemsetAsync(catB.p, 0, (size_t)B * catB.bs * es, c.st) != hipSuccess)
      return kk_fail("kk_forward_audio: memset failed");
  }
  {
    ConvOpt o;
    o.stride = 2;
    o.pad = 1;
    Buf dsts[3] = {cat514.slice(H), catA.slice(DH + 64), catB.slice(DH + 64)};
    for (int k = 0; k < 3; ++k) {
      KK_TRY(c.conv(m->f0_conv, f0n[0], l2, dsts[k], lF, Fmax, o));
      KK_TRY(c.conv(m->n_conv, f0n[1], l2, dsts[k].slice(1), lF, Fmax, o));
    }
    ConvOpt p;
    Buf a1 = catA.slice(DH), a2 = catB.slice(DH);
    KK_TRY(c.conv(m->asr_res, cat514, lF, a1, lF, Fmax, p));
    KK_TRY(c.conv(m->asr_res, cat514, lF, a2, lF, Fmax, p));
  }
  Buf dA = c.act(L2, ldcat), dB = c.act(L2, DH), dC = c.act(L2, DH);
  KK_TRY(run_resblk1d(c, m->enc, cat514, lF, Fmax, catA, style_d, m->Nd, dA, dB, dC));
  KK_TRY(c.dbg("dec_encode", catA, DH));
  KK_TRY(run_resblk1d(c, m->dec[0], catA, lF, Fmax, catB, style_d, m->Nd, dA, dB, dC));
  KK_TRY(run_resblk1d(c, m->dec[1], catB, lF, Fmax, catA, style_d, m->Nd, dA, dB, dC));
  KK_TRY(run_resblk1d(c, m->dec[2], catA, lF, Fmax, catB, style_d, m->Nd, dA, dB, dC));
  Buf gx = c.act(L2, H);
  KK_TRY(run_resblk1d(c, m->dec[3], catB, lF, Fmax, gx, style_d, m->Nd, dA, dB, dC));
  KK_TRY(c.dbg("dec_out", gx, H));
  // ---- Generator front end (istftnet.py:770-775)
  float* phase = c.f32((size_t)B * 9 * L2);
  float* har_source = c.f32((size_t)B * Nw);
  // bf16: pitch 64 so the k=1 noise conv can run on the MFMA kernel; 16 spare (zero) rows so the strided noise convs can
  // read whole row groups (Packer::strided_rows)
  Buf har = c.act(c.adt == KK_BF16 ? Tf + 16 : Tf, c.adt == KK_BF16 ? 64 : 24);
  KK_TRY(c.begin_side(1));  // source + STFT beside the decoder blocks enqueued above
  if (!c.dry && c.adt == KK_BF16 && hipMemsetAsync(har.p, 0, (size_t)B * har.bs * 2, c.st) != hipSuccess) return kk_fail("kk_forward_audio: memset failed");
  if (!c.dry) {
    KKSourceArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.f0 = (const float*)f0n[0].p; sa.L2max = L2; sa.len2 = lens4; sa.phase = phase; sa.lin_w = m->lin_w.p; sa.lin_b = m->lin_b;
    sa.noise = noise; sa.seed = seed; sa.seed_dev = cx->capturing ? cx->seed_dev : nullptr; sa.noise_mode = noise_mode; sa.har_source = har_source; sa.Nmax = Nw; sa.upsample = u0 * u1 * hop;
    c.prof_start();
    KK_TRY(kk_launch_source(sa, B, c.st));
    c.prof_stop(8, 0.0, (double)B * Nw * 4.0);
    Buf hsb;
    hsb.p = har_source; hsb.ld = 1; hsb.bs = Nw; hsb.rows = Nw; hsb.dtype = KK_F32;
    KK_TRY(c.dbg("har_source", hsb, 1));
    c.prof_start();
    KK_TRY(kk_launch_stft20(har_source, Nw, lens4 + 3 * B, har.p, har.bs, har.ld, Tf, B, har.dtype, c.st));
    c.prof_stop(9, 880.0 * B * Tf, (double)B * (Nw * 4.0 + Tf * 22.0 * Ctx::esz(har.dtype)));
  }
  KK_TRY(c.end_side(1));
  KK_TRY(c.join_side(1));
  KK_TRY(c.dbg("har", har, 22));
  // ---- up-sampling stages (istftnet.py:776-796)
  // The fused head will run (bf16 mode) and nobody looks at gen_stage1 / overrides a stage: the LeakyReLU(0.01) that conv_post applies to its
  // input (istftnet.py:797) moves into the epilogue of the conv that finishes the generator's last stage, and the head's slab staging --
  // a quarter of its vector instructions (profiles/r03_head_pmc.json) -- becomes a copy.
  const bool post_lrelu = m->head_wf && !cx->no_head_fusion && !cx->force_generic && !cx->no_fusion && c.adt == KK_BF16 && !cx->keep_debug && cx->dbg_over.empty() &&
                          !getenv("KK_NO_POST_LRELU");
  Buf cur = gx;           // input of ups[i]
  KKLen lcur = l2;
  int Lcur = L2;
  for (int i = 0; i < cf.n_upsamples; ++i) {
    const int Cst = C0 >> (i + 1);
    const bool last = i == cf.n_upsamples - 1;
    const int Lst = last ? Tf : L20;
    const KKLen lst = last ? lTf : l20;
    Buf xsrc = c.act(Lst, Cst), t1 = c.act(Lst, Cst), t2 = c.act(Lst, Cst), xi = c.act(Lst, Cst), yb = c.act(Lst, Cst), accb = c.act(Lst, Cst);
    ConvOpt on;
    if (!last) {
      int sf0 = 1;
      for (int j = i + 1; j < cf.n_upsamples; ++j) sf0 *= cf.upsample_rates[j];
      on.stride = sf0;
      on.pad = (sf0 + 1) / 2;
    }
    const bool fuse_ok = c.adt == KK_BF16 && c.fz_part && !cx->no_fusion && !cx->force_generic;
    bool xsrc_stats = false, xi_stats = false;
    const ConvW& nrows = m->noise_conv_rows[i];
    if (!last && nrows.mfma && c.adt == KK_BF16 && !cx->force_generic && har.ld == 64 && on.stride <= 16 && nrows.Cin == on.stride * har.ld) {
      // stride-1 form over groups of `stride` rows (see Packer::strided_rows): valid groups = ceil(len / stride) = lst + 1
      Buf hg = har;
      hg.ld = on.stride * har.ld;
      hg.rows = har.rows / on.stride;
      ConvOpt og;
      og.pad = m->noise_rows_pad[i];
      og.alg_taps_cin = (double)m->noise_conv[i].Kw * m->noise_conv[i].Cin;
      const KKLen lg = {lst.len, lst.mul, lst.add + 1};
      og.want_stats = xsrc_stats = fuse_ok && c.can_mfma(nrows, hg, xsrc, og);
      KK_TRY(c.conv(nrows, hg, lg, xsrc, lst, Lst, og));
    } else {
      on.want_stats = xsrc_stats = fuse_ok && c.can_mfma(m->noise_conv[i], har, xsrc, on);
      KK_TRY(c.conv(m->noise_conv[i], har, lTf, xsrc, lst, Lst, on));
    }
    // the instance-norm statistics of the tensors the resblocks start from come out of their producers' epilogues
    if (xsrc_stats) KK_TRY(c.finalize(Cst, lst, nullptr, 0, c.st_mean, c.st_rstd, nullptr, nullptr));
    KK_TRY(run_resblock1(c, m->noise_res[i], xsrc, xsrc, t1, t2, Lst, lst, style_d, m->Nd, nullptr, 1.f, 0, xsrc_stats));
    ConvOpt ou;
    ou.mode = KK_CONVT;
    ou.stride = cf.upsample_rates[i];
    ou.pad = (cf.upsample_kernel_sizes[i] - cf.upsample_rates[i]) / 2;
    ou.in_slope = 0.1f;
    const int Qt = kk_cdiv(Lcur * cf.upsample_rates[i], cf.upsample_rates[i]);
    if (!last) {
      ou.res = &xsrc;
      ou.want_stats = xi_stats = fuse_ok && c.can_mfma(m->ups[i], cur, xi, ou);
      KK_TRY(c.conv(m->ups[i], cur, lcur, xi, lst, Qt, ou));
      if (xi_stats) KK_TRY(c.finalize(Cst, lst, nullptr, 0, c.st_mean, c.st_rstd, nullptr, nullptr));
    } else {
      // zero left pad of one frame (istftnet.py:786-787, "ReflectionPad1d" = mx.pad) then + x_source: the transposed conv
      // writes one row further down with x_source (same offset) as its residual; row 0 is x_source[0] alone.
      if (!c.dry)
        KK_TRY(kk_launch_copy_slice(xsrc.p, xsrc.bs, xsrc.ld, xi.p, xi.bs, xi.ld, 0, Cst, 1, KKLen{lenF, 1, 0}, B, xi.dtype, c.st));
      Buf xi1 = xi.row_offset(1), xs1 = xsrc.row_offset(1);
      ou.res = &xs1;
      ou.want_stats = xi_stats = fuse_ok && c.can_mfma(m->ups[i], cur, xi1, ou);
      KK_TRY(c.conv(m->ups[i], cur, lcur, xi1, lTfm1, Qt, ou));
      if (xi_stats) {  // rows 1.. come from the conv's epilogue, row 0 (x_source[0], copied above) is added to tile 0
        if (!c.dry) KK_TRY(kk_launch_stat_add_row(xi.p, xi.bs, c.fz_part, c.last_ntiles, Cst, B, c.st));
        KK_TRY(c.finalize(Cst, lst, nullptr, 0, c.st_mean, c.st_rstd, nullptr, nullptr));
      }
    }
    KK_TRY(c.dbg(i == 0 ? "gen_pre_res0" : "gen_pre_res1", xi, Cst));
    for (int j = 0; j < nk; ++j)
      KK_TRY(run_resblock1(c, m->resblocks[i * nk + j], xi, yb, t1, t2, Lst, lst, style_d, m->Nd, &accb, 1.0f / (float)nk, j > 0 ? 1 : 0, j > 0 || xi_stats,
                           (last && j == nk - 1 && post_lrelu) ? 0.01f : 1.f));
    KK_TRY(c.dbg(i == 0 ? "gen_stage0" : "gen_stage1", accb, Cst));
    cur = accb;
    lcur = lst;
    Lcur = Lst;
  }
  // ---- conv_post + iSTFT head (istftnet.py:798-806)
  // (post_lrelu: the LeakyReLU(0.01) in front of conv_post was applied by the epilogue of the last resblock conv -- the head then stages raw rows)
  Buf cp = c.act(Tf, 24);
  if (m->head_wf && !cx->no_head_fusion && !cx->force_generic && cur.dtype == KK_BF16 && cur.ld % 8 == 0 && cur.ld >= m->conv_post.Cin &&
      !((uintptr_t)cur.p & 15) && cx->dbg_over.find("conv_post") == cx->dbg_over.end()) {
    // one kernel: the 22-channel tensor never exists in HBM (kk_head.hip); `keep_debug` also writes it for kk_debug_fetch
    if (!c.dry) {
      KKHeadArgs h;
      memset(&h, 0, sizeof h);
      h.x = (const bf16_t*)cur.p; h.xbs = cur.bs; h.ldx = cur.ld; h.wf = m->head_wf; h.bias = m->conv_post.b; h.in_slope = post_lrelu ? 1.0f : 0.01f;
      h.len_frames = lens4 + 2 * B; h.Tfmax = Tf; h.wav = wav_out; h.wbs = (long long)Nw;
      if (cx->keep_debug) { h.cp_out = (bf16_t*)cp.p; h.cp_bs = cp.bs; h.cp_ld = cp.ld; }
      for (int n = 0; n < 20; ++n) h.hann_per[n] = (float)(0.5 * (1.0 - cos(2.0 * 3.14159265358979323846 * n / 20.0)));
      if (cx->keep_debug && hipMemsetAsync(cp.p, 0, (size_t)B * cp.bs * 2, c.st) != hipSuccess) return kk_fail("kk_forward_audio: memset failed");
      c.prof_start();
      KK_TRY(kk_launch_conv_post_istft(h, B, c.st));
      // algorithmic bytes of the fused head: 128 bf16 channels in + 5 fp32 samples out per frame column
      c.prof_stop(5, B * (double)Tf * (2.0 * 22 * 128 * 7 + 800.0), (double)B * Tf * (128.0 * 2.0 + 20.0));
    }
    KK_TRY(c.dbg("conv_post", cp, 22));
    return 0;
  }
  ConvOpt op;
  op.pad = 3;
  op.in_slope = post_lrelu ? 1.0f : 0.01f;
  op.pad_out_ok = true;  // cp has pitch 24 for 22 channels
  KK_TRY(c.conv(m->conv_post, cur, lTf, cp, lTf, Tf, op));
  KK_TRY(c.dbg("conv_post", cp, 22));
  if (!c.dry) {
    c.prof_start();
    KK_TRY(kk_launch_istft_head(cp.p, cp.bs, cp.ld, lens4 + 2 * B, Tf, wav_out, (long long)Nw, B, cp.dtype, c.adt == KK_BF16 ? 1 : 0, c.st));
    // algorithmic bytes: 22 inputs per frame column + 5 fp32 samples per frame column
    c.prof_stop(5, 800.0 * B * Tf, (double)B * Tf * (22.0 * Ctx::esz(cp.dtype) + 20.0));
  }
  return 0;
}

int check_common(kk_model* m, int B, int Tmax, const char* who) {
  if (!m) return kk_fail("null model");
  if (!m->finalized) return failf("%s: kk_finalize has not been called", who);
  if (B <= 0 || Tmax <= 0 || Tmax > 512 || Tmax > m->cfg.plbert_max_pos) return failf("%s: need 0 < B, 0 < Tmax <= 512 (kokoro.py:131-134)", who);
  return 0;
}

}  // namespace

static size_t plan_bytes(const kk_model* m, kk_context* cx, int B, int Tmax, int Fmax) {
  if (!m || !m->finalized || B <= 0 || Tmax <= 0) return 0;
  Ctx c;
  c.m = const_cast<kk_model*>(m); c.cx = cx; c.st = nullptr; c.dry = true; c.base = nullptr; c.cap = 0; c.used = 0;
  c.B = B;
  c.adt = m->adt;
  TextState ts;
  if (run_text(c, Tmax, nullptr, nullptr, nullptr, nullptr, ts, nullptr) != 0) return 0;
  if (Fmax > 0 && run_audio(c, Tmax, nullptr, nullptr, nullptr, Fmax, 0, nullptr, 0, ts, nullptr, nullptr) != 0) return 0;
  return c.used + 256;
}
// (a context with default switches; the debug switches of a real context can change which intermediates exist: kk_context_workspace_bytes)
extern "C" size_t kk_workspace_bytes(const kk_model* m, int B, int Tmax, int Fmax) {
  kk_context def;
  def.m = const_cast<kk_model*>(m);
  return plan_bytes(m, &def, B, Tmax, Fmax);
}
extern "C" size_t kk_context_workspace_bytes(const kk_context* cx, int B, int Tmax, int Fmax) {
  return cx ? plan_bytes(cx->m, const_cast<kk_context*>(cx), B, Tmax, Fmax) : 0;
}

static int make_ctx(kk_context* cx, void* stream, int B, void* ws, size_t ws_bytes, Ctx& c) {
  kk_model* m = cx->m;
  c.m = m; c.cx = cx; c.st = (hipStream_t)stream; c.dry = false; c.base = (char*)ws; c.cap = ws_bytes; c.used = 0; c.B = B; c.adt = m->adt;
  if (!ws) return kk_fail("workspace is null");
  if (((uintptr_t)ws & 255) != 0) return kk_fail("workspace must be 256-byte aligned");
  return 0;
}

extern "C" int kk_forward_text(kk_context* cx, void* stream, int B, int Tmax, const int32_t* ids, const int32_t* lens, const float* ref_s,
                               const float* speed, void* workspace, size_t workspace_bytes, int32_t* pred_dur_out) {
  if (!cx) return kk_fail("kk_forward_text: null context");
  kk_model* m = cx->m;
  KK_TRY(check_common(m, B, Tmax, "kk_forward_text"));
  if (!ids || !lens || !ref_s || !speed) return kk_fail("kk_forward_text: null input");
  if (workspace_bytes < kk_context_workspace_bytes(cx, B, Tmax, 0)) return kk_fail("kk_forward_text: workspace too small");
  Ctx c;
  KK_TRY(make_ctx(cx, stream, B, workspace, workspace_bytes, c));
  TextState ts;
  return run_text(c, Tmax, ids, lens, ref_s, speed, ts, pred_dur_out);
}

extern "C" int kk_forward_audio(kk_context* cx, void* stream, int B, int Tmax, const int32_t* lens, const float* ref_s, const int32_t* dur,
                                int Fmax, int noise_mode, const float* sine_noise, uint64_t seed, void* workspace, size_t workspace_bytes,
                                float* wav_out, int32_t* nframes_out) {
  if (!cx) return kk_fail("kk_forward_audio: null context");
  kk_model* m = cx->m;
  KK_TRY(check_common(m, B, Tmax, "kk_forward_audio"));
  if (!lens || !ref_s || !dur || !wav_out || Fmax <= 0) return kk_fail("kk_forward_audio: bad argument");
  if (noise_mode == KK_NOISE_INJECTED && !sine_noise) return kk_fail("kk_forward_audio: KK_NOISE_INJECTED needs sine_noise");
  if (workspace_bytes < kk_context_workspace_bytes(cx, B, Tmax, Fmax)) return kk_fail("kk_forward_audio: workspace too small");
  // replay the text stage's allocation plan (no launches) to find where its results live in the workspace
  Ctx c;
  KK_TRY(make_ctx(cx, stream, B, workspace, workspace_bytes, c));
  TextState ts;
  c.dry = true;
  KK_TRY(run_text(c, Tmax, nullptr, nullptr, nullptr, nullptr, ts, nullptr));
  c.dry = false;
  return run_audio(c, Tmax, lens, ref_s, dur, Fmax, noise_mode, sine_noise, seed, ts, wav_out, nframes_out);
}

extern "C" int kk_forward(kk_context* cx, void* stream, int B, int Tmax, const int32_t* ids, const int32_t* lens, const float* ref_s,
                          const float* speed, const int32_t* forced_dur, int Fmax, int noise_mode, const float* sine_noise, uint64_t seed,
                          void* workspace, size_t workspace_bytes, float* wav_out, int32_t* pred_dur_out, int32_t* nframes_out) {
  if (!cx) return kk_fail("kk_forward: null context");
  kk_model* m = cx->m;
  KK_TRY(check_common(m, B, Tmax, "kk_forward"));
  if (!ids || !lens || !ref_s || !speed || !wav_out || Fmax <= 0) return kk_fail("kk_forward: bad argument");
  if (noise_mode == KK_NOISE_INJECTED && !sine_noise) return kk_fail("kk_forward: KK_NOISE_INJECTED needs sine_noise");
  if (workspace_bytes < kk_context_workspace_bytes(cx, B, Tmax, Fmax)) return kk_fail("kk_forward: workspace too small");
  auto eager = [&](void* on_stream) -> int {
    Ctx c;
    KK_TRY(make_ctx(cx, on_stream, B, workspace, workspace_bytes, c));
    TextState ts;
    KK_TRY(run_text(c, Tmax, ids, lens, ref_s, speed, ts, pred_dur_out));
    return run_audio(c, Tmax, lens, ref_s, forced_dur ? forced_dur : ts.pred_dur, Fmax, noise_mode, sine_noise, seed, ts, wav_out, nframes_out);
  };
  // ---- graph replay: ~450 launches become one hipGraphLaunch.  Only for the plain forward: no debug overrides, no profiling.
  if (!cx->graph_mode || cx->prof_on || !cx->dbg_over.empty()) return eager(stream);
  hipStream_t st = (hipStream_t)stream;
  const std::vector<unsigned long long> key = {(unsigned long long)B, (unsigned long long)Tmax, (unsigned long long)Fmax,
      (unsigned long long)noise_mode, (unsigned long long)(uintptr_t)ids, (unsigned long long)(uintptr_t)lens,
      (unsigned long long)(uintptr_t)ref_s, (unsigned long long)(uintptr_t)speed, (unsigned long long)(uintptr_t)forced_dur,
      (unsigned long long)(uintptr_t)sine_noise, (unsigned long long)(uintptr_t)workspace, (unsigned long long)workspace_bytes,
      (unsigned long long)(uintptr_t)wav_out, (unsigned long long)(uintptr_t)pred_dur_out, (unsigned long long)(uintptr_t)nframes_out,
      (unsigned long long)cx->force_generic, (unsigned long long)cx->no_fusion, (unsigned long long)cx->no_v4,
      (unsigned long long)cx->no_head_fusion, (unsigned long long)cx->keep_debug, (unsigned long long)cx->v5_mode, (unsigned long long)cx->no_side};
  kk_context::GraphEntry* ge = nullptr;
  for (auto& g : cx->graphs)
    if (g.key == key) ge = &g;
  if (!ge) {
    if (cx->graphs.size() >= 16) {  // drop the oldest entry
      if (cx->graphs.front().exec) (void)hipGraphExecDestroy(cx->graphs.front().exec);
      if (cx->graphs.front().graph) (void)hipGraphDestroy(cx->graphs.front().graph);
      cx->graphs.erase(cx->graphs.begin());
    }
    cx->graphs.emplace_back();
    ge = &cx->graphs.back();
    ge->key = key;
  }
  if (ge->seen == 0) {  // first sight of this argument tuple: run eagerly (one-time attribute calls must not land in a capture)
    ge->seen = 1;
    return eager(stream);
  }
  if (!cx->seed_dev && hipMalloc((void**)&cx->seed_dev, 8) != hipSuccess) return kk_fail("kk_forward: hipMalloc(seed) failed");
  if (ge->seen == 1) {
    // capture on a private stream (the caller's may be the legacy default stream, which cannot be captured); nothing runs
    // during capture, and the instantiated graph is launched on the caller's stream
    if (!cx->cap_stream && hipStreamCreateWithFlags(&cx->cap_stream, hipStreamNonBlocking) != hipSuccess)
      return kk_fail("kk_forward: hipStreamCreate failed");
    if (hipStreamBeginCapture(cx->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
      return kk_fail("kk_forward: hipStreamBeginCapture failed");
    cx->capturing = true;
    const int rc = eager((void*)cx->cap_stream);
    cx->capturing = false;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(cx->cap_stream, &g);
    if (rc != 0) {
      if (g) (void)hipGraphDestroy(g);
      return rc;
    }
    if (e != hipSuccess || !g) return kk_fail("kk_forward: hipStreamEndCapture failed");
    hipGraphExec_t ex = nullptr;
    if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGraphDestroy(g);
      return kk_fail("kk_forward: hipGraphInstantiate failed");
    }
    ge->graph = g;
    ge->exec = ex;
    ge->seen = 2;
  }
  // the seed is the one by-value argument that changes between replays: it travels through device memory
  KK_TRY(kk_launch_set_u64(cx->seed_dev, seed, st));
  if (hipGraphLaunch(ge->exec, st) != hipSuccess) return kk_fail("kk_forward: hipGraphLaunch failed");
  return 0;
}

extern "C" int kk_set_graph_mode(kk_context* cx, int on) {
  if (!cx) return kk_fail("kk_set_graph_mode: null context");
  cx->graph_mode = on != 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// single-kernel entry points
// ------------------------------------------------------------------------------------------------
extern "C" int kk_op_conv1d(void* stream, int B, const void* x, int ldx, int Lin_rows, const int32_t* lin, const float* w_packed, int ldw,
                            const float* bias, int Cin, int Cout, int Kw, int transposed, int stride, int pad, int dil, int in_shift,
                            float in_slope, int act, float act_slope, const void* res, int ldr, float scale, int accumulate, void* out,
                            int ldo, int Lout_rows, const int32_t* lout, int in_dtype, int out_dtype) {
  KKConvArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.xbs = (long long)Lin_rows * ldx; a.ldx = ldx; a.w = w_packed; a.ldw = ldw; a.bias = bias;
  a.out = out; a.obs = (long long)Lout_rows * ldo; a.ldo = ldo;
  a.res = res; a.rbs = (long long)Lout_rows * ldr; a.ldr = ldr;
  a.Cin = Cin; a.Cout = Cout; a.Kw = Kw; a.mode = transposed ? KK_CONVT : KK_CONV; a.stride = stride; a.pad = pad; a.dil = dil;
  a.in_shift = in_shift; a.Q = transposed ? kk_cdiv(Lout_rows, stride) : Lout_rows; a.Lo_rows = Lout_rows;
  a.lin = KKLen{lin, lin ? 1 : 0, lin ? 0 : Lin_rows};
  a.lout = KKLen{lout, lout ? 1 : 0, lout ? 0 : Lout_rows};
  a.in_slope = in_slope; a.scale = scale; a.accumulate = accumulate; a.act = act; a.act_slope = act_slope;
  return kk_launch_conv_generic(a, B, in_dtype, out_dtype, (hipStream_t)stream);
}

extern "C" int kk_op_conv1d_bf16(void* stream, int B, const void* x, int ldx, int Lin_rows, const int32_t* lin, const void* w_bf16, int CinP,
                                 int CoutP, const float* bias, int Cout, int Kw, int transposed, int stride, int pad, int dil, int in_shift,
                                 float in_slope, int act, float act_slope, const void* res, int ldr, float scale, int accumulate, void* out,
                                 int ldo, int Lout_rows, const int32_t* lout, int out_dtype) {
  KKMfmaArgs g;
  memset(&g, 0, sizeof g);
  g.x = (const bf16_t*)x; g.xbs = (long long)Lin_rows * ldx; g.ldx = ldx; g.w = (const bf16_t*)w_bf16; g.CinP = CinP; g.CoutP = CoutP; g.bias = bias;
  g.out = out; g.obs = (long long)Lout_rows * ldo; g.ldo = ldo; g.res = res; g.rbs = (long long)Lout_rows * ldr; g.ldr = ldr;
  g.Cout = Cout; g.Kw = Kw; g.mode = transposed ? KK_CONVT : KK_CONV; g.stride = stride; g.pad = pad; g.dil = dil; g.in_shift = in_shift;
  g.Q = transposed ? kk_cdiv(Lout_rows, stride) : Lout_rows; g.Lo_rows = Lout_rows;
  g.lin = KKLen{lin, lin ? 1 : 0, lin ? 0 : Lin_rows};
  g.lout = KKLen{lout, lout ? 1 : 0, lout ? 0 : Lout_rows};
  g.in_slope = in_slope; g.scale = scale; g.accumulate = accumulate; g.act = act; g.act_slope = act_slope;
  if (!kk_mfma_eligible(CinP, Cout, Kw, g.mode, stride, dil)) return kk_fail("kk_op_conv1d_bf16: shape not eligible for the MFMA kernel");
  if (g_op_wfrag && out_dtype == KK_BF16) {
    g.wf = (const bf16_t*)g_op_wfrag;
    if (g_op_variant == 5 && kk_mfma5_eligible(g, out_dtype)) return kk_launch_conv_mfma5(g, B, out_dtype, (hipStream_t)stream);
    return kk_launch_conv_mfma4(g, B, out_dtype, (hipStream_t)stream);
  }
  return kk_launch_conv_mfma(g, B, out_dtype, (hipStream_t)stream);
}

// the fused form used inside the bf16 generator: AdaIN + activation applied to the input while it is staged
// (y = act(x * nrm_a[b][c] + nrm_b[b][c])), per-tile column sums of the stored output written to stat_part
extern "C" int kk_op_conv1d_bf16_fused(void* stream, int B, const void* x, int ldx, int L_rows, const int32_t* len, const void* w_bf16, int CinP,
                                       int CoutP, const float* bias, int Cin, int Cout, int Kw, int pad, int dil, const float* nrm_a,
                                       const float* nrm_b, int nrm_stride, int nrm_act, float nrm_slope, const float* nrm_alpha,
                                       const void* res, int ldr, float scale, void* out, int ldo, float* stat_part, int* stat_ntiles_out) {
  KKMfmaArgs g;
  memset(&g, 0, sizeof g);
  g.x = (const bf16_t*)x; g.xbs = (long long)L_rows * ldx; g.ldx = ldx; g.w = (const bf16_t*)w_bf16; g.CinP = CinP; g.Cin = Cin; g.CoutP = CoutP; g.bias = bias;
  g.out = out; g.obs = (long long)L_rows * ldo; g.ldo = ldo; g.res = res; g.rbs = (long long)L_rows * ldr; g.ldr = ldr;
  g.Cout = Cout; g.Kw = Kw; g.mode = KK_CONV; g.stride = 1; g.pad = pad; g.dil = dil; g.Q = L_rows; g.Lo_rows = L_rows;
  g.lin = KKLen{len, len ? 1 : 0, len ? 0 : L_rows};
  g.lout = g.lin;
  g.in_slope = 1.f; g.scale = scale;
  g.nrm_a = nrm_a; g.nrm_b = nrm_b; g.nrm_stride = nrm_stride; g.nrm_act = nrm_act; g.nrm_slope = nrm_slope; g.nrm_alpha = nrm_alpha; g.nrm_C = Cin;
  g.stat_part = stat_part;
  g.stat_ntiles = kk_cdiv(L_rows, kk_mfma_stat_tile_rows(g, KK_BF16));
  if (stat_ntiles_out) *stat_ntiles_out = g.stat_ntiles;
  if (!kk_mfma_eligible(CinP, Cout, Kw, KK_CONV, 1, dil)) return kk_fail("kk_op_conv1d_bf16_fused: shape not eligible for the MFMA kernel");
  if (g_op_wfrag) {
    g.wf = (const bf16_t*)g_op_wfrag;
    if (g_op_variant == 5 && kk_mfma5_eligible(g, KK_BF16)) return kk_launch_conv_mfma5(g, B, KK_BF16, (hipStream_t)stream);
    return kk_launch_conv_mfma4(g, B, KK_BF16, (hipStream_t)stream);
  }
  return kk_launch_conv_mfma(g, B, KK_BF16, (hipStream_t)stream);
}

extern "C" int kk_op_adain(void* stream, int B, const void* x, int ldx, int L_rows, const int32_t* len, int C, const float* gamma_beta,
                           int gbs, int act, float slope, const float* alpha, int pool, const float* pool_w, const float* pool_b, void* out,
                           int ldo, int Cpad, int Lout_rows, float* scratch, size_t scratch_floats, int dtype, int fast) {
  const size_t np = kk_stats_partial_floats(B, C, L_rows, 512);
  if (scratch_floats < np + 2 * (size_t)B * C) return kk_fail("kk_op_adain: scratch too small");
  KKStatsArgs s;
  memset(&s, 0, sizeof s);
  s.x = x; s.xbs = (long long)L_rows * ldx; s.ldx = ldx; s.C = C; s.Lmax = L_rows; s.len = KKLen{len, len ? 1 : 0, len ? 0 : L_rows};
  s.partial = scratch; s.rows_per_chunk = 512; s.mean = scratch + np; s.rstd = scratch + np + (size_t)B * C; s.eps = 1e-5f;
  KK_TRY(kk_launch_instnorm_stats(s, B, dtype, (hipStream_t)stream));
  KKAdainArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.xbs = s.xbs; a.ldx = ldx; a.out = out; a.obs = (long long)Lout_rows * ldo; a.ldo = ldo; a.C = C; a.Cpad = Cpad;
  a.Lmax_out = Lout_rows; a.len_in = s.len; a.mean = s.mean; a.rstd = s.rstd; a.gb = gamma_beta; a.gbs = gbs; a.act = act; a.slope = slope;
  a.alpha = alpha; a.pool = pool; a.pool_w = pool_w; a.pool_b = pool_b; a.fast = fast;
  return kk_launch_adain_act(a, B, dtype, (hipStream_t)stream);
}

extern "C" int kk_op_layernorm(void* stream, int B, const void* x, int ldx, const void* res, int ldr, int L_rows, const int32_t* len, int C,
                               const float* w, const float* b, const float* gamma_beta, int gbs, float eps, int act, float slope, void* out,
                               int ldo, int dtype) {
  KKLnArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.xbs = (long long)L_rows * ldx; a.ldx = ldx; a.res = res; a.rbs = (long long)L_rows * ldr; a.ldr = ldr;
  a.out = out; a.obs = (long long)L_rows * ldo; a.ldo = ldo; a.C = C; a.Lmax = L_rows; a.len = KKLen{len, len ? 1 : 0, len ? 0 : L_rows};
  a.w = w; a.bias = b; a.gb = gamma_beta; a.gbs = gbs; a.eps = eps; a.act = act; a.slope = slope;
  return kk_launch_layernorm(a, B, dtype, (hipStream_t)stream);
}

extern "C" int kk_op_lstm(void* stream, int B, const float* xproj, const float* whT, int H, int L_rows, const int32_t* len, void* out,
                          int ldo, int dtype) {
  KKLstmArgs a;
  memset(&a, 0, sizeof a);
  a.xproj = xproj; a.whT = whT; a.out = out; a.obs = (long long)L_rows * ldo; a.ldo = ldo; a.H = H; a.Lmax = L_rows;
  a.len = KKLen{len, len ? 1 : 0, len ? 0 : L_rows};
  return kk_launch_lstm(a, B, dtype, (hipStream_t)stream);
}

extern "C" int kk_op_lstm_bf16(void* stream, int B, const float* xproj, const void* wh_bf16, int L_rows, const int32_t* len, void* out,
                               int ldo, int dtype) {
  KKLstmArgs a;
  memset(&a, 0, sizeof a);
  a.xproj = xproj; a.out = out; a.obs = (long long)L_rows * ldo; a.ldo = ldo; a.H = 256; a.Lmax = L_rows;
  a.len = KKLen{len, len ? 1 : 0, len ? 0 : L_rows};
  return kk_launch_lstm_h256_bf16(a, wh_bf16, B, dtype, (hipStream_t)stream);
}

// the streaming matrix-core Linear on its own (tests): x bf16 [B][rows][ldx] at item pitch xbs elements, w_bf16 [N][K] row-major bf16 (packed into the
// kernel's fragment order here: `pack_scratch` device memory of ceil(N / 16) * 16 * K bf16), bias fp32 [N] or NULL, out bf16 [B][rows][ldo] at pitch obs
namespace {
__global__ void pack_linear_kernel(const bf16_t* w, bf16_t* wl, int N, int K) {
  const long long n = (long long)((N + 15) / 16 * 16) * K;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int o = (int)(e / K), i = (int)(e - (long long)o * K);
    wl[kk_linear_pack_index(o, i, K)] = o < N ? w[(long long)o * K + i] : (bf16_t)0.f;
  }
}
}  // namespace
extern "C" int kk_op_linear_rows(void* stream, int B, const void* x_bf16, long long xbs, int ldx, int rows, const int32_t* len, const void* w_bf16, int N, int K,
                                 const float* bias, int act, void* pack_scratch, void* out_bf16, long long obs, int ldo) {
  if (!x_bf16 || !w_bf16 || !pack_scratch || !out_bf16 || B < 1 || rows < 1 || K % 32 || N % 2) return kk_fail("kk_op_linear_rows: bad argument");
  hipLaunchKernelGGL(pack_linear_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w_bf16, (bf16_t*)pack_scratch, N, K);
  KK_CHECK_LAUNCH();
  KKLinMfmaArgs f;
  memset(&f, 0, sizeof f);
  f.x = (const bf16_t*)x_bf16; f.xbs = xbs; f.ldx = ldx; f.wl = (const bf16_t*)pack_scratch; f.bias = bias; f.Nb = N;
  f.out = (bf16_t*)out_bf16; f.obs = obs; f.ldo = ldo; f.K = K; f.N = N; f.rows = rows; f.items = B;
  f.flat = (xbs == (long long)rows * ldx && obs == (long long)rows * ldo) ? 1 : 0;
  f.len = KKLen{len, len ? 1 : 0, len ? 0 : rows}; f.act = act;
  return kk_launch_linear_rows_mfma(f, (hipStream_t)stream);
}

extern "C" int kk_op_attention(void* stream, int B, const void* qkv, int ld, int T_rows, const int32_t* len, int heads, void* out, int ldo,
                               int dtype) {
  KKAttnArgs a;
  memset(&a, 0, sizeof a);
  a.qkv = qkv; a.bs = (long long)T_rows * ld; a.ld = ld; a.out = out; a.obs = (long long)T_rows * ldo; a.ldo = ldo; a.heads = heads;
  a.hs = heads * 64; a.Tmax = T_rows; a.len = KKLen{len, len ? 1 : 0, len ? 0 : T_rows}; a.scale = 0.125f;
  return kk_launch_attention(a, B, dtype, (hipStream_t)stream);
}

extern "C" int kk_op_source_stft(void* stream, int B, const float* f0, int L2_rows, const int32_t* len2, const float* lin_w9, float lin_b,
                                 int noise_mode, const float* noise, uint64_t seed, float* phase_scratch, float* har_source, void* har,
                                 int ldhar, int dtype) {
  KKSourceArgs sa;
  memset(&sa, 0, sizeof sa);
  sa.f0 = f0; sa.L2max = L2_rows; sa.len2 = len2; sa.phase = phase_scratch; sa.lin_w = lin_w9; sa.lin_b = lin_b; sa.noise = noise;
  sa.seed = seed; sa.noise_mode = noise_mode; sa.har_source = har_source; sa.Nmax = 300 * L2_rows; sa.upsample = 300;
  KK_TRY(kk_launch_source(sa, B, (hipStream_t)stream));
  if (len2) return kk_fail("kk_op_source_stft: ragged lengths are exercised through kk_forward only");
  const int Tf = 60 * L2_rows + 1;
  return kk_launch_stft20(har_source, sa.Nmax, nullptr, har, (long long)Tf * ldhar, ldhar, Tf, B, dtype, (hipStream_t)stream);
}

extern "C" int kk_op_istft_head(void* stream, int B, const void* x, int ldx, int Tf_rows, const int32_t* len_frames, float* wav, int dtype,
                                int fast) {
  return kk_launch_istft_head(x, (long long)Tf_rows * ldx, ldx, len_frames, Tf_rows, wav, (long long)5 * (Tf_rows - 1), B, dtype, fast,
                              (hipStream_t)stream);
}

extern "C" int kk_op_pack_head_w(void* stream, const void* w_bf16, void* w_frag) {
  if (!w_bf16 || !w_frag) return kk_fail("kk_op_pack_head_w: null argument");
  return kk_launch_pack_head_w((const bf16_t*)w_bf16, (bf16_t*)w_frag, (hipStream_t)stream);
}
extern "C" int kk_op_conv_post_istft(void* stream, int B, const void* x, int ldx, int Tf_rows, const int32_t* len_frames, const void* w_frag,
                                     const float* bias, float in_slope, float* wav, void* cp_out, int cp_ld) {
  if (!x || !w_frag || !bias || !wav) return kk_fail("kk_op_conv_post_istft: null argument");
  KKHeadArgs h;
  memset(&h, 0, sizeof h);
  h.x = (const bf16_t*)x; h.xbs = (long long)Tf_rows * ldx; h.ldx = ldx; h.wf = (const bf16_t*)w_frag; h.bias = bias; h.in_slope = in_slope;
  h.len_frames = len_frames; h.Tfmax = Tf_rows; h.wav = wav; h.wbs = (long long)5 * (Tf_rows - 1);
  h.cp_out = (bf16_t*)cp_out; h.cp_bs = (long long)Tf_rows * cp_ld; h.cp_ld = cp_ld;
  for (int n = 0; n < 20; ++n) h.hann_per[n] = (float)(0.5 * (1.0 - cos(2.0 * 3.14159265358979323846 * n / 20.0)));
  if (const char* e = getenv("KK_HEAD_DBG")) h.dbg = atoi(e);  // timing experiments (tools/bench_head.py)
  return kk_launch_conv_post_istft(h, B, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// debug hooks
// ------------------------------------------------------------------------------------------------
extern "C" int kk_debug_info(kk_context* cx, const char* name, int64_t* rows, int64_t* channels) {
  if (!cx || !name) return kk_fail("kk_debug_info: null argument");
  auto it = cx->dbg.find(name);
  if (it == cx->dbg.end()) return failf("kk_debug_info: no intermediate named %s in the last forward", name);
  if (rows) *rows = it->second.rows;
  if (channels) *channels = it->second.C;
  return 0;
}
extern "C" int kk_debug_fetch(kk_context* cx, void* stream, const char* name, float* dst) {
  if (!cx || !name || !dst) return kk_fail("kk_debug_fetch: null argument");
  auto it = cx->dbg.find(name);
  if (it == cx->dbg.end()) return failf("kk_debug_fetch: no intermediate named %s in the last forward", name);
  const DebugEntry& e = it->second;
  return kk_launch_convert(e.p, e.dtype, e.bs, e.ld, dst, KK_F32, (long long)e.rows * e.C, e.C, e.C, e.rows, e.B, (hipStream_t)stream);
}
extern "C" int kk_debug_override(kk_context* cx, const char* name, const float* src) {
  if (!cx || !name || !src) return kk_fail("kk_debug_override: null argument");
  cx->dbg_over[name] = src;
  return 0;
}
extern "C" void kk_debug_force_generic(kk_context* cx, int on) {
  if (!cx) return;
  cx->force_generic = (on & 1) != 0;  // bit 0: no MFMA kernel at all
  cx->no_fusion = (on & 2) != 0;      // bit 1: MFMA convs, but stand-alone statistics / AdaIN kernels
  cx->no_v4 = (on & 4) != 0;          // bit 2: the LDS-staged MFMA kernel (variant 2) instead of variant 4
  cx->no_fp8 = (on & 8) != 0;         // bit 3: quantised model, Q1 layer set on the bf16 kernel (same dequantised weights)
  cx->keep_debug = (on & 16) != 0;      // bit 4: also materialise the tensors that fused kernels skip (conv_post), for kk_debug_fetch
  cx->no_head_fusion = (on & 32) != 0;  // bit 5: stand-alone conv_post + iSTFT head kernels instead of the fused head
  cx->no_side = (on & 256) != 0;        // bit 8: no side stream (every launch of a forward on the caller's stream)
  cx->linrows_mode = (on & 512) ? 2 : (on & 1024) ? 1 : 0;  // bit 9: Linear layers never on the streaming kernel; bit 10: always (default: by the rows in flight)
  cx->v5_mode = (on & 64) ? 1 : (on & 128) ? 2 : 0;  // bit 6: conv variant 5 (wave-specialised persistent) wherever eligible; bit 7: never (default: >= 9 taps)
}

// load_model's quantization branch (mlx_audio/tts/utils.py:241-260): the checkpoint's Linear / Embedding weights went through MLX's
// affine `bits`-bit group quantisation.  The caller hands over the DEQUANTISED weights (quant.py); with compute_dtype bf16 the
// layer set with eligible shapes (Albert's five linears, bert_encoder) is re-quantised in kk_finalize to e4m3 with one
// power-of-two scale per group and runs on v_mfma_scale_f32_32x32x64_f8f6f4.
extern "C" int kk_set_quantization(kk_model* m, int group_size, int bits) {
  if (!m) return kk_fail("kk_set_quantization: null model");
  if (m->finalized) return kk_fail("kk_set_quantization: call before kk_finalize");
  if (bits != 8) return kk_fail("kk_set_quantization: only the 8-bit path exists (bits must be 8)");
  if (group_size < 32 || group_size % 32) return kk_fail("kk_set_quantization: group_size must be a multiple of 32");
  if (m->adt != KK_BF16) return kk_fail("kk_set_quantization: the fp8 path needs compute_dtype bf16");
  m->q_group = group_size;
  return 0;
}
extern "C" int kk_quantized_layers(const kk_model* m) {
  if (!m || !m->finalized) return -1;
  const ConvW* set[] = {&m->map_in, &m->qkv, &m->att_dense, &m->ffn, &m->ffn_out, &m->bert_encoder};
  int n = 0;
  for (const ConvW* c : set) n += c->fp8 ? 1 : 0;
  return n;
}

// ---- MX-fp8 single-op entry points (tests)
extern "C" int kk_mxfp8_bytes(int rows, int K, size_t* q_bytes, size_t* s_bytes) {
  if (rows <= 0 || K <= 0 || K % 64 || !q_bytes || !s_bytes) return kk_fail("kk_mxfp8_bytes: bad argument");
  *q_bytes = kk_mxfp8_q_bytes(rows, K);
  *s_bytes = kk_mxfp8_s_bytes(rows, K);
  return 0;
}
extern "C" int kk_mxfp8_pack_weight(const float* w_host, int N, int K, int group, uint8_t* q_host, uint8_t* s_host) {
  if (!w_host || !q_host || !s_host) return kk_fail("kk_mxfp8_pack_weight: null argument");
  return kk_mxfp8_pack_weight_host(w_host, N, K, group, q_host, s_host);
}
extern "C" int kk_op_linear_mxfp8(void* stream, const void* x_bf16, int ldx, int M, int rows_per_item, const int32_t* len, int K,
                                  const void* wq, const void* ws, int N, const float* bias, int act, void* aq, void* as, void* out_bf16,
                                  int ldo) {
  if (!x_bf16 || !wq || !ws || !aq || !as || !out_bf16) return kk_fail("kk_op_linear_mxfp8: null argument");
  KK_TRY(kk_launch_mxfp8_quant_rows(x_bf16, ldx, M, K, aq, as, (hipStream_t)stream));
  KKFp8Args f;
  memset(&f, 0, sizeof f);
  f.aq = (const uint4*)aq; f.as = (const unsigned char*)as; f.wq = (const uint4*)wq; f.ws = (const unsigned char*)ws;
  f.M = M; f.N = N; f.K = K; f.bias = bias; f.out = (bf16_t*)out_bf16; f.ldo = ldo; f.rows_per_item = rows_per_item;
  f.lout = KKLen{len, len ? 1 : 0, len ? 0 : rows_per_item};
  f.act = act;
  return kk_launch_linear_mxfp8(f, (hipStream_t)stream);
}
extern "C" void kk_debug_set_op_wfrag(const void* w_frag) { g_op_wfrag = w_frag; }
extern "C" void kk_debug_set_op_variant(int v) { g_op_variant = v == 5 ? 5 : 4; }
extern "C" int kk_op_pack_w_frag(void* stream, const void* w_bf16, void* w_frag, int Kw, int CoutP, int CinP) {
  return kk_launch_pack_w_frag(w_bf16, w_frag, Kw, CoutP, CinP, (hipStream_t)stream);
}
extern "C" void kk_debug_clear(kk_context* cx) {
  if (!cx) return;
  cx->dbg_over.clear();
  cx->dbg.clear();
}

// ------------------------------------------------------------------------------------------------
// per-kernel-class timing (bench.py / profiling only)
// ------------------------------------------------------------------------------------------------
extern "C" int kk_profile_begin(kk_context* cx, int max_launches) {
  if (!cx || max_launches <= 0) return kk_fail("kk_profile_begin: bad argument");
  while ((int)cx->prof_ev.size() < 2 * max_launches) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return kk_fail("kk_profile_begin: hipEventCreate failed");
    cx->prof_ev.push_back(e);
  }
  cx->prof_rec.clear();
  cx->prof_on = true;
  return 0;
}
// Sums the recorded launches per class: ms[cls], flops[cls], bytes[cls], count[cls] for cls < ncls.  Synchronises the
// events it reads.  Classes: 0 conv_generic 1 conv_mfma 2 instnorm_stats 3 adain_act 4 lstm 5 istft_head 6 layernorm
// 7 attention 8 source 9 stft
extern "C" int kk_profile_end(kk_context* cx, int ncls, double* ms, double* flops, double* bytes, int64_t* count) {
  if (!cx || !ms || !flops || !bytes || !count) return kk_fail("kk_profile_end: null argument");
  for (int i = 0; i < ncls; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; count[i] = 0; }
  cx->prof_on = false;
  for (size_t i = 0; i < cx->prof_rec.size(); ++i) {
    if (hipEventSynchronize(cx->prof_ev[2 * i + 1]) != hipSuccess) return kk_fail("kk_profile_end: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, cx->prof_ev[2 * i], cx->prof_ev[2 * i + 1]) != hipSuccess) return kk_fail("kk_profile_end: elapsed failed");
    const auto& r = cx->prof_rec[i];
    if (r.cls >= 0 && r.cls < ncls) { ms[r.cls] += t; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes; count[r.cls] += 1; }
  }
  cx->prof_rec.clear();
  return 0;
}
