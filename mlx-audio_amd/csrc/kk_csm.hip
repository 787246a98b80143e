// CSM-1B frame generator: SesameModel.generate_frame, mlx_audio/tts/models/sesame/sesame.py:349-395
//   tokens [B][S][n_cb+1] -> masked sum of 33 embeddings -> Llama backbone (KV cache) -> codebook0 head -> sample ->
//   31 x { projection -> Llama depth decoder (fresh cache per frame) -> audio_head[i-1] -> sample -> embed } -> codes [B][n_cb]
// Llama layer (mlx_lm LlamaModel with the reference's Attention, sesame.py:296-299; attention.py:113-195):
//   h += Wo . attn(rope(Wq x), rope(Wk x), Wv x), x = rms(h);   h += Wdown (silu(Wgate x) * Wup x), x = rms(h);   final rms
//   RoPE: interleaved pairs, llama3-scaled frequencies (attention.py:33-110); GQA: query head h reads kv head h / (H / KV);
//   mask: causal inside the new block, every cached key visible (sesame.py:37-48).
// Round 1: fp32, the library's generic conv kernel for every linear (M = B*S rows is tiny in the decode loop), one simple attention
// kernel for both head sizes.  The KV caches are library-owned device memory (kk_csm_setup_caches), like the module-owned caches of
// the reference (sesame.py:320-333).
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/kokoro_hip.h"
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

#include "kk_csm_gemvm.h"

struct Lin {  // generic-kernel pack [1][Cin][ldw]
  size_t off = 0;
  int Cin = 0, Cout = 0, ldw = 0;
  const float* w = nullptr;
  // bf16 weight mode (kk_csm_set_weight_dtype): the same matrix as bf16 in COLUMN-BLOCK order for the fused GEMV of the single-token
  // steps: [Cout / CB][Cin][CB] with CB = 8 * oct columns per workgroup, so a workgroup streams one contiguous region
  size_t boff = 0;
  const uint16_t* wb = nullptr;
  int oct = 0;  // 0 = no block pack (K not a multiple of 64)
  // the same bf16 matrix in FRAGMENT order for the matrix-core GEMV (gemvm_kernel): [Cout / (16 nsub)][Cin / 32][nsub][64 lanes][8]: a wave
  // load of 1 KiB is one 32 x 16 B operand of v_mfma_f32_16x16x32_bf16 (lane L: k = 32 c + 8 (L / 16) + j, n = 16 s + L % 16)
  size_t moff = 0;
  const uint16_t* wm = nullptr;
  int nsub = 0;  // 0 = no fragment pack (K not a multiple of 32)
  int ks = 1;    // split-K slices of a deep projection (K >= 4096): partial tiles + combine
  int cached = 0;  // 1: plain (cacheable) weight loads instead of nontemporal ones
};
struct Vec {
  size_t off = 0;
  size_t n = 0;
  const float* p = nullptr;
};
struct LlamaLayer {
  Lin qkv, o, gu, down;
  Vec n1, n2;
};
struct Stack {
  kk_llama_args a;
  std::vector<LlamaLayer> layers;
  Vec norm, rope;  // rope: [max_pos][hd/2][2] cos, sin
  float* kc = nullptr;  // [layers][maxB][max_pos][KV*hd]
  float* vc = nullptr;
  int max_pos = 0, offset = 0;
  int* pos_dev = nullptr;  // backbone only: device copy of `offset` (null: positions are launch constants)
  int* pad_dev = nullptr;  // backbone only: [max_batch] left padding of each item's prompt (kk_csm_set_padding), zeros by default
};

}  // namespace

struct kk_csm {
  kk_csm_config cfg;
  std::map<std::string, std::vector<float>> host;
  std::vector<float> pack;
  float* dev = nullptr;
  bool finalized = false;
  int wdt = KK_F32;              // weight storage of the single-token steps (KK_F32 / KK_BF16)
  std::vector<uint16_t> packb;   // host staging of the bf16 copies
  uint16_t* devb = nullptr;
  Stack bb, dec;
  Vec text_emb, audio_emb;
  Lin proj, c0_head;
  std::vector<Lin> audio_head;
  int max_batch = 0;
  float* dbg_logits = nullptr;  // [n_cb][maxB][V] of the last frame
  float* proj_table = nullptr;  // bf16 weight mode: projection(audio_embeddings) [n_cb * V][decoder hidden], computed once at finalize (weights: shared by kk_csm_share)
  // graph replay of the single-token frame step (kk_csm_set_graph_mode)
  struct GraphEntry {
    std::vector<unsigned long long> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int seen = 0;
  };
  bool graph_mode = false;
  std::vector<GraphEntry> graphs;
  hipStream_t cap_stream = nullptr;
  // kk_csm_reset_caches / kk_csm_set_padding take no stream: what they change on the device is applied on the NEXT frame's stream (a
  // synchronous hipMemset / hipMemcpy here would touch the legacy stream and break another thread's graph capture)
  bool reset_pending = false, pad_pending = false;
  std::vector<int32_t> pad_host;
  const kk_csm* weights_of = nullptr;  // kk_csm_share: `dev` / `devb` belong to that generator (immutable after finalize), not to this one
};

namespace {

int rup(int v, int m) { return (v + m - 1) / m * m; }

// kk_csm_debug_skip (TIMING ONLY, wrong results): kernel classes of the single-token step that are not launched -- bit 0 q|k|v, 1 attention,
// 2 o, 3 gate|up, 4 down, 5 split-K combine, 6 heads, 7 sampler, 8 projection: the in-situ cost of a class is the frame time it removes
int g_skip = 0;
// A/B switches consulted on the launch path (environment, read once): bit 0 KK_CSM_PROMPT_F32, 1 KK_CSM_OLD_ATTN, 2 KK_CSM_NO_PROJ_TABLE
int ab_switches() {
  static int v = -1;
  if (v < 0) v = (getenv("KK_CSM_PROMPT_F32") ? 1 : 0) | (getenv("KK_CSM_OLD_ATTN") ? 2 : 0) | (getenv("KK_CSM_NO_PROJ_TABLE") ? 4 : 0);
  return v;
}
// kk_csm_debug_timestamps: in-kernel wall-clock marks (100 MHz) of the instrumented kernels, 8 words per launch in launch order:
// [class id, earliest workgroup start, latest workgroup end, workgroup 0 after its input loads, workgroup 0's start / shader clock at start / end / shader
// clock at end]; null in production
unsigned long long* g_ts = nullptr;
int g_ts_cap = 0, g_ts_next = 0;
unsigned long long* ts_slot() { return (g_ts && g_ts_next < g_ts_cap) ? g_ts + 8 * (size_t)g_ts_next++ : nullptr; }

// ------------------------------------------------------------------------------------------------------------- kernels
// h[b][s][:] = sum_j mask[b][s][j] * emb_j(tokens[b][s][j]),  j < n_cb: audio_embeddings[token + j*V], j = n_cb: text_embeddings
__global__ __launch_bounds__(256) void embed_sum_kernel(const int* tokens, const float* mask, const float* audio, const float* text, int ncb, int V, int TV,
                                                        int D, float* h) {
  // grid (row = b * S + s, column block of 256).  The row's ids and masks go to LDS first so that the 33 embedding loads of a column do not wait on 33
  // dependent id loads (the first form walked them one L2 round trip at a time: 60-130 us for the 8 rows of a single-token frame); the loads of 16
  // code books are in flight together, the sum stays in code-book order.
  __shared__ int tk_s[72];
  __shared__ float mk_s[72];
  const long long row = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if ((int)threadIdx.x <= ncb) {
    const int j = threadIdx.x;
    tk_s[j] = j < ncb ? clamp_id(tokens[row * (ncb + 1) + j], V) + j * V : clamp_id(tokens[row * (ncb + 1) + j], TV);
    mk_s[j] = mask[row * (ncb + 1) + j];
  }
  __syncthreads();
  if (c >= D) return;
  float acc = 0.f;
  for (int j0 = 0; j0 <= ncb; j0 += 16) {
    float v[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = j0 + jj <= ncb ? j0 + jj : ncb;
      v[jj] = (j < ncb ? audio : text)[(long long)tk_s[j] * D + c];
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj)
      if (j0 + jj <= ncb) acc += v[jj] * mk_s[j0 + jj];
  }
  h[row * D + c] = acc;
}

// rows of audio_embeddings for code book `cb`: out[b][pos][:] = audio[(codes[b] + cb*V)][:]   (out row pitch = rows*D per item)
__global__ __launch_bounds__(256) void embed_audio_kernel(const int* codes, int cstride, const float* audio, int cb, int V, int D, float* out, int rows,
                                                          int pos) {
  const int b = blockIdx.x;
  const float* e = audio + ((long long)clamp_id(codes[(long long)b * cstride], V) + (long long)cb * V) * D;
  for (int c = threadIdx.x; c < D; c += 256) out[((long long)b * rows + pos) * D + c] = e[c];
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* src, long long sbs, float* dst, long long dbs, int D) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) dst[(long long)b * dbs + c] = src[(long long)b * sbs + c];
}

// RMSNorm over the last axis, fp32: x * rsqrt(mean(x^2) + eps) * w
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, const float* w, int D, float eps, float* out) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* xr = x + row * D;
  float ss = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) ss = __builtin_fmaf(xr[c], xr[c], ss);
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = (red[0] + red[1]) + (red[2] + red[3]);
  const float r = 1.0f / sqrtf(tot / (float)D + eps);
  for (int c = threadIdx.x; c < D; c += 256) out[row * D + c] = xr[c] * r * w[c];
}

// RoPE on q (in place) and k, then k / v of the new rows go into the cache at [offset + s]
// `pad` (nullable, [B]): ragged prompts are LEFT-padded to a common length; item b's first pad[b] cache slots hold nothing, its token in
// slot p sits at position p - pad[b] (RoPE) and attends to slots >= pad[b] only.  Padding rows are rotated with position 0 (never read).
__global__ __launch_bounds__(256) void rope_append_kernel(float* qkv, int S, int H, int KV, int hd, const float* rope, const int* pos_dev, int offset,
                                                          float* kc, float* vc, int max_pos, const int* pad) {
  const int s = blockIdx.x, b = blockIdx.y;
  if (pos_dev) offset += *pos_dev;  // the backbone's position lives in device memory so that a captured frame step can be replayed
  const int W = (H + 2 * KV) * hd, half = hd / 2;
  float* row = qkv + ((long long)b * S + s) * W;
  const int pd = pad ? pad[b] : 0;
  const int rpos = offset + s - pd > 0 ? offset + s - pd : 0;
  const float* cs = rope + (long long)rpos * half * 2;
  float* kdst = kc + ((long long)b * max_pos + offset + s) * KV * hd;
  float* vdst = vc + ((long long)b * max_pos + offset + s) * KV * hd;
  for (int e = threadIdx.x; e < (H + KV) * half; e += 256) {
    const int hh = e / half, i = e - hh * half;
    float* p = row + hh * hd + 2 * i;  // q heads first, k heads right behind them
    const float c = cs[2 * i], sn = cs[2 * i + 1];
    const float x0 = p[0], x1 = p[1];
    const float y0 = x0 * c - x1 * sn, y1 = x1 * c + x0 * sn;
    if (hh < H) {
      p[0] = y0; p[1] = y1;
    } else {
      kdst[(hh - H) * hd + 2 * i] = y0;
      kdst[(hh - H) * hd + 2 * i + 1] = y1;
    }
  }
  const float* vsrc = row + (H + KV) * hd;
  for (int e = threadIdx.x; e < KV * hd; e += 256) vdst[e] = vsrc[e];
}

// one workgroup per (query s, head h, item b): scores over the cached keys 0 .. offset+s, softmax, weighted sum of V.
// Single-token steps are latency-bound, so the dependent chains are kept short: a thread takes a whole key row as independent 16-byte
// loads (q sits in LDS), and the P.V product splits the keys over G = 512 / hd groups of threads (each thread one float4 of the head
// dimension), whose partial sums are added in group order through LDS.
// `causal`: 1 = query s sees keys 0 .. offset+s (index_causal_mask, sesame.py:41-48); 0 = every query of the block sees all offset+S
// keys (Mimi's streaming transformer passes no mask, transformer.py:79-104).  `ctx` >= 0: only the last ctx CACHED keys (+ the block).
// FUSE (single-token steps: S = 1, causal, no context limit): the RoPE of q and of the new key and the append of the new key / value row
// happen HERE instead of in rope_append_kernel (one launch less per layer and step): q and the new k are rotated while they are staged in
// LDS -- the same expressions, so the same bits as the two-kernel path --, the new row is the last key / value of the walk, and the first
// query head of each kv group writes it to the cache for the steps to come (the other heads of the group never read that row here).
template <bool FUSE>
__global__ __launch_bounds__(128) void attn_cache_kernel(const float* qkv, int S, int H, int KV, int hd, const int* pos_dev, int offset, float* kc,
                                                         float* vc, int max_pos, float scale, float* out, int causal, int ctx, const float* rope,
                                                         const int* pad) {
  extern __shared__ __attribute__((aligned(16))) float sc[];  // [max_pos] scores, [hd] q, [G][hd] partial outputs, [hd] new k, [hd] new v
  __shared__ float red[2];
  if (pos_dev) offset += *pos_dev;
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int pd = pad ? pad[b] : 0;  // left padding of a ragged prompt: the keys start at slot pd, key j of the walk is slot klo + j as before
  const int klo = ctx >= 0 && offset > ctx ? offset - ctx : pd;
  const int W = (H + 2 * KV) * hd, kvh = h / (H / KV), nk = (causal ? offset + s + 1 : offset + S) - klo;
  if (nk <= 0) {  // a padding row: no key, the output is defined as zero (nothing reads it)
    for (int e = tid; e < hd; e += 128) out[((long long)b * S + s) * H * hd + h * hd + e] = 0.f;
    return;
  }
  const int mp4 = (max_pos + 3) & ~3;
  float* qs = sc + mp4;        // [hd]
  float* po = qs + hd;         // [G][hd]
  float* kn = po + (512 / hd) * hd;  // [hd] (FUSE)
  float* vn = kn + hd;               // [hd] (FUSE)
  const float* q = qkv + ((long long)b * S + s) * W + h * hd;
  const float* kb = kc + ((long long)b * max_pos + klo) * KV * hd + kvh * hd;
  const float* vb = vc + ((long long)b * max_pos + klo) * KV * hd + kvh * hd;
  const int jn = FUSE ? nk - 1 : -1;  // the key that is not in the cache yet
  if (FUSE) {
    const float* cs = rope + (long long)(offset + s - pd) * (hd / 2) * 2;
    const float* kq = qkv + ((long long)b * S + s) * W + (H + kvh) * hd;
    const float* vq = qkv + ((long long)b * S + s) * W + (H + KV + kvh) * hd;
    for (int i = tid; i < hd / 2; i += 128) {
      const float c = cs[2 * i], sn = cs[2 * i + 1];
      const float x0 = q[2 * i], x1 = q[2 * i + 1];
      qs[2 * i] = x0 * c - x1 * sn;
      qs[2 * i + 1] = x1 * c + x0 * sn;
      const float k0 = kq[2 * i], k1 = kq[2 * i + 1];
      kn[2 * i] = k0 * c - k1 * sn;
      kn[2 * i + 1] = k1 * c + k0 * sn;
    }
    for (int e = tid; e < hd; e += 128) vn[e] = vq[e];
  } else {
    for (int e = tid; e < hd; e += 128) qs[e] = q[e];
  }
  __syncthreads();
  if (FUSE && h % (H / KV) == 0) {
    float* kdst = kc + ((long long)b * max_pos + offset + s) * KV * hd + kvh * hd;
    float* vdst = vc + ((long long)b * max_pos + offset + s) * KV * hd + kvh * hd;
    for (int e = tid; e < hd; e += 128) { kdst[e] = kn[e]; vdst[e] = vn[e]; }
  }
  const int hd4 = hd >> 2;
  float mx = -INFINITY;
  for (int j = tid; j < nk; j += 128) {
    const float4* kr = j == jn ? (const float4*)kn : (const float4*)(kb + (long long)j * KV * hd);
    float d = 0.f;
    for (int e0 = 0; e0 < hd4; e0 += 16) {  // hd = 64 or 128: 16 independent loads per trip
      float4 kv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) kv[t] = kr[e0 + t];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float4 qv = *(const float4*)(qs + 4 * (e0 + t));
        d = __builtin_fmaf(qv.x, kv[t].x, d);
        d = __builtin_fmaf(qv.y, kv[t].y, d);
        d = __builtin_fmaf(qv.z, kv[t].z, d);
        d = __builtin_fmaf(qv.w, kv[t].w, d);
      }
    }
    d *= scale;
    sc[j] = d;
    mx = fmaxf(mx, d);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(red[0], red[1]);
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < nk; j += 128) {
    const float p = expf(sc[j] - mx);
    sc[j] = p;
    sum += p;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1]);
  const int G = 128 / hd4, e4 = tid % hd4, g = tid / hd4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = g; j < nk; j += G) {
    const float p = sc[j];
    const float4 v = j == jn ? *(const float4*)(vn + 4 * e4) : *(const float4*)(vb + (long long)j * KV * hd + 4 * e4);
    acc.x = __builtin_fmaf(p, v.x, acc.x);
    acc.y = __builtin_fmaf(p, v.y, acc.y);
    acc.z = __builtin_fmaf(p, v.z, acc.z);
    acc.w = __builtin_fmaf(p, v.w, acc.w);
  }
  *(float4*)(po + g * hd + 4 * e4) = acc;
  __syncthreads();
  for (int e = tid; e < hd; e += 128) {
    float a = 0.f;
    for (int gg = 0; gg < G; ++gg) a += po[gg * hd + e];  // group order
    out[((long long)b * S + s) * H * hd + h * hd + e] = a * inv;
  }
}

// attn_step_kernel (round 3): the single-token attention over a SHORT cache (max_pos <= 64: the depth decoder's 33 positions, small test
// stacks) as ONE memory round trip.  attn_cache_kernel walks the keys in dependent steps (a thread per key with 2 x 16 loads, then the
// values 4 keys at a time): 7.6 us per launch in the frame (tools/csm_skip_sweep.py), 124 launches.  Here a workgroup owns one (item, kv
// head) and its G = H / KV query heads: every cached K and V row, the new q / k / v and the RoPE row are requested at once, land in LDS
// (K rows padded to hd + 4 floats: a lane per key reads 16-byte pieces without bank conflicts), and scores, softmax (a lane per key, one
// wave per head) and the weighted sum run out of LDS.  RoPE of q / k and the cache append are fused as in attn_cache_kernel<true>.
template <int HD>
__global__ __launch_bounds__(256) void attn_step_kernel(const float* qkv, int H, int KV, const int* pos_dev, int offset, float* kc, float* vc, int max_pos,
                                                         float scale, float* out, const float* rope, const int* pad, unsigned long long* ts) {
  constexpr int HD4 = HD / 4, KP = HD + 4;
  extern __shared__ __attribute__((aligned(16))) float sma[];
  ts_begin(ts, 1);
  const int G = H / KV, kvh = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* ks = sma;                   // [max_pos][KP]
  float* vs = ks + max_pos * KP;     // [max_pos rounded up to 4][HD]
  float* qs = vs + ((max_pos + 3) & ~3) * HD;  // [G][HD]
  float* sc = qs + G * HD;           // [G][64]
  float* inv = sc + G * 64;          // [G]
  if (pos_dev) offset += *pos_dev;
  const int pd = pad ? pad[b] : 0, nk = offset + 1 - pd;
  const int W = (H + 2 * KV) * HD;
  if (nk <= 0) {  // a padding row: no key, the output is defined as zero (nothing reads it)
    for (int o = tid; o < G * HD; o += 256) out[(long long)b * H * HD + (long long)kvh * G * HD + o] = 0.f;
    return;
  }
  const float* kb = kc + ((long long)b * max_pos + pd) * KV * HD + kvh * HD;
  const float* vb = vc + ((long long)b * max_pos + pd) * KV * HD + kvh * HD;
  const int nold = (nk - 1) * HD4;  // float4 of the cached rows (<= 63 * 32)
  float4 kr[8], vr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i;
    if (idx < nold) {  // (kept under the branch: at the depth decoder's 2-32 keys most of the 8 rounds are skipped by whole waves; clamped unconditional loads measured 3 % slower per frame)
      const int j = idx / HD4, e = idx - j * HD4;
      kr[i] = *(const float4*)(kb + (long long)j * KV * HD + 4 * e);
      vr[i] = *(const float4*)(vb + (long long)j * KV * HD + 4 * e);
    }
  }
  {  // the new position: RoPE on q (G heads) and k, v as is; k / v also go to the cache
    const float* cs = rope + (long long)(offset - pd) * (HD / 2) * 2;
    const float* q = qkv + (long long)b * W + (long long)kvh * G * HD;
    const float* kq = qkv + (long long)b * W + (H + kvh) * HD;
    const float* vq = qkv + (long long)b * W + (H + KV + kvh) * HD;
    float* kdst = kc + ((long long)b * max_pos + offset) * KV * HD + kvh * HD;
    float* vdst = vc + ((long long)b * max_pos + offset) * KV * HD + kvh * HD;
    for (int i = tid; i < G * (HD / 2); i += 256) {
      const int ii = i % (HD / 2);
      const float2 c = *(const float2*)(cs + 2 * ii), x = *(const float2*)(q + 2 * i);
      *(float2*)(qs + 2 * i) = make_float2(x.x * c.x - x.y * c.y, x.y * c.x + x.x * c.y);
    }
    if (tid < HD / 2) {
      const float2 c = *(const float2*)(cs + 2 * tid), x = *(const float2*)(kq + 2 * tid);
      const float2 kn = make_float2(x.x * c.x - x.y * c.y, x.y * c.x + x.x * c.y);
      *(float2*)(ks + (nk - 1) * KP + 2 * tid) = kn;
      *(float2*)(kdst + 2 * tid) = kn;
    } else if (tid >= 128 && tid < 128 + HD / 2) {
      const int t = tid - 128;
      const float2 v = *(const float2*)(vq + 2 * t);
      *(float2*)(vs + (nk - 1) * HD + 2 * t) = v;
      *(float2*)(vdst + 2 * t) = v;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i;
    if (idx < nold) {
      const int j = idx / HD4, e = idx - j * HD4;
      *(float4*)(ks + j * KP + 4 * e) = kr[i];
      *(float4*)(vs + j * HD + 4 * e) = vr[i];
    }
  }
  for (int i = tid; i < (((nk + 3) & ~3) - nk) * HD; i += 256) vs[nk * HD + i] = 0.f;  // V rows of the padded key count (weight exactly 0)
  __syncthreads();
  ts_mid(ts);
  for (int p = tid >> 2; p < G * nk; p += 64) {  // (head, key) per 4 lanes: each lane a quarter of the head dimension, then two butterfly adds
    const int g = p / nk, j = p - g * nk, qd = tid & 3;
    const float4* kr4 = (const float4*)(ks + j * KP) + qd * (HD4 / 4);
    const float4* q4 = (const float4*)(qs + g * HD) + qd * (HD4 / 4);
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int e = 0; e < HD4 / 4; e += 2) {
      const float4 ka = kr4[e], qa = q4[e], kb2 = kr4[e + 1], qb = q4[e + 1];
      d0 = __builtin_fmaf(qa.x, ka.x, d0); d0 = __builtin_fmaf(qa.y, ka.y, d0); d0 = __builtin_fmaf(qa.z, ka.z, d0); d0 = __builtin_fmaf(qa.w, ka.w, d0);
      d1 = __builtin_fmaf(qb.x, kb2.x, d1); d1 = __builtin_fmaf(qb.y, kb2.y, d1); d1 = __builtin_fmaf(qb.z, kb2.z, d1); d1 = __builtin_fmaf(qb.w, kb2.w, d1);
    }
    float d = d0 + d1;
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    if (qd == 0) sc[g * 64 + j] = d * scale;
  }
  __syncthreads();
  for (int g = wave; g < G; g += 4) {  // softmax of one head: a lane per key (nk <= 64)
    const float v = lane < nk ? sc[g * 64 + lane] : -INFINITY;
    float mx = v;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float pr = lane < nk ? expf(v - mx) : 0.f;
    float sum = pr;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    sc[g * 64 + lane] = pr;  // (lanes >= nk: exact zeros, so the weighted sum below may run over a padded key count)
    if (lane == 0) inv[g] = 1.0f / sum;
  }
  __syncthreads();
  const int nk4 = (nk + 3) & ~3;  // <= 64; the V rows behind nk are finite (stale or zero-filled below) and meet weight 0
  for (int o = tid; o < G * HD; o += 256) {
    const int g = o / HD, e = o - g * HD;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = 0; j < nk4; j += 4) {  // (four chains; key order inside each)
      const float4 pw = *(const float4*)(sc + g * 64 + j);
      a0 = __builtin_fmaf(pw.x, vs[j * HD + e], a0);
      a1 = __builtin_fmaf(pw.y, vs[(j + 1) * HD + e], a1);
      a2 = __builtin_fmaf(pw.z, vs[(j + 2) * HD + e], a2);
      a3 = __builtin_fmaf(pw.w, vs[(j + 3) * HD + e], a3);
    }
    out[(long long)b * H * HD + (long long)kvh * G * HD + o] = ((a0 + a1) + (a2 + a3)) * inv[g];
  }
  ts_end(ts);
}
static size_t attn_step_lds_bytes(int max_pos, int hd, int G) {
  return ((size_t)max_pos * (hd + 4) + (size_t)((max_pos + 3) & ~3) * hd + (size_t)G * hd + (size_t)G * 64 + 16) * 4;
}

// attn_decode_kernel (round 3): the single-token attention over a LONG cache (the backbone: up to 2048 positions) in chunks of 8192 / hd keys
// with an online softmax.  attn_cache_kernel<true> walks the keys in dependent steps (a thread per key, then the values 8 keys at a
// time, one L2 round trip per step): ~9 us at 70 keys, ~20 us at 315 (config 4's prompts), x 16 layers per frame.  Here, as in
// attn_step_kernel, a workgroup owns one (item, kv head) and its G query heads; a chunk's K and V rows are requested at once (8 + 8
// 16-byte loads per thread) one chunk AHEAD of the arithmetic, land in LDS, and scores (4 lanes per (head, key)), the running maximum /
// sum (a wave per head) and the weighted sum (a thread per output, rescaled per chunk) run out of LDS.  RoPE + cache append fused.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* qkv, int H, int KV, const int* pos_dev, int offset, float* kc, float* vc, int max_pos,
                                                           float scale, float* out, const float* rope, const int* pad, int nsplit, float* part) {
  constexpr int HD4 = HD / 4, KP = HD + 4, CH = 8192 / HD;
  extern __shared__ __attribute__((aligned(16))) float smd[];
  const int G = H / KV, kvh = blockIdx.x, b = blockIdx.y, z = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* ks = smd;                 // [CH][KP]
  float* vs = ks + CH * KP;        // [CH][HD]
  float* qs = vs + CH * HD;        // [G][HD]
  float* sc = qs + G * HD;         // [G][CH]
  float* kn = sc + G * CH;         // [HD] the new key (RoPE applied)
  float* vn = kn + HD;             // [HD]
  float* mrun = vn + HD;           // [G] running maximum
  float* lrun = mrun + 8;          // [G] running sum
  float* alpha = lrun + 8;         // [G] rescale factor of the current chunk
  if (pos_dev) offset += *pos_dev;
  const int pd = pad ? pad[b] : 0, nk = offset + 1 - pd;
  const int W = (H + 2 * KV) * HD;
  // key split (flash decoding): workgroup z of nsplit takes the chunks z, z + nsplit, ...; with nsplit > 1 it leaves an UNNORMALISED partial result
  // (sum, running maximum, running sum per head) in `part` and attn_merge_kernel combines the splits
  float* pz = part + (((long long)b * KV + kvh) * nsplit + z) * (long long)G * (HD + 2);
  if (nk <= 0 || z * CH >= nk) {  // a padding row (no key: the output is defined as zero) or a split without keys
    if (nsplit == 1) {
      for (int o = tid; o < G * HD; o += 256) out[(long long)b * H * HD + (long long)kvh * G * HD + o] = 0.f;
    } else {
      for (int o = tid; o < G * (HD + 2); o += 256) pz[o] = (o % (HD + 2)) == HD ? -INFINITY : 0.f;
    }
    return;
  }
  const float* kb = kc + ((long long)b * max_pos + pd) * KV * HD + kvh * HD;
  const float* vb = vc + ((long long)b * max_pos + pd) * KV * HD + kvh * HD;
  const int nold = nk - 1;  // cached keys; key nk - 1 is the new one (kn / vn)
  float4 kr[8], vr[8];
  // rows j0 .. j0 + CH - 1 of the cache (clamped: rows past nold are read and never used; every load unconditional)
#define KK_LOAD_CHUNK(J0)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                     \
    const int idx = tid + 256 * i, j = (J0) + idx / HD4, e = idx % HD4, jc = j < nold ? j : (nold > 0 ? nold - 1 : 0); \
    kr[i] = *(const float4*)(kb + (long long)jc * KV * HD + 4 * e);                                                   \
    vr[i] = *(const float4*)(vb + (long long)jc * KV * HD + 4 * e);                                                   \
  }
  KK_LOAD_CHUNK(z * CH)
  {  // the new position: RoPE on q (G heads) and k, v as is; k / v also go to the cache
    const float* cs = rope + (long long)(offset - pd) * (HD / 2) * 2;
    const float* q = qkv + (long long)b * W + (long long)kvh * G * HD;
    const float* kq = qkv + (long long)b * W + (H + kvh) * HD;
    const float* vq = qkv + (long long)b * W + (H + KV + kvh) * HD;
    float* kdst = kc + ((long long)b * max_pos + offset) * KV * HD + kvh * HD;
    float* vdst = vc + ((long long)b * max_pos + offset) * KV * HD + kvh * HD;
    const bool appender = (nold / CH) % nsplit == z;  // the split that owns the new key's chunk also appends it to the cache
    for (int i = tid; i < G * (HD / 2); i += 256) {
      const int ii = i % (HD / 2);
      const float2 c = *(const float2*)(cs + 2 * ii), x = *(const float2*)(q + 2 * i);
      *(float2*)(qs + 2 * i) = make_float2(x.x * c.x - x.y * c.y, x.y * c.x + x.x * c.y);
    }
    if (tid < HD / 2) {
      const float2 c = *(const float2*)(cs + 2 * tid), x = *(const float2*)(kq + 2 * tid);
      const float2 k2 = make_float2(x.x * c.x - x.y * c.y, x.y * c.x + x.x * c.y);
      *(float2*)(kn + 2 * tid) = k2;
      if (appender) *(float2*)(kdst + 2 * tid) = k2;
    } else if (tid >= 128 && tid < 128 + HD / 2) {
      const int t = tid - 128;
      const float2 v = *(const float2*)(vq + 2 * t);
      *(float2*)(vn + 2 * t) = v;
      if (appender) *(float2*)(vdst + 2 * t) = v;
    }
    if (tid < 8) { mrun[tid] = -INFINITY; lrun[tid] = 0.f; }
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // outputs o = tid + 256 i of the G x HD (<= 1024)
  for (int j0 = z * CH; j0 < nk; j0 += nsplit * CH) {
    const int cn = nk - j0 < CH ? nk - j0 : CH, cn4 = (cn + 3) & ~3;
    // this chunk's rows into LDS (the loads went out a chunk ago); the new key / value take their slot if it falls into the chunk
    // (rows past the cached keys get zeros: the new key's slot is filled below -- kn / vn are visible only behind the barrier -- and the rows that pad
    // the key count to a multiple of 4 meet weight exactly 0; unconditional stores, so the loop unrolls and the rows stay in registers)
#pragma clang loop unroll(full)
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, j = idx / HD4, e = idx % HD4;
      const bool live = j0 + j < nold;
      *(float4*)(ks + j * KP + 4 * e) = live ? kr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      *(float4*)(vs + j * HD + 4 * e) = live ? vr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (nold >= j0 && nold < j0 + CH) {  // (uniform) the new position lies in this chunk: row nold - j0
      const int jn = nold - j0;
      for (int e = tid; e < HD; e += 256) { ks[jn * KP + e] = kn[e]; vs[jn * HD + e] = vn[e]; }
    }
    KK_LOAD_CHUNK(j0 + nsplit * CH)  // this split's next chunk: its loads go out behind the LDS stores that freed the registers
    __syncthreads();
    for (int p = tid >> 2; p < G * cn; p += 64) {  // (head, key) per 4 lanes
      const int g = p / cn, j = p - g * cn, qd = tid & 3;
      const float4* kr4 = (const float4*)(ks + j * KP) + qd * (HD4 / 4);
      const float4* q4 = (const float4*)(qs + g * HD) + qd * (HD4 / 4);
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int e = 0; e < HD4 / 4; e += 2) {
        const float4 ka = kr4[e], qa = q4[e], kb2 = kr4[e + 1], qb = q4[e + 1];
        d0 = __builtin_fmaf(qa.x, ka.x, d0); d0 = __builtin_fmaf(qa.y, ka.y, d0); d0 = __builtin_fmaf(qa.z, ka.z, d0); d0 = __builtin_fmaf(qa.w, ka.w, d0);
        d1 = __builtin_fmaf(qb.x, kb2.x, d1); d1 = __builtin_fmaf(qb.y, kb2.y, d1); d1 = __builtin_fmaf(qb.z, kb2.z, d1); d1 = __builtin_fmaf(qb.w, kb2.w, d1);
      }
      float d = d0 + d1;
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      if (qd == 0) sc[g * CH + j] = d * scale;
    }
    __syncthreads();
    for (int g = wave; g < G; g += 4) {  // running softmax of one head over this chunk (<= 128 keys: two per lane)
      const float v0 = lane < cn ? sc[g * CH + lane] : -INFINITY, v1 = (CH > 64 && lane + 64 < cn) ? sc[g * CH + lane + 64] : -INFINITY;
      float mx = fmaxf(v0, v1);
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float mo = mrun[g], mn = fmaxf(mo, mx);
      const float p0 = lane < cn ? expf(v0 - mn) : 0.f, p1 = (CH > 64 && lane + 64 < cn) ? expf(v1 - mn) : 0.f;
      float sum = p0 + p1;
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      sc[g * CH + lane] = p0;  // (slots >= cn: exact zeros, the weighted sum runs over the padded count)
      if (CH > 64) sc[g * CH + lane + 64] = p1;
      if (lane == 0) {
        const float al = expf(mo - mn);  // exp(-inf) = 0 on the first chunk
        alpha[g] = al;
        mrun[g] = mn;
        lrun[g] = lrun[g] * al + sum;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = tid + 256 * i;
      if (o < G * HD) {
        const int g = o / HD, e = o - g * HD;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int j = 0; j < cn4; j += 4) {
          const float4 pw = *(const float4*)(sc + g * CH + j);
          a0 = __builtin_fmaf(pw.x, vs[j * HD + e], a0);
          a1 = __builtin_fmaf(pw.y, vs[(j + 1) * HD + e], a1);
          a2 = __builtin_fmaf(pw.z, vs[(j + 2) * HD + e], a2);
          a3 = __builtin_fmaf(pw.w, vs[(j + 3) * HD + e], a3);
        }
        acc[i] = acc[i] * alpha[g] + ((a0 + a1) + (a2 + a3));
      }
    }
    __syncthreads();  // ks / vs / sc are rewritten by the next chunk
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = tid + 256 * i;
    if (o < G * HD) {
      if (nsplit == 1) out[(long long)b * H * HD + (long long)kvh * G * HD + o] = acc[i] / lrun[o / HD];
      else pz[(o / HD) * (HD + 2) + o % HD] = acc[i];
    }
  }
  if (nsplit > 1 && tid < G) { pz[tid * (HD + 2) + HD] = mrun[tid]; pz[tid * (HD + 2) + HD + 1] = lrun[tid]; }
}
// out[b][head][:] = sum_z exp(m_z - m) O_z / sum_z exp(m_z - m) l_z over the key splits of attn_decode_kernel (split order; <= 8 splits, all loads at once).
// part is [b][kv head][split][G][HD + 2]: head = kv head * G + g, the splits of one head sit G (HD + 2) floats apart.
template <int HD>
__global__ __launch_bounds__(HD) void attn_merge_kernel(const float* part, int H, int G, int nsplit, float* out) {
  const int h = blockIdx.x, b = blockIdx.y, e = threadIdx.x;
  const int kvh = h / G, g = h - kvh * G;
  const long long zs = (long long)G * (HD + 2);
  const float* base = part + (((long long)b * (H / G) + kvh) * nsplit) * zs + (long long)g * (HD + 2);
  float mz[8], lz[8], oz[8];
#pragma unroll
  for (int zz = 0; zz < 8; ++zz) {
    const float* q = base + (long long)(zz < nsplit ? zz : 0) * zs;
    mz[zz] = zz < nsplit ? q[HD] : -INFINITY;
    lz[zz] = q[HD + 1];
    oz[zz] = q[e];
  }
  float m = mz[0];
#pragma unroll
  for (int zz = 1; zz < 8; ++zz) m = fmaxf(m, mz[zz]);
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int zz = 0; zz < 8; ++zz) {
    const float w = mz[zz] == -INFINITY ? 0.f : expf(mz[zz] - m);  // an empty or absent split weighs nothing
    num = __builtin_fmaf(w, oz[zz], num);
    den = __builtin_fmaf(w, lz[zz], den);
  }
  out[((long long)b * H + h) * HD + e] = den > 0.f ? num / den : 0.f;  // (no key at all: zero, as in the unsplit form)
}
#undef KK_LOAD_CHUNK
static size_t attn_decode_lds_bytes(int hd, int G) {
  const size_t CH = 8192 / hd;
  return (CH * (hd + 4) + CH * hd + (size_t)G * hd + (size_t)G * CH + 2 * hd + 24) * 4;
}

// dynamic LDS of attn_cache_kernel: scores (padded to 4) + q + G partial outputs of hd floats (G = 512 / hd) + the new k and v rows
static size_t attn_lds_bytes(int max_pos, int hd) { return ((size_t)((max_pos + 3) & ~3) + hd + (size_t)(512 / hd) * hd + 2 * hd) * 4; }

__global__ void advance_pos_kernel(int* pos, int by) { *pos += by; }

// silu(gate) * up, gu [rows][2I] -> [rows][I]
__global__ __launch_bounds__(256) void swiglu_kernel(const float* gu, int I, long long n, float* out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const long long r = e / I;
  const int c = (int)(e - r * I);
  const float g = gu[r * 2 * I + c], u = gu[r * 2 * I + I + c];
  out[e] = (g / (1.0f + expf(-g))) * u;
}

// one workgroup per item: argmax (temp == 0 or no uniforms) or inverse CDF over the top_k logits in descending order
// (ties: lower index first) of softmax(logit / temp) with the injected uniform u[b].
// Round-based selection (the fallback of sample_select_kernel below): every thread keeps its V / 256 logits and their running maximum in registers; a round is one wave-shuffle argmax, one
// LDS exchange between the four waves (double-buffered: one barrier per round) and a re-scan by the single thread that owned the
// winner -- ~0.3 us per round instead of a scan of all V logits from LDS plus an eight-level LDS tree (90 -> ~15 us for top-50).
template <int NPER>
__device__ void sample_rounds(float (&v)[NPER], int V, float temp, int top_k, const float* u, int ustride, int* out, int ostride) {
  __shared__ float wv[2][4];
  __shared__ int wi[2][4];
  __shared__ float topv[64];
  __shared__ int topi[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NPER; ++i)
    if (v[i] > bv) { bv = v[i]; bi = tid + 256 * i; }  // ascending index: the lower index wins a tie
  const bool greedy = u == nullptr || temp == 0.f;
  const int k = greedy ? 1 : (top_k < 64 ? (top_k < V ? top_k : V) : 64);
  for (int r = 0; r < k; ++r) {
    float cv = bv;
    int ci = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(cv, o);
      const int oi = __shfl_xor(ci, o);
      if (ov > cv || (ov == cv && oi < ci)) { cv = ov; ci = oi; }
    }
    if (lane == 0) { wv[r & 1][wave] = cv; wi[r & 1][wave] = ci; }
    __syncthreads();
    float gv = wv[r & 1][0];
    int gi = wi[r & 1][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = wv[r & 1][w];
      const int oi = wi[r & 1][w];
      if (ov > gv || (ov == gv && oi < gi)) { gv = ov; gi = oi; }
    }
    if (tid == 0) { topv[r] = gv; topi[r] = gi; }
    if (gi != 0x7fffffff && (gi & 255) == tid) {  // the owner retires the winner and re-scans its own logits
      bv = -INFINITY;
      bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < NPER; ++i) {
        if (i == (gi >> 8)) v[i] = -INFINITY;
        if (v[i] > bv) { bv = v[i]; bi = tid + 256 * i; }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    int pick = topi[0];
    if (!greedy) {
      __shared__ float c[64];  // (LDS, not a private array: scratch memory would be set up for every dispatch of the kernel)
      const float z0 = topv[0] / temp;
      float run = 0.f;
      for (int r = 0; r < k; ++r) {
        run += expf(topv[r] / temp - z0);
        c[r] = run;
      }
      const float target = u[(long long)b * ustride] * run;
      int j = 0;
      while (j < k - 1 && c[j] < target) ++j;
      pick = topi[j];
    }
    out[(long long)b * ostride] = pick;
  }
}
// Top-k by radix SELECT instead of k rounds of arg-max (rounds are serial: 50 x ~1.3 us however they are organised -- a 256-thread
// barrier per round or a 12-shuffle chain per round in one wave).  Logits become order-preserving 32-bit keys in registers; four 8-bit
// histogram passes (LDS atomics, one wave scans the 256 bins from the top) find the key of the k-th largest logit; every logit >= that key
// is a candidate (k of them plus ties of the k-th); ONE wave sorts the <= 64 candidates (bitonic, value descending, index ascending on
// ties: the oracle's order) and the softmax / inverse-CDF tail is the old one.  More than 64 candidates (a wall of exactly equal logits)
// falls back to the round-based kernel.  Same picks as before, bit for bit.
__device__ __forceinline__ unsigned sk_key(float f) {
  const unsigned bits = __float_as_uint(f);
  return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
__device__ __forceinline__ float sk_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

template <int NPER>
__global__ __launch_bounds__(256) void sample_select_kernel(const float* logits, int V, float temp, int top_k, const float* u, int ustride, int* out,
                                                            int ostride) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_krem, s_bin, ccount;
  __shared__ unsigned ckey[64];
  __shared__ int cidx[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned key[NPER];
#pragma unroll
  for (int i = 0; i < NPER; ++i) {
    const int j = tid + 256 * i;
    key[i] = j < V ? sk_key(logits[(long long)b * V + j]) : 0u;
  }
  const bool greedy = u == nullptr || temp == 0.f;
  const int k = greedy ? 1 : (top_k < 64 ? (top_k < V ? top_k : V) : 64);
  if (greedy) {  // arg-max (lower index on ties): the key order is the float order, so one max over (key, ~index) does it -- no histogram
    __shared__ unsigned long long wbest[4];
    unsigned long long best = 0ull;
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
      const int j = tid + 256 * i;
      const unsigned long long c = j < V ? (((unsigned long long)key[i] << 32) | (unsigned)(0x7fffffff - j)) : 0ull;
      best = c > best ? c : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_xor(best, o);
      best = t > best ? t : best;
    }
    if (lane == 0) wbest[wave] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long g = wbest[0];
      for (int w = 1; w < 4; ++w) g = wbest[w] > g ? wbest[w] : g;
      out[(long long)b * ostride] = 0x7fffffff - (int)(unsigned)(g & 0xffffffffull);
    }
    return;
  }
  unsigned prefix = 0u, mask = 0u, krem = (unsigned)k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPER; ++i)
      if (tid + 256 * i < V && (key[i] & mask) == prefix) atomicAdd(&hist[(key[i] >> shift) & 255u], 1u);
    __syncthreads();
    if (wave == 0) {  // lane l owns bins 255 - 4l .. 252 - 4l: ascending lanes walk the digits from the top
      const unsigned c0 = hist[255 - 4 * lane], c1 = hist[254 - 4 * lane], c2 = hist[253 - 4 * lane], c3 = hist[252 - 4 * lane];
      const unsigned sum = c0 + c1 + c2 + c3;
      unsigned inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      const unsigned exc = inc - sum;
      if (exc < krem && krem <= inc) {  // exactly one lane: the digit that holds the krem-th largest of the surviving keys
        const unsigned r = krem - exc;
        unsigned bin, above;
        if (r <= c0) { bin = 255 - 4 * lane; above = exc; }
        else if (r <= c0 + c1) { bin = 254 - 4 * lane; above = exc + c0; }
        else if (r <= c0 + c1 + c2) { bin = 253 - 4 * lane; above = exc + c0 + c1; }
        else { bin = 252 - 4 * lane; above = exc + c0 + c1 + c2; }
        s_prefix = prefix | (bin << shift);
        s_krem = krem - above;
        s_bin = hist[bin];
      }
    }
    __syncthreads();
    prefix = s_prefix;
    krem = s_krem;
    mask |= 255u << shift;
    // every key >= prefix (lower digits zero) is a candidate: the k - krem keys above the chosen digit and the s_bin keys that share it.  Once
    // they fit the 64-entry sort the remaining digits need not be resolved (typically after two passes: 16 bits separate the top-50 of 2051 logits)
    if ((unsigned)k - krem + s_bin <= 64u) break;
  }
  if (tid == 0) ccount = 0u;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NPER; ++i)
    if (tid + 256 * i < V && key[i] >= prefix) {
      const unsigned p = atomicAdd(&ccount, 1u);
      if (p < 64u) { ckey[p] = key[i]; cidx[p] = tid + 256 * i; }
    }
  __syncthreads();
  const unsigned nc = ccount;
  if (nc > 64u) {  // block-uniform: a wall of equal logits -- the round-based selection handles any input
    float v[NPER];
#pragma unroll
    for (int i = 0; i < NPER; ++i) v[i] = tid + 256 * i < V ? sk_unkey(key[i]) : -INFINITY;
    sample_rounds<NPER>(v, V, temp, top_k, u, ustride, out, ostride);
    return;
  }
  if (wave != 0) return;
  unsigned kk = lane < (int)nc ? ckey[lane] : 0u;
  int ii = lane < (int)nc ? cidx[lane] : 0x7fffffff;
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const unsigned ok = __shfl_xor(kk, stride);
      const int oi = __shfl_xor(ii, stride);
      const bool other_first = ok > kk || (ok == kk && oi < ii);  // `other` precedes `mine` in the wanted order
      const bool want_first = ((lane & stride) == 0) == ((lane & size) == 0 || size == 64);  // this lane keeps the earlier element
      if (other_first == want_first) { kk = ok; ii = oi; }
    }
  // lane r holds the r-th candidate.  The cumulative sums run in candidate order (the oracle's cumsum) on values read lane by lane
  // (v_readlane with a uniform index): the whole wave walks the same scalar chain, no scratch array, no LDS round trips
  const float tv = sk_unkey(kk);
  auto lane_f = [](float v, int l) __attribute__((always_inline)) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); };
  const float t0 = lane_f(tv, 0);
  const float e = (!greedy && lane < k) ? expf(tv / temp - t0 / temp) : 0.f;
  int pick = __builtin_amdgcn_readlane(ii, 0);
  if (!greedy) {
    float run = 0.f;
    for (int r = 0; r < k; ++r) run += lane_f(e, r);
    const float target = u[(long long)b * ustride] * run;
    int j = 0;
    float c = lane_f(e, 0);
    while (j < k - 1 && c < target) {
      ++j;
      c += lane_f(e, j);
    }
    pick = __builtin_amdgcn_readlane(ii, __builtin_amdgcn_readfirstlane(j));
  }
  if (lane == 0) out[(long long)b * ostride] = pick;
}

int launch_sample(const float* logits, int V, float temp, int top_k, const float* u, int ustride, int* out, int ostride, int B, hipStream_t st) {
  if (V <= 256 * 4) hipLaunchKernelGGL(sample_select_kernel<4>, dim3(B), dim3(256), 0, st, logits, V, temp, top_k, u, ustride, out, ostride);
  else if (V <= 256 * 9) hipLaunchKernelGGL(sample_select_kernel<9>, dim3(B), dim3(256), 0, st, logits, V, temp, top_k, u, ustride, out, ostride);
  else if (V <= 256 * 32) hipLaunchKernelGGL(sample_select_kernel<32>, dim3(B), dim3(256), 0, st, logits, V, temp, top_k, u, ustride, out, ostride);
  else return kk_fail("kk_csm: audio vocabulary larger than 8192 entries");
  KK_CHECK_LAUNCH();
  return 0;
}

// Skinny GEMM for the single-token steps (M = B rows <= 16): out[m][n] = sum_k x[m][k] W[k][n].  The grid is (column blocks) x (KS
// slices of K): every thread streams ONE column (fp32 weights) or TWO adjacent columns (bf16 weights, one 4-byte load) of its K slice
// with coalesced row reads, 16 loads in flight; the slices' partial sums are added in slice order by the second kernel
// (deterministic, residual fused).  The rows are held as PACKED PAIRS: xs[k][m] keeps the M activations of one k adjacent, so a
// weight meets rows (2i, 2i+1) in one v_pk_fma_f32 -- MT/2 packed FMAs per weight (MT = 8 or 16 rows, a template parameter).  The
// first version issued 16 predicated scalar FMAs per weight whatever M was and was VALU-bound (K*N*16 lane-FMAs: 13.6 us for the
// backbone's gate|up matrix against a 13 us HBM floor in bf16), which is why halving the weight bytes did not pay before.
// `GATED`: x is the gate|up pair of a SwiGLU MLP ([M][2K]) and the staged input is silu(gate) * up (the stand-alone swiglu kernel of
// the single-token step disappears).
constexpr int SK_MAXM = 16, SK_KC = 256;
typedef float sk2f __attribute__((ext_vector_type(2)));
template <int MT, bool BF16W, bool GATED>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const float* x, int M, int K, const void* wv_, int ldw, int N, int kchunk, float* part) {
  constexpr int NC = BF16W ? 2 : 1;  // columns per thread
  constexpr int MP = MT / 2;         // row pairs
  __shared__ __attribute__((aligned(16))) float xs[SK_KC][MT];
  const int n = (blockIdx.x * 256 + threadIdx.x) * NC, ks = blockIdx.y;
  const int k0 = ks * kchunk, k1 = min(K, k0 + kchunk);
  sk2f acc[NC][MP];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < MP; ++i) acc[c][i] = sk2f{0.f, 0.f};
  for (int kb = k0; kb < k1; kb += SK_KC) {
    const int kn = min(SK_KC, k1 - kb);
    __syncthreads();
    for (int e = threadIdx.x; e < MT * kn; e += 256) {
      const int m = e / kn, k = e - m * kn;  // consecutive threads read consecutive k of one row
      float v = 0.f;
      if (m < M) {
        if (GATED) {
          const float g = x[(long long)m * 2 * K + kb + k], u = x[(long long)m * 2 * K + K + kb + k];
          v = g / (1.0f + expf(-g)) * u;
        } else {
          v = x[(long long)m * K + kb + k];
        }
      }
      xs[k][m] = v;
    }
    __syncthreads();
    if (n < N) {
      const unsigned* wp = BF16W ? (const unsigned*)((const uint16_t*)wv_ + (long long)kb * ldw + n) : (const unsigned*)((const float*)wv_ + (long long)kb * ldw + n);
      const long long ldq = BF16W ? (ldw >> 1) : ldw;  // row pitch in 4-byte words
      auto fma_k = [&](unsigned wbits, int k) {
        const sk2f* xr = (const sk2f*)&xs[k][0];
        if (BF16W) {
          const float w0 = __uint_as_float(wbits << 16), w1 = __uint_as_float(wbits & 0xffff0000u);
#pragma unroll
          for (int i = 0; i < MP; ++i) {
            const sk2f xv = xr[i];
            acc[0][i] = __builtin_elementwise_fma(xv, sk2f{w0, w0}, acc[0][i]);
            acc[NC - 1][i] = __builtin_elementwise_fma(xv, sk2f{w1, w1}, acc[NC - 1][i]);
          }
        } else {
          const float w0 = __uint_as_float(wbits);
#pragma unroll
          for (int i = 0; i < MP; ++i) acc[0][i] = __builtin_elementwise_fma(xr[i], sk2f{w0, w0}, acc[0][i]);
        }
      };
      int k = 0;
      for (; k + 16 <= kn; k += 16) {  // 16 independent loads in flight per thread
        unsigned wv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) wv[j] = wp[(long long)(k + j) * ldq];
#pragma unroll
        for (int j = 0; j < 16; ++j) fma_k(wv[j], k + j);
      }
      for (; k < kn; ++k) fma_k(wp[(long long)k * ldq], k);
    }
  }
  if (n < N) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (n + c >= N) continue;
#pragma unroll
      for (int i = 0; i < MP; ++i) {
        // slices of one output element are contiguous
        if (2 * i < M) part[((long long)(2 * i) * N + n + c) * gridDim.y + ks] = acc[c][i].x;
        if (2 * i + 1 < M) part[((long long)(2 * i + 1) * N + n + c) * gridDim.y + ks] = acc[c][i].y;
      }
    }
  }
}

__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* part, int KS, int M, int N, const float* res, float* out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)M * N) return;
  const float* p = part + e * KS;
  float v = 0.f;
  int ks = 0;
  if ((KS & 3) == 0) {
    for (; ks < KS; ks += 4) {
      const float4 q = *(const float4*)(p + ks);
      v += q.x; v += q.y; v += q.z; v += q.w;  // slice order
    }
  }
  for (; ks < KS; ++ks) v += p[ks];
  if (res) v += res[e];
  out[e] = v;
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Fused GEMV of the single-token steps (bf16 weight mode): out[m][n] = sum_k f(x)[m][k] W[k][n] for M <= 16 rows, with NO split-K through
// HBM and no separate reduce / norm / activation launches -- a Llama layer is five launches (qkv, attention, o, gate|up, down).
//   * a workgroup owns CB = 8 * OCT output columns for ALL of K; the weights are packed [N / CB][K][CB] (bf16), so it streams one contiguous
//     region: lane = (k-sub, column octet) takes 16 bytes = 8 columns of one k row, a wave instruction covers 64 / OCT consecutive k rows
//     (1 KiB contiguous), U of them in flight per thread; OCT = 1 for the narrow matrices (>= 128 workgroups even for N = 1024), 8 for gate|up;
//   * the input rows live in LDS as xs[row quad][k][4] fp32, staged per 1024-row K chunk with the PROLOGUE applied on the way in:
//       PRO 0 plain rows; PRO 1 RMSNorm(x) * w (every workgroup recomputes the row norms: M x K floats from L2, against its own 16-256 KB of
//       weights); PRO 2 silu(gate) * up of a gate|up pair; PRO 3 rows gathered from the audio embedding table by code (the depth decoder's
//       input `curr`, sesame.py:373-392), for two-row items the first row from x;
//   * a weight meets all rows in packed fp32 FMAs (accumulators acc[8 columns][MT / 2 row pairs]); per lane the k order is fixed by the
//     layout, so a row's bits do not depend on its batch neighbours or on MT;
//   * the lanes' / waves' partial sums meet in LDS in a fixed order; EPI 1 adds the residual (h += ...) in place.
// h[m][n] += sum over the K slices of a split-K launch (slice order): the combine of the deep down projections
__global__ __launch_bounds__(256) void combine_slices_kernel(const float* part, int KS, long long pss, long long n, float* h, unsigned long long* ts) {
  ts_begin(ts, 2);
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  // all slices are requested at once (a plain loop is KS dependent L2 round trips); KS <= 16 (gemv_slices)
  float v[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) v[ks] = part[(long long)(ks < KS ? ks : 0) * pss + e];
  const float h0 = h[e];
  float t = 0.f;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) t += ks < KS ? v[ks] : 0.f;
  h[e] = h0 + t;
  ts_end(ts);
}

// sum over the 32 lanes of a half wave (butterfly)
__device__ __forceinline__ float q32_sum(float v) {
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 16-byte weight load that is not kept in the caches (a frame reads every matrix once; MI355X_MICROARCH.md nt-weights: -5 ... -10 % per layer)
__device__ __forceinline__ uint4 ld_w_nt(const uint4* p, int dbg = 0) {
  if (dbg & 8) return make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// NPRE: weight loads a wave keeps in flight (requested BEFORE the prologue, then a ring in the main loop).  Round 2 had 8 before the prologue
// and groups of FG_U = 4 afterwards, each group an exposed HBM round trip: 16-32 KB outstanding per CU, gate|up 33.5 MB in 16 us = 2.1 TB/s.
template <int MT, int OCT, int PRO, int EPI, int NPRE>
__global__ __launch_bounds__(256) void fused_gemv_kernel(FGArgs a) {
  constexpr int CB = 8 * OCT, KSUB = 64 / OCT, MP = MT / 2, NQ = MT / 4, OUT = CB * 8;
  constexpr int KCH = 16384 / MT;           // k rows staged at a time: xs is 64 KB (2048 rows for 8 input rows, 1024 for 16)
  constexpr int NH = (PRO == 2 && KCH / 4 / 256 >= 2) ? 2 : 1;  // the gated prologue holds two values per item: two half passes where there is more than one quad per thread
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                                   // [NQ][KCH][4]
  float* red = sm;                                  // [4 waves][KSUB][OUT] (aliases xs after the main loop)
  float* red2 = sm + 4 * KSUB * OUT;                // [4][OUT]
  float* rs = red2 + 4 * OUT;                       // [16] row scales (PRO 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int oct = lane % OCT, ksub = lane / OCT;
  const int nb = blockIdx.x;
  const int K = a.K, M = a.M;
  // split-K (gridDim.y > 1, the deep down projections): this workgroup owns k rows [k_lo, k_hi) and writes a partial tile
  const int Kper = K / gridDim.y, k_lo = blockIdx.y * Kper, k_hi = k_lo + Kper;
  const uint4* wblk = (const uint4*)(a.w + (long long)nb * K * CB) + oct;  // row k of the block at + k * OCT uint4
  // ---- the first NPRE weight loads of this wave go out before anything else: they do not depend on the input rows
  uint4 wpre[NPRE];
  {
    const int kn0 = min(KCH, k_hi - k_lo), nL0 = kn0 / KSUB;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int L = wave + 4 * i;
      wpre[i] = ld_w_nt(wblk + (long long)(k_lo + (L < nL0 ? L : wave % nL0) * KSUB + ksub) * OCT, a.dbg);
    }
  }
  // input row m of this launch as an element offset from its base (PRO 3: an item's last row comes from the audio embedding table,
  // sesame.py:373-392, its other rows from x)
  int rowoff[MT];
  bool rowemb[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mm = m < M ? m : 0;
    rowemb[m] = false;
    if (PRO == 3) {
      const int item = mm / a.rows, r = mm - item * a.rows;
      rowemb[m] = r == a.rows - 1;
      rowoff[m] = rowemb[m] ? (clamp_id(a.codes[(long long)item * a.cstride], a.V) + a.cb * a.V) * K : (int)(item * a.xrs);
    } else {
      rowoff[m] = (int)(mm * a.xrs);
    }
  }
  auto rowptr = [&](int m) __attribute__((always_inline)) -> const float* { return (PRO == 3 && rowemb[m] ? a.emb : a.x) + rowoff[m]; };
  if (PRO == 1 && K > KCH) {  // rows longer than a chunk (16 input rows of the backbone): the row scales need their own pass
    float ssr[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ssr[m] = 0.f;
    const int n4 = K >> 2;
    for (int c0 = tid; c0 < n4; c0 += 256 * 2) {
      float4 v[2][MT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) v[i][m] = *((const float4*)rowptr(m) + (c0 + 256 * i < n4 ? c0 + 256 * i : c0));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float live = c0 + 256 * i < n4 ? 1.0f : 0.0f;
#pragma unroll
        for (int m = 0; m < MT; ++m) ssr[m] += live * (((v[i][m].x * v[i][m].x + v[i][m].y * v[i][m].y) + v[i][m].z * v[i][m].z) + v[i][m].w * v[i][m].w);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float t = ssr[m];
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (lane == 0) red2[wave * 16 + m] = t;
    }
    __syncthreads();
    if (tid < MT) rs[tid] = 1.0f / sqrtf((((red2[tid] + red2[16 + tid]) + red2[32 + tid]) + red2[48 + tid]) / (float)K + a.eps);
  }
  float ssq[MT];  // PRO 1 with the whole row in one chunk: sums of squares of this thread's share of every input row
#pragma unroll
  for (int m = 0; m < MT; ++m) ssq[m] = 0.f;
  sk2f acc[8][MP];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < MP; ++i) acc[c][i] = sk2f{0.f, 0.f};
  for (int kc = k_lo; kc < k_hi; kc += KCH) {
    const int kn = min(KCH, k_hi - kc);
    __syncthreads();  // (the previous chunk's readers are done)
    // ---- stage the chunk (round 3: 16-byte loads): thread t takes the k QUADS t, t + 256, ... of EVERY input row (one float4 per row, coalesced
    // along k; round 2 read single floats, half of them clamped duplicates) and writes one 16-byte LDS store per (k, row quad).  With the norm
    // prologue and the whole row in this chunk the row scale is NOT on the way in: x * w is staged, the sums of squares stay in registers through
    // the main loop and rms^-1 multiplies the finished dot products (defer_rs): no reduction + two barriers ahead of the weight stream.
    const bool defer_rs = PRO == 1 && K <= KCH;
    if (!(a.dbg & 1))
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      constexpr int NI4 = KCH / 4 / 256 / NH;  // k quads per thread and pass
      float4 g[NI4][MT], u[PRO == 2 ? NI4 : 1][MT], nw4[PRO == 1 ? NI4 : 1];
      const int nq = kn >> 2;
#pragma unroll
      for (int i = 0; i < NI4; ++i) {
        const int k4 = tid + 256 * (hh * NI4 + i);
        const int kk = kc + 4 * (k4 < nq ? k4 : tid % nq);
        if (PRO == 1) nw4[i] = *(const float4*)(a.nw + kk);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          g[i][m] = *(const float4*)(rowptr(m) + kk);
          if (PRO == 2) u[i][m] = *(const float4*)(rowptr(m) + K + kk);
        }
      }
#pragma unroll
      for (int i = 0; i < NI4; ++i) {
        const int k4 = tid + 256 * (hh * NI4 + i);
        if (k4 < nq) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float4 t = g[i][m];
            if (PRO == 2) {
              const float4 uu = u[i][m];
              t.x = t.x / (1.0f + expf(-t.x)) * uu.x; t.y = t.y / (1.0f + expf(-t.y)) * uu.y;
              t.z = t.z / (1.0f + expf(-t.z)) * uu.z; t.w = t.w / (1.0f + expf(-t.w)) * uu.w;
            } else if (PRO == 1) {
              if (defer_rs) {
                ssq[m] = __builtin_fmaf(t.x, t.x, ssq[m]); ssq[m] = __builtin_fmaf(t.y, t.y, ssq[m]);
                ssq[m] = __builtin_fmaf(t.z, t.z, ssq[m]); ssq[m] = __builtin_fmaf(t.w, t.w, ssq[m]);
                t.x *= nw4[i].x; t.y *= nw4[i].y; t.z *= nw4[i].z; t.w *= nw4[i].w;
              } else {
                const float r = rs[m];
                t.x = t.x * r * nw4[i].x; t.y = t.y * r * nw4[i].y; t.z = t.z * r * nw4[i].z; t.w = t.w * r * nw4[i].w;
              }
            }
            if (m >= M) t = make_float4(0.f, 0.f, 0.f, 0.f);
            g[i][m] = t;
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            float* d = xs + ((long long)q * KCH + 4 * k4) * 4;
            *(float4*)(d) = make_float4(g[i][4 * q].x, g[i][4 * q + 1].x, g[i][4 * q + 2].x, g[i][4 * q + 3].x);
            *(float4*)(d + 4) = make_float4(g[i][4 * q].y, g[i][4 * q + 1].y, g[i][4 * q + 2].y, g[i][4 * q + 3].y);
            *(float4*)(d + 8) = make_float4(g[i][4 * q].z, g[i][4 * q + 1].z, g[i][4 * q + 2].z, g[i][4 * q + 3].z);
            *(float4*)(d + 12) = make_float4(g[i][4 * q].w, g[i][4 * q + 1].w, g[i][4 * q + 2].w, g[i][4 * q + 3].w);
          }
        }
      }
    }
    __syncthreads();
    const int nL = kn / KSUB;  // wave loads in this chunk; wave w takes L = w, w + 4, ...
    auto fma_row = [&](const uint4& wq, int kl, float live) __attribute__((always_inline)) {
      if (a.dbg & 2) { acc[0][0].x += __uint_as_float(wq.x ^ wq.y ^ wq.z ^ wq.w) * live; return; }
      sk2f xv[MP];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 t = *(const float4*)(xs + ((long long)q * KCH + kl) * 4);
        xv[2 * q] = sk2f{t.x, t.y};
        xv[2 * q + 1] = sk2f{t.z, t.w};
      }
      const unsigned wd[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) {
        const float w0 = __uint_as_float(wd[c2] << 16) * live, w1 = __uint_as_float(wd[c2] & 0xffff0000u) * live;
#pragma unroll
        for (int mp = 0; mp < MP; ++mp) {
          acc[2 * c2][mp] = __builtin_elementwise_fma(xv[mp], sk2f{w0, w0}, acc[2 * c2][mp]);
          acc[2 * c2 + 1][mp] = __builtin_elementwise_fma(xv[mp], sk2f{w1, w1}, acc[2 * c2 + 1][mp]);
        }
      }
    };
    // The weight stream of this wave is a RING of NPRE loads in flight (round 3): slot i holds wave load L0 + 4 i; it is refilled with load
    // L0 + 4 (i + NPRE) right before its value is used, so NPRE x 1 KiB per wave (64 KiB per CU at NPRE = 16) stay outstanding for the whole
    // chunk -- vmcnt retires in order, hipcc waits for exactly the oldest.  The first chunk's ring was requested before the prologue (wpre).
    if (kc != k_lo) {
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int L = wave + 4 * i;
        wpre[i] = ld_w_nt(wblk + (long long)(kc + (L < nL ? L : wave % nL) * KSUB + ksub) * OCT, a.dbg);
      }
    }
    int L0 = wave;
    for (; L0 + 4 * NPRE < nL; L0 += 4 * NPRE) {  // every slot of this round has a successor (clamped in the last round of a ragged count)
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int L = L0 + 4 * i, Ln = L + 4 * NPRE;
        const uint4 wq = wpre[i];
        wpre[i] = ld_w_nt(wblk + (long long)(kc + (Ln < nL ? Ln : L) * KSUB + ksub) * OCT, a.dbg);
        fma_row(wq, L * KSUB + ksub, 1.0f);
      }
    }
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {  // the last ring-full: nothing left to request
      const int L = L0 + 4 * i;
      fma_row(wpre[i], (L < nL ? L : wave % nL) * KSUB + ksub, L < nL ? 1.0f : 0.0f);
    }
  }
  if (a.dbg & 4) {  // (timing only)
    if (acc[0][0].x == 123.456f) a.out[tid] = acc[1][1].y;
    return;
  }
  // deferred RMSNorm scale: the waves' sums of squares -> LDS (visible behind the barriers below), applied to the finished dot products
  const bool defer_rs_out = PRO == 1 && K <= KCH;
  float* rsp = rs + 16;  // [4 waves][16]
  if (defer_rs_out) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float t = ssq[m];
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (lane == 0) rsp[wave * 16 + m] = t;
    }
  }
  // ---- reduction over the lanes that share columns (k-sub) and the 4 waves, 8 rows per pass; output o = mloc * CB + column
#pragma unroll
  for (int mh = 0; mh < MT / 8; ++mh) {
    __syncthreads();  // the chunk (or the previous pass) is no longer read
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* d = red + ((long long)(wave * KSUB + ksub)) * OUT + oct * 8 + c;
        d[(2 * i) * CB] = acc[c][4 * mh + i].x;
        d[(2 * i + 1) * CB] = acc[c][4 * mh + i].y;
      }
    __syncthreads();
    for (int pr = tid; pr < 4 * OUT; pr += 256) {  // (wave q, output o): the q-th wave's KSUB lanes in k-sub order
      const int q = pr / OUT, o = pr - q * OUT;
      const float* sp = red + (long long)q * KSUB * OUT + o;
      float t = 0.f;
#pragma unroll 8
      for (int ks = 0; ks < KSUB; ++ks) t += sp[(long long)ks * OUT];
      red2[pr] = t;
    }
    __syncthreads();
    for (int o = tid; o < OUT; o += 256) {
      const int mloc = o / CB, col = o - mloc * CB;
      const int m = 8 * mh + mloc, n = nb * CB + col;
      if (m < M && n < a.N) {
        float t = ((red2[o] + red2[OUT + o]) + red2[2 * OUT + o]) + red2[3 * OUT + o];  // wave order
        if (defer_rs_out) t *= 1.0f / sqrtf((((rsp[m] + rsp[16 + m]) + rsp[32 + m]) + rsp[48 + m]) / (float)K + a.eps);
        if (EPI == 1) t += a.res[(long long)m * a.rrs + n];
        if (EPI == 2) a.out[(long long)blockIdx.y * a.pss + (long long)m * a.ors + n] = t;  // K slice blockIdx.y of a split-K launch
        else a.out[(long long)m * a.ors + n] = t;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// gemv8_kernel (round 3): the same product for M <= 8 rows and a single K chunk (K per slice <= 2048, a multiple of 128) in COMPACT code.
// What the phase ablation of fused_gemv_kernel showed (KK_CSM_DBG): with the input staging, the FMAs, the reduction AND the
// weight loads all switched off a frame still took 6.1 of 9.6 ms -- ~6 us per launch of an empty kernel -- and the time followed the
// amount of straight-line CODE on the executed path, not the work: the kernels are 16-38 KB of fully unrolled instructions that run
// once per launch, i.e. a launch is bound by instruction fetch from a cold instruction cache.  This kernel keeps the arithmetic and its
// order (a row's bits are those of fused_gemv_kernel up to where the RMSNorm scale is applied) and shrinks the code:
//   * staging: thread (row = tid / 32, lane32) takes the k quads lane32, lane32 + 32, ... of ONE row (8 float4 in flight per pass) instead of
//     one column of every row; sums of squares stay per row half-wave (5 shuffle steps once, not 6 per row); fast exp / reciprocal in SwiGLU;
//   * weight stream: ring of 8 loads per wave, a ROLLED loop over rounds of 8 (use + refill), then one use-only round;
//   * the RMSNorm scale multiplies the finished dot products (no reduction + barriers ahead of the weight stream).
template <int OCT, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv8_kernel(FGArgs a) {
  constexpr int CB = 8 * OCT, KSUB = 64 / OCT, OUT = CB * 8, KCH = 2048, RING = 8;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                     // [8 rows][KCH] row-major: staged by conflict-free 16-byte stores, read as 8 broadcast words per k
  float* red = sm;                    // [4 waves][KSUB][OUT] (aliases xs after the main loop)
  float* red2 = sm + 4 * KSUB * OUT;  // [4][OUT]
  float* rsq = red2 + 4 * OUT;        // [8] sums of squares of the input rows (PRO 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int oct = lane % OCT, ksub = lane / OCT;
  const int nb = blockIdx.x, K = a.K, M = a.M;
  const int Kper = K / gridDim.y, k_lo = blockIdx.y * Kper;
  const int nL = Kper / KSUB;  // wave loads of this workgroup; wave w takes L = w, w + 4, ...
  const uint4* wblk = (const uint4*)(a.w + (long long)nb * K * CB) + oct + (long long)k_lo * OCT;  // row k of the slice at + k * OCT uint4
  // ---- the ring's first loads go out before anything else
  uint4 ring[RING];
#pragma unroll
  for (int i = 0; i < RING; ++i) {
    const int L = wave + 4 * i;
    ring[i] = ld_w_nt(wblk + (long long)((L < nL ? L : wave % nL) * KSUB + ksub) * OCT, a.dbg);
  }
  // ---- stage the input rows
  {
    const int m = tid >> 5, l32 = tid & 31, mm = m < M ? m : 0;
    const float* row;
    if (PRO == 3) {  // an item's last row comes from the audio embedding table (sesame.py:373-392), its other rows from x
      const int item = mm / a.rows, r = mm - item * a.rows;
      row = r == a.rows - 1 ? a.emb + (long long)(clamp_id(a.codes[(long long)item * a.cstride], a.V) + a.cb * a.V) * K : a.x + (long long)item * a.xrs;
    } else {
      row = a.x + (long long)mm * a.xrs;
    }
    float ssq = 0.f;
    float* dst = xs + (long long)m * KCH;
    const int nq = Kper >> 2;  // k quads of the slice: a multiple of 32 (launcher)
    for (int j0 = 0; j0 < nq; j0 += 32 * 8) {  // 8 quads per thread in flight (K = 1024: one pass)
      float4 g[8], u[PRO == 2 ? 8 : 1], nw[PRO == 1 ? 8 : 1];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = j0 + l32 + 32 * j, qc = q < nq ? q : l32;
        g[j] = *(const float4*)(row + k_lo + 4 * qc);
        if (PRO == 2) u[j] = *(const float4*)(row + K + k_lo + 4 * qc);
        if (PRO == 1) nw[j] = *(const float4*)(a.nw + k_lo + 4 * qc);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = j0 + l32 + 32 * j;
        float4 t = g[j];
        if (PRO == 2) {  // silu(gate) * up
          t.x = t.x * __builtin_amdgcn_rcpf(1.0f + __expf(-t.x)) * u[j].x; t.y = t.y * __builtin_amdgcn_rcpf(1.0f + __expf(-t.y)) * u[j].y;
          t.z = t.z * __builtin_amdgcn_rcpf(1.0f + __expf(-t.z)) * u[j].z; t.w = t.w * __builtin_amdgcn_rcpf(1.0f + __expf(-t.w)) * u[j].w;
        } else if (PRO == 1) {
          const float live = q < nq ? 1.0f : 0.0f;  // (a clamped duplicate of a ragged pass does not count)
          ssq = __builtin_fmaf(t.x * live, t.x, ssq); ssq = __builtin_fmaf(t.y * live, t.y, ssq);
          ssq = __builtin_fmaf(t.z * live, t.z, ssq); ssq = __builtin_fmaf(t.w * live, t.w, ssq);
          t.x *= nw[j].x; t.y *= nw[j].y; t.z *= nw[j].z; t.w *= nw[j].w;
        }
        if (m >= M) t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) *(float4*)(dst + 4 * q) = t;
      }
    }
    if (PRO == 1) {
      ssq = q32_sum(ssq);
      if (l32 == 0) rsq[m] = ssq;  // (its own LDS words: visible behind the barriers below)
    }
  }
  sk2f acc[8][4];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = sk2f{0.f, 0.f};
  __syncthreads();
  auto fma_row = [&](const uint4& wq, int kl, float live) __attribute__((always_inline)) {
    const float* xp = xs + kl;
    const sk2f xv[4] = {sk2f{xp[0], xp[KCH]}, sk2f{xp[2 * KCH], xp[3 * KCH]}, sk2f{xp[4 * KCH], xp[5 * KCH]}, sk2f{xp[6 * KCH], xp[7 * KCH]}};
    const unsigned wd[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      const float w0 = __uint_as_float(wd[c2] << 16) * live, w1 = __uint_as_float(wd[c2] & 0xffff0000u) * live;
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) {
        acc[2 * c2][mp] = __builtin_elementwise_fma(xv[mp], sk2f{w0, w0}, acc[2 * c2][mp]);
        acc[2 * c2 + 1][mp] = __builtin_elementwise_fma(xv[mp], sk2f{w1, w1}, acc[2 * c2 + 1][mp]);
      }
    }
  };
  int L0 = wave;
#pragma unroll 1
  for (; L0 + 4 * RING < nL; L0 += 4 * RING) {  // every slot of this round has a successor (clamped in the last round of a ragged count)
#pragma unroll
    for (int i = 0; i < RING; ++i) {
      const int L = L0 + 4 * i, Ln = L + 4 * RING;
      const uint4 wq = ring[i];
      ring[i] = ld_w_nt(wblk + (long long)((Ln < nL ? Ln : L) * KSUB + ksub) * OCT, a.dbg);
      fma_row(wq, L * KSUB + ksub, 1.0f);
    }
  }
#pragma unroll
  for (int i = 0; i < RING; ++i) {  // the last ring-full: nothing left to request
    const int L = L0 + 4 * i;
    fma_row(ring[i], (L < nL ? L : wave % nL) * KSUB + ksub, L < nL ? 1.0f : 0.0f);
  }
  // ---- reduction over the lanes that share columns (k-sub) and the 4 waves; output o = m * CB + column
  __syncthreads();  // xs is no longer read
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* d = red + ((long long)(wave * KSUB + ksub)) * OUT + oct * 8 + c;
      d[(2 * i) * CB] = acc[c][i].x;
      d[(2 * i + 1) * CB] = acc[c][i].y;
    }
  __syncthreads();
  for (int pr = tid; pr < 4 * OUT; pr += 256) {  // (wave q, output o): the q-th wave's KSUB lanes in k-sub order
    const int q = pr / OUT, o = pr - q * OUT;
    const float* sp = red + (long long)q * KSUB * OUT + o;
    float t = 0.f;
#pragma unroll 8
    for (int ks = 0; ks < KSUB; ++ks) t += sp[(long long)ks * OUT];
    red2[pr] = t;
  }
  __syncthreads();
  for (int o = tid; o < OUT; o += 256) {
    const int m = o / CB, col = o - m * CB, n = nb * CB + col;
    if (m < M && n < a.N) {
      float t = ((red2[o] + red2[OUT + o]) + red2[2 * OUT + o]) + red2[3 * OUT + o];  // wave order
      if (PRO == 1) t *= 1.0f / sqrtf(rsq[m] / (float)K + a.eps);
      if (EPI == 1) t += a.res[(long long)m * a.rrs + n];
      if (EPI == 2) a.out[(long long)blockIdx.y * a.pss + (long long)m * a.ors + n] = t;  // K slice blockIdx.y of a split-K launch
      else a.out[(long long)m * a.ors + n] = t;
    }
  }
}
template <int OCT>
static size_t g8_lds_bytes() {
  constexpr size_t xsb = (size_t)65536, redb = (size_t)4 * (64 / OCT) * (8 * OCT * 8) * 4;
  return (xsb > redb ? xsb : redb) + (size_t)4 * (8 * OCT * 8) * 4 + 64;
}

template <int MT, int OCT>
static size_t fg_lds_bytes() {
  constexpr size_t xsb = (size_t)65536, redb = (size_t)4 * (64 / OCT) * (8 * OCT * 8) * 4;
  return (xsb > redb ? xsb : redb) + (size_t)4 * (8 * OCT * 8) * 4 + 64 + 256;  // + rs [16] + the waves' sums of squares [4][16]
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// gemmp_kernel (round 3): the PROMPT block's Linear layers (M = B x S rows, hundreds to thousands) on the matrix cores, from the same bf16
// fragment pack and with the same exact three-way bf16 split of the fp32 input as gemvm_kernel -- out[M][N] = x[M][K] W (+ res) in
// fp32 arithmetic on bf16 weights.  Round 2 ran the prompt through the library's generic fp32 conv kernel (64 launches, 37 ms for a
// 64-token prompt at B = 8).  A workgroup (4 waves) owns 64 rows x 64 columns (four 16-column sub-blocks, whatever the pack's nsub) and
// walks K in 32-row chunks: the A operands of a chunk (4 row tiles x 3 terms x 1 KiB) are split on the way from global memory into a
// double-buffered LDS tile while the matrix instructions of the previous chunk run; wave w multiplies row tile w with all four
// B fragments (16-byte loads, one chunk ahead).  The three terms of a row accumulate into the SAME accumulator (x1 W + x2 W + x3 W), so a
// row's result depends on nothing but its own input row: the prompt block is batch-invariant.
struct GPArgs {
  const float* x; long long xrs;
  const uint16_t* w;
  int K, N, M, nsub;
  const float* res; long long rrs;
  float* out; long long ors;
};
template <int RT, int KC>  // RT row tiles of 16 per workgroup (4: one per wave, 8: two per wave); KC 32-row K chunks per barrier (1 or 2)
__global__ __launch_bounds__(256) void gemmp_kernel(GPArgs a) {
  constexpr int NIT = RT / 4;      // row tiles per wave
  constexpr int NLD = RT / 2;      // float4 loads per thread and chunk: (16 RT rows) x (8 quads of 4 k) / 256 threads
  constexpr int KOP = 384, FRAGB = 4 * KOP;  // k-octet pitch in bytes (256 of data: the two k octets a wave's 8-byte writes touch share no bank), fragment bytes
  constexpr int CHB = RT * 3 * FRAGB;        // one chunk's A operands: [row tile][term][k octet][16 rows][8 bf16]
  extern __shared__ __attribute__((aligned(16))) unsigned char abuf[];  // [2 buffers][KC chunks][CHB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * (16 * RT), sb0 = blockIdx.x * 4, nch = a.K >> 5;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // B fragment of (chunk c, global sub-block sb): block sb / nsub, sub sb % nsub of the pack [block][chunk][sub][lane]
  const int nsbt = (a.N + 15) >> 4;
  const u32x4* bp[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int sb = sb0 + s < nsbt ? sb0 + s : nsbt - 1;  // (a sub-block past the end repeats the last one; its columns are not stored)
    bp[s] = (const u32x4*)a.w + ((long long)(sb / a.nsub) * nch * a.nsub + sb % a.nsub) * 64 + lane;
  }
  const long long bstep = (long long)a.nsub * 64;  // per chunk
  // A staging: thread (row tid / 8 + 32 it, quad tid % 8) takes 4 consecutive k of its row: the 8 lanes of a row read one whole 128-byte line
  // (a thread per (row, k octet) read 64 different lines per wave instruction: the first form of this kernel was bound by exactly that)
  const int aq = tid & 7, ar0 = tid >> 3;
  const float* arow[NLD];
  int aoff[NLD];
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int rl = ar0 + 32 * it, row = m0 + rl;
    arow[it] = a.x + (long long)(row < a.M ? row : a.M - 1) * a.xrs + aq * 4;
    aoff[it] = (rl >> 4) * 3 * FRAGB + (aq >> 1) * KOP + (rl & 15) * 16 + (aq & 1) * 8;
  }
  float4 g[KC][NLD];
  u32x4 b[KC][4], bn[KC][4];
  // (a chunk index past the end is clamped: the loads stay unconditional, the extra operands are never multiplied)
  auto load_ab = [&](int c0, u32x4 (&bd)[KC][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int c = c0 + kc < nch ? c0 + kc : nch - 1;
#pragma unroll
      for (int it = 0; it < NLD; ++it) g[kc][it] = *(const float4*)(arow[it] + 32 * c);
#pragma unroll
      for (int s = 0; s < 4; ++s) bd[kc][s] = *(bp[s] + (long long)c * bstep);
    }
  };
  auto store_a = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const float t[4] = {g[kc][it].x, g[kc][it].y, g[kc][it].z, g[kc][it].w};
        unsigned x1[2], x2[2], x3[2];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const float a0 = t[e], a1 = t[e + 1];
          const float b0 = a0 - __uint_as_float(__float_as_uint(a0) & 0xffff0000u), b1 = a1 - __uint_as_float(__float_as_uint(a1) & 0xffff0000u);
          const float c0 = b0 - __uint_as_float(__float_as_uint(b0) & 0xffff0000u), c1 = b1 - __uint_as_float(__float_as_uint(b1) & 0xffff0000u);
          x1[e >> 1] = __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);
          x2[e >> 1] = __builtin_amdgcn_perm(__float_as_uint(b1), __float_as_uint(b0), 0x07060302u);
          x3[e >> 1] = __builtin_amdgcn_perm(__float_as_uint(c1), __float_as_uint(c0), 0x07060302u);
        }
        unsigned char* d = abuf + (buf * KC + kc) * CHB + aoff[it];
        *(uint2*)d = make_uint2(x1[0], x1[1]);
        *(uint2*)(d + FRAGB) = make_uint2(x2[0], x2[1]);
        *(uint2*)(d + 2 * FRAGB) = make_uint2(x3[0], x3[1]);
      }
  };
  kk_f32x4 acc[NIT][4];
#pragma unroll
  for (int mi = 0; mi < NIT; ++mi)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[mi][s] = kk_f32x4{0.f, 0.f, 0.f, 0.f};
  load_ab(0, b);
  store_a(0);
  __syncthreads();
  const int rdoff = (lane >> 4) * KOP + (lane & 15) * 16;
  int buf = 0;
  for (int c = 0; c < nch; c += KC, buf ^= 1) {
    load_ab(c + KC, bn);  // the next step's operands: in flight under this step's matrix instructions
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if (c + kc < nch) {
#pragma unroll
        for (int mi = 0; mi < NIT; ++mi)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const kk_bf16x8 af = *(const kk_bf16x8*)(abuf + (buf * KC + kc) * CHB + ((NIT * wave + mi) * 3 + t) * FRAGB + rdoff);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[mi][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(kk_bf16x8, b[kc][s]), acc[mi][s], 0, 0, 0);
          }
      }
    }
    store_a(buf ^ 1);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[kc][s] = bn[kc][s];
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < NIT; ++mi)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = (sb0 + s) * 16 + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long row = m0 + (NIT * wave + mi) * 16 + 4 * (lane >> 4) + i;
        if (row < a.M && n < a.N) {
          float v = acc[mi][s][i];
          if (a.res) v += a.res[row * a.rrs + n];
          a.out[row * a.ors + n] = v;
        }
      }
    }
}
// one launch of the prompt GEMM: KK_CSM_GEMM = "RT,KC" picks the tile for A/B runs
int launch_gemmp(const GPArgs& g, hipStream_t st) {
  static int rt = 0, kc = 0;
  if (!rt) {
    const char* e = getenv("KK_CSM_GEMM");
    rt = 4; kc = 2;  // (measured level: prefill of 190 positions 30.7 / 31.1 / 33.9 / 36.3 ms for 4,2 / 4,1 / 8,2 / 8,1 -- the split arithmetic redone per column tile, not the tile shape, is what is left)
    if (e && e[0] && e[1] == ',' && e[2]) { rt = e[0] == '4' ? 4 : 8; kc = e[2] == '1' ? 1 : 2; }
  }
#define GP_GO(RT, KC)                                                                                                                   \
  do {                                                                                                                                   \
    const size_t lds = (size_t)2 * KC * RT * 3 * 1536;                                                                                   \
    static KKDevOnce attr;                                                                                                               \
    if (attr.first()) {                                                                                                                  \
      (void)hipFuncSetAttribute((const void*)gemmp_kernel<RT, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
      attr.done();                                                                                                                       \
    }                                                                                                                                    \
    hipLaunchKernelGGL((gemmp_kernel<RT, KC>), dim3((g.N + 63) / 64, (g.M + 16 * RT - 1) / (16 * RT)), dim3(256), lds, st, g);           \
  } while (0)
  if (rt == 4 && kc == 1) GP_GO(4, 1);
  else if (rt == 4) GP_GO(4, 2);
  else if (kc == 1) GP_GO(8, 1);
  else GP_GO(8, 2);
#undef GP_GO
  KK_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------- host
struct Packer {
  kk_csm* m;
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
  Vec vec(const std::string& name, size_t n) {
    Vec r;
    const std::vector<float>* v = get(name, n);
    if (!v) return r;
    r.n = n;
    r.off = alloc(n);
    memcpy(&m->pack[r.off], v->data(), n * 4);
    m->host.erase(name);
    return r;
  }
  // nn.Linear weights [O_i][I] stacked along the output axis -> one [I][ldw] pack; `transposed_src`: the source is [I][O] (audio_head)
  Lin linear(const std::vector<std::string>& names, const std::vector<int>& outs, int I, const float* raw = nullptr) {
    Lin l;
    int O = 0;
    for (int o : outs) O += o;
    l.Cin = I; l.Cout = O; l.ldw = rup(O, 64);
    l.off = alloc((size_t)I * l.ldw);
    int base = 0;
    for (size_t k = 0; k < outs.size(); ++k) {
      const float* src = raw;
      if (!raw) {
        const std::vector<float>* w = get(names[k], (size_t)outs[k] * I);
        if (!w) return l;
        src = w->data();
      }
      float* dst = &m->pack[l.off];
      if (raw) {  // [I][O]
        for (int i = 0; i < I; ++i)
          for (int o = 0; o < outs[k]; ++o) dst[(size_t)i * l.ldw + base + o] = src[(size_t)i * outs[k] + o];
      } else {
        for (int o = 0; o < outs[k]; ++o)
          for (int i = 0; i < I; ++i) dst[(size_t)i * l.ldw + base + o] = src[(size_t)o * I + i];
        m->host.erase(names[k]);
      }
      base += outs[k];
    }
    if (m->wdt == KK_BF16) {
      // bf16 weight mode: the matrix IS its bf16 rounding everywhere (the fp32 pack the multi-token prompt block reads holds the rounded
      // values too, so a prompt block and single-token steps multiply by identical weights); lossless for a bf16 checkpoint.  The bf16
      // copy is laid out for the fused GEMV: [Cout / CB][Cin][CB], CB = 8 * oct columns per workgroup (fused_gemv_kernel).
      float* dst = &m->pack[l.off];
      for (size_t e = 0; e < (size_t)I * l.ldw; ++e) {
        uint32_t u;
        memcpy(&u, &dst[e], 4);
        if ((u & 0x7FFFFFFFu) <= 0x7F800000u) u += 0x7FFFu + ((u >> 16) & 1u);
        u &= 0xFFFF0000u;
        memcpy(&dst[e], &u, 4);
      }
      // the round-2 / round-3a column-block pack (fused_gemv_kernel, gemv8_kernel) is built only for the A/B runs that ask for those kernels
      if (I % 64 == 0 && (getenv("KK_CSM_NO_MFMA") || getenv("KK_CSM_OLD") || I % 32 != 0)) {
        l.oct = (O >= 8192 || I >= 4096) ? 8 : 1;  // wide (gate|up) and deep (down) matrices: 64 columns per workgroup
        const int CB = 8 * l.oct, nblk = (O + CB - 1) / CB;
        l.boff = m->packb.size();
        m->packb.resize(l.boff + (size_t)nblk * I * CB, 0);
        uint16_t* db = &m->packb[l.boff];
        for (int i = 0; i < I; ++i)
          for (int o = 0; o < O; ++o) {
            uint32_t u;
            memcpy(&u, &dst[(size_t)i * l.ldw + o], 4);
            db[((size_t)(o / CB) * I + i) * CB + o % CB] = (uint16_t)(u >> 16);
          }
      }
      if (I % 32 == 0 && !getenv("KK_CSM_NO_MFMA")) {
        // fragment pack of gemvm_kernel.  Split-K for the deep projections (K >= 4096 in slices of 1024 rows), then the widest column
        // block (16 * nsub) that still gives ~200 workgroups; both depend on the matrix only, never on the batch.
        l.ks = (I >= 4096 && I % 1024 == 0) ? (I / 1024 > 16 ? 16 : I / 1024) : 1;
        while (l.ks > 1 && (I % l.ks != 0 || (I / l.ks) % 32 != 0)) --l.ks;
        const int nb16 = (O + 15) / 16;
        l.nsub = nb16 * l.ks / 4 >= 192 ? 4 : (nb16 * l.ks / 2 >= 192 ? 2 : 1);
        if (const char* e = getenv("KK_CSM_NSUB_MAX")) { const int mx = atoi(e); if (mx >= 1 && l.nsub > mx) l.nsub = mx; }  // (A/B: narrower column blocks = more workgroups; measured: 2 instead of 4 sub-blocks for gate|up costs +0.09 ms per frame)
        const int CBm = 16 * l.nsub, nblk = (O + CBm - 1) / CBm, nchunk = I / 32;
        l.moff = m->packb.size();
        m->packb.resize(l.moff + (size_t)nblk * I * CBm, 0);
        uint16_t* dm = &m->packb[l.moff];
        for (int i = 0; i < I; ++i) {
          const int c = i >> 5, kq = (i & 31) >> 3, j = i & 7;
          for (int o = 0; o < O; ++o) {
            uint32_t u;
            memcpy(&u, &dst[(size_t)i * l.ldw + o], 4);
            const int nbk = o / CBm, sb = (o % CBm) >> 4, L = kq * 16 + (o & 15);
            dm[((((size_t)nbk * nchunk + c) * l.nsub + sb) * 64 + L) * 8 + j] = (uint16_t)(u >> 16);
          }
        }
      }
    }
    return l;
  }
};

void llama3_theta(const kk_llama_args& a, std::vector<float>& th) {  // attention.py:33-82, float32 like the reference
  const int half = a.head_dim / 2;
  th.resize(half);
  const double low_w = 8192.0 / 1.0, high_w = 8192.0 / 4.0;
  for (int i = 0; i < half; ++i) {
    const float f = 1.0f / powf(a.rope_theta, (float)(2 * i) / (float)a.head_dim);
    const double wl = 2.0 * M_PI / (double)f;
    double v;
    if (wl < high_w) v = f;
    else if (wl > low_w) v = (double)f / a.rope_factor;
    else {
      const double smooth = (8192.0 / wl - 1.0) / (4.0 - 1.0);
      v = (1.0 - smooth) * (double)f / a.rope_factor + smooth * (double)f;
    }
    th[i] = (float)v;
  }
}

void pack_stack(Packer& P, const std::string& name, Stack& st, int max_pos) {
  const kk_llama_args& a = st.a;
  const int H = a.num_heads, KV = a.num_kv_heads, hd = a.head_dim, D = a.hidden, I = a.intermediate;
  st.layers.resize(a.num_layers);
  for (int i = 0; i < a.num_layers; ++i) {
    const std::string p = name + ".layers." + std::to_string(i);
    LlamaLayer& L = st.layers[i];
    L.n1 = P.vec(p + ".input_layernorm.weight", D);
    L.n2 = P.vec(p + ".post_attention_layernorm.weight", D);
    L.qkv = P.linear({p + ".self_attn.q_proj.weight", p + ".self_attn.k_proj.weight", p + ".self_attn.v_proj.weight"}, {H * hd, KV * hd, KV * hd}, D);
    L.o = P.linear({p + ".self_attn.o_proj.weight"}, {D}, H * hd);
    L.gu = P.linear({p + ".mlp.gate_proj.weight", p + ".mlp.up_proj.weight"}, {I, I}, D);
    L.down = P.linear({p + ".mlp.down_proj.weight"}, {D}, I);
  }
  st.norm = P.vec(name + ".norm.weight", D);
  std::vector<float> th;
  llama3_theta(a, th);
  st.max_pos = max_pos;
  st.rope.n = (size_t)max_pos * (hd / 2) * 2;
  st.rope.off = P.alloc(st.rope.n);
  float* r = &P.m->pack[st.rope.off];
  for (int pos = 0; pos < max_pos; ++pos)
    for (int i = 0; i < hd / 2; ++i) {
      const float ang = (float)pos * th[i];  // einsum in float32 (attention.py:56-58)
      r[((size_t)pos * (hd / 2) + i) * 2] = cosf(ang);
      r[((size_t)pos * (hd / 2) + i) * 2 + 1] = sinf(ang);
    }
}

void resolve(kk_csm* m, Lin& l) {
  l.w = m->dev + l.off;
  l.wb = (m->devb && l.oct) ? m->devb + l.boff : nullptr;
  l.wm = (m->devb && l.nsub) ? m->devb + l.moff : nullptr;
}
void resolve(kk_csm* m, Vec& v) { v.p = v.n ? m->dev + v.off : nullptr; }
void resolve(kk_csm* m, Stack& st) {
  for (auto& L : st.layers) { resolve(m, L.qkv); resolve(m, L.o); resolve(m, L.gu); resolve(m, L.down); resolve(m, L.n1); resolve(m, L.n2); }
  resolve(m, st.norm); resolve(m, st.rope);
}

struct Run {
  kk_csm* m;
  hipStream_t st;
  int B;
  char* base;
  size_t cap, used;
  bool dry, oom;
  float* skinny_scratch = nullptr;  // partial sums of the skinny GEMM
  size_t skinny_floats = 0;
  float* f32(size_t n) {
    const size_t off = (used + 255) & ~(size_t)255;
    used = off + n * 4;
    if (dry) return nullptr;
    if (used > cap) { oom = true; return nullptr; }
    return (float*)(base + off);
  }
  // out[b][row][:] = W x[b][row][:] (+ res); x rows: `rows` per item at pitch `xbs` elements between items
  // `gated`: x is [.. rows][2 * Cin] = gate | up and the input of the product is silu(gate) * up (skinny bf16 path only; callers check
  // can_gate() first and run the stand-alone swiglu kernel otherwise).  `nw` / `xn`: RMSNorm of the result rows, launched right behind.
  bool can_gate(const Lin& w, int rows) const { (void)w; (void)rows; return false; }  // (the gated form lives in the fused GEMV now)
  int lin(const Lin& w, const float* x, long long xbs, int rows, float* out, long long obs, const float* res, bool gated = false,
          const float* nw = nullptr, float* xn = nullptr, float eps = 0.f) {
    if (dry) return 0;
    if (gated && !(can_gate(w, rows) && xbs == (long long)rows * 2 * w.Cin && obs == (long long)rows * w.Cout)) return kk_fail("kk_csm: internal: gated input");
    if (gated) xbs = (long long)rows * w.Cin;
    KKConvArgs a;
    memset(&a, 0, sizeof a);
    a.x = x; a.xbs = xbs; a.ldx = w.Cin; a.w = w.w; a.ldw = w.ldw;
    a.out = out; a.obs = obs; a.ldo = w.Cout;
    if (res) { a.res = res; a.rbs = obs; a.ldr = w.Cout; }
    a.Cin = w.Cin; a.Cout = w.Cout; a.Kw = 1; a.mode = KK_CONV; a.stride = 1; a.dil = 1;
    a.Q = rows; a.Lo_rows = rows; a.lin = KKLen{nullptr, 0, rows}; a.lout = KKLen{nullptr, 0, rows};
    a.in_slope = 1.f; a.scale = 1.f;
    int nb = B;
    if (xbs == (long long)rows * w.Cin && obs == (long long)rows * w.Cout && rows <= 2 && skinny_scratch) {
      // single-token steps (and the decoder's 2-token first step): the HBM-bound skinny GEMM (every CU streams a slice of W once for up
      // to 16 rows).  The choice depends on the rows PER ITEM only, never on B, so a stream's bits do not depend on its batch.
      const int Mtot = B * rows, nblk256 = kk_cdiv(w.Cout, 256), nblk = nblk256;
      int KS = 1024 / nblk256;  // ~4 workgroups per CU (measured: fewer, longer slices are slower -- the kernel is latency-bound)
      int maxks = kk_cdiv(w.Cin, 32);
      if (maxks > 128) maxks = 128;  // deep, narrow matrices (down projections: K = 8192, N = 1024 / 2048) need the slices to fill the chip
      KS = KS < 1 ? 1 : (KS > maxks ? maxks : KS);
      const int kchunk = kk_cdiv(kk_cdiv(w.Cin, KS), 32) * 32;
      KS = kk_cdiv(w.Cin, kchunk);
      if ((size_t)KS * SK_MAXM * w.Cout <= skinny_floats) {
        for (int m0 = 0; m0 < Mtot; m0 += SK_MAXM) {
          const int M = Mtot - m0 < SK_MAXM ? Mtot - m0 : SK_MAXM;
          const float* xin = x + (size_t)m0 * (gated ? 2 : 1) * w.Cin;
          const dim3 g(nblk, KS), t(256);
#define SK_GO(MT) hipLaunchKernelGGL((skinny_gemm_kernel<MT, false, false>), g, t, 0, st, xin, M, w.Cin, (const void*)w.w, w.ldw, w.Cout, kchunk, skinny_scratch)
          // MT depends on the rows per launch only through "fits in 8": a row's arithmetic is the same in both instantiations
          if (M <= 8) SK_GO(8); else SK_GO(16);
#undef SK_GO
          KK_CHECK_LAUNCH();
          const float* resp = res ? res + (size_t)m0 * w.Cout : nullptr;
          float* outp = out + (size_t)m0 * w.Cout;
          hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)(((long long)M * w.Cout + 255) / 256)), dim3(256), 0, st, skinny_scratch, KS, M, w.Cout, resp, outp);
          KK_CHECK_LAUNCH();
        }
        if (xn) {  // (one workgroup per row summing the slices AND normalising was tried: 17-38 us against 5 + 5 for the two launches)
          hipLaunchKernelGGL(rmsnorm_kernel, dim3(Mtot), dim3(256), 0, st, out, nw, w.Cout, eps, xn);
          KK_CHECK_LAUNCH();
        }
        return 0;
      }
    }
    if (xbs == (long long)rows * w.Cin && obs == (long long)rows * w.Cout && w.wm && w.nsub && !gated && rows > 2 && !(ab_switches() & 1)) {
      // the prompt block in bf16 weight mode: matrix cores (gemmp_kernel); KK_CSM_PROMPT_F32=1 keeps the round-2 generic fp32 kernel (A/B).
      // The choice depends on the rows PER ITEM only, never on B: a stream's bits do not depend on its batch.
      GPArgs g;
      memset(&g, 0, sizeof g);
      g.x = x; g.xrs = w.Cin; g.w = w.wm; g.K = w.Cin; g.N = w.Cout; g.M = B * rows; g.nsub = w.nsub;
      g.res = res; g.rrs = w.Cout; g.out = out; g.ors = w.Cout;
      { const int rc_ = launch_gemmp(g, st); if (rc_ != 0) return rc_; }
      if (xn) {
        hipLaunchKernelGGL(rmsnorm_kernel, dim3(B * rows), dim3(256), 0, st, out, nw, w.Cout, eps, xn);
        KK_CHECK_LAUNCH();
      }
      return 0;
    }
    if (xbs == (long long)rows * w.Cin && obs == (long long)rows * w.Cout) {
      // items are contiguous: one launch over B*rows rows, so a weight tile is read once for the whole batch (single-token steps would
      // otherwise re-read every matrix once per item)
      a.Q = a.Lo_rows = B * rows;
      a.lin = a.lout = KKLen{nullptr, 0, B * rows};
      a.xbs = a.obs = a.rbs = 0;
      nb = 1;
    }
    const int rc = kk_launch_conv_generic(a, nb, KK_F32, KK_F32, st);
    if (rc != 0 || !xn) return rc;
    if (obs != (long long)rows * w.Cout) return kk_fail("kk_csm: internal: norm of a strided result");
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(B * rows), dim3(256), 0, st, out, nw, w.Cout, eps, xn);
    KK_CHECK_LAUNCH();
    return 0;
  }
};

// fused GEMV launcher: `a` carries everything but the weights; rows in chunks of 16 (PRO 3: `rows` rows per item, chunk = whole items)
// split-K slices of a deep projection (K >= 4096 on 64-column blocks): enough workgroups to fill the chip; depends on the matrix only
int gemv_slices(const Lin& w) {
  if (w.nsub) return w.ks;
  if (w.oct != 8 || w.Cin < 4096) return 1;
  const int nblk = (w.Cout + 63) / 64;
  int ks = 256 / nblk;
  if (ks > 16) ks = 16;
  while (ks > 1 && (w.Cin % ks != 0 || (w.Cin / ks) % 32 != 0)) --ks;
  return ks < 1 ? 1 : ks;
}
// the matrix-core GEMV: one launch for all rows (grid z = 8-row chunks); KS must be the pack's w.ks
int launch_gemvm(const Lin& w, int pro, int epi, FGArgs a, int Mtot, hipStream_t st) {
  a.w = w.wm; a.K = w.Cin; a.N = w.Cout; a.M = Mtot; a.dbg = w.cached; a.kper = w.Cin / w.ks;
  a.ts = ts_slot(); a.ts_id = (w.Cout << 4) | (pro << 2) | epi;
  const int CB = 16 * w.nsub, nblk = (w.Cout + CB - 1) / CB, kper = w.Cin / w.ks;
  const size_t lds = gm_lds_bytes(w.nsub, kper);
  if (lds > 160 * 1024 || kper / 32 > 96) return kk_fail("kk_csm: internal: matrix-core GEMV: K slice too long for LDS");
  if (pro == 3 && 8 % a.rows != 0) return kk_fail("kk_csm: internal: matrix-core GEMV: item rows");
  const dim3 grid(nblk, w.ks, (Mtot + 7) / 8);
  const int rounds = (kper / 32 + 31) / 32;  // straight-line rounds of 4 chunks x 8 waves (K slice <= 1024: one)
#define GM_GO1(NSUB, PRO, EPI, RD)                                                                                                       \
  do {                                                                                                                                   \
    static KKDevOnce attr;                                                                                                               \
    if (attr.first()) {                                                                                                                  \
      (void)hipFuncSetAttribute((const void*)gemvm_kernel<NSUB, PRO, EPI, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
      attr.done();                                                                                                                       \
    }                                                                                                                                    \
    hipLaunchKernelGGL((gemvm_kernel<NSUB, PRO, EPI, RD>), grid, dim3(512), lds, st, a);                                                 \
  } while (0)
#define GM_GO(NSUB, PRO, EPI)                                              \
  do {                                                                      \
    if (rounds == 1) GM_GO1(NSUB, PRO, EPI, 1);                             \
    else if (rounds == 2) GM_GO1(NSUB, PRO, EPI, 2);                        \
    else GM_GO1(NSUB, PRO, EPI, 3);                                         \
  } while (0)
#define GM_PE(NSUB)                                                         \
  do {                                                                       \
    if (pro == 1 && epi == 0) GM_GO(NSUB, 1, 0);                             \
    else if (pro == 0 && epi == 1) GM_GO(NSUB, 0, 1);                        \
    else if (pro == 2 && epi == 1) GM_GO(NSUB, 2, 1);                        \
    else if (pro == 2 && epi == 2) GM_GO(NSUB, 2, 2);                        \
    else if (pro == 3 && epi == 0) GM_GO(NSUB, 3, 0);                        \
    else if (pro == 0 && epi == 0) GM_GO(NSUB, 0, 0);                        \
    else return kk_fail("kk_csm: internal: fused GEMV form");                \
  } while (0)
  if (w.nsub == 4) GM_PE(4); else if (w.nsub == 2) GM_PE(2); else GM_PE(1);
#undef GM_PE
#undef GM_GO
#undef GM_GO1
  KK_CHECK_LAUNCH();
  return 0;
}

int launch_gemv(const Lin& w, int pro, int epi, FGArgs a, int Mtot, hipStream_t st, int KS = 1) {
  if (w.wm && w.nsub) {
    if (KS != w.ks || (epi == 2) != (w.ks > 1)) return kk_fail("kk_csm: internal: split-K form");
    return launch_gemvm(w, pro, epi, a, Mtot, st);
  }
  if (!w.wb || !w.oct) return kk_fail("kk_csm: internal: fused GEMV without a block pack");
  a.w = w.wb; a.K = w.Cin; a.N = w.Cout;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("KK_CSM_DBG"); dbg = e ? atoi(e) : 0; }
    a.dbg = dbg;
  }
  const int CB = 8 * w.oct, nblk = (w.Cout + CB - 1) / CB;
  if ((epi == 2) != (KS > 1 || epi == 2)) return kk_fail("kk_csm: internal: split-K form");
  for (int m0 = 0; m0 < Mtot; m0 += 16) {
    FGArgs g = a;
    g.M = Mtot - m0 < 16 ? Mtot - m0 : 16;
    if (pro == 3) {
      const int item0 = m0 / a.rows;
      g.x = a.x + (long long)item0 * a.xrs;
      g.codes = a.codes + (long long)item0 * a.cstride;
    } else {
      g.x = a.x + (long long)m0 * a.xrs;
    }
    if (a.res) g.res = a.res + (long long)m0 * a.rrs;
    g.out = a.out + (long long)m0 * a.ors;
#define FG_GO1(MT, OCT, PRO, EPI, NPRE)                                                                                                 \
  do {                                                                                                                                   \
    static KKDevOnce attr;                                                                                                               \
    const size_t lds_ = (fg_lds_bytes<MT, OCT>());                                                                                       \
    if (attr.first()) {                                                                                                                  \
      (void)hipFuncSetAttribute((const void*)fused_gemv_kernel<MT, OCT, PRO, EPI, NPRE>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                (int)lds_);                                                                                              \
      attr.done();                                                                                                                       \
    }                                                                                                                                    \
    hipLaunchKernelGGL((fused_gemv_kernel<MT, OCT, PRO, EPI, NPRE>), dim3(nblk, KS), dim3(256), lds_, st, g);                               \
  } while (0)
#define FG_GO(MT, OCT, PRO, EPI) FG_GO1(MT, OCT, PRO, EPI, (OCT == 8 && MT == 8 ? 16 : 8))  /* (16 rows x 16 in flight would spill) */
#define FG_PE(MT, OCT)                                                      \
  do {                                                                       \
    if (pro == 1 && epi == 0) FG_GO(MT, OCT, 1, 0);                          \
    else if (pro == 0 && epi == 1) FG_GO(MT, OCT, 0, 1);                     \
    else if (pro == 2 && epi == 1) FG_GO(MT, OCT, 2, 1);                     \
    else if (pro == 2 && epi == 2) FG_GO(MT, OCT, 2, 2);                     \
    else if (pro == 3 && epi == 0) FG_GO(MT, OCT, 3, 0);                     \
    else if (pro == 0 && epi == 0) FG_GO(MT, OCT, 0, 0);                     \
    else return kk_fail("kk_csm: internal: fused GEMV form");                \
  } while (0)
    // M <= 8 rows and one K chunk: the compact kernel (gemv8_kernel); KK_CSM_OLD=1 keeps the round-2 kernel for A/B timing
    static int old_kernel = -1, g8_mask = 0xFF;
    if (old_kernel < 0) {
      old_kernel = getenv("KK_CSM_OLD") ? 1 : 0;
      if (getenv("KK_CSM_G8_MASK")) g8_mask = atoi(getenv("KK_CSM_G8_MASK"));  // (debugging: bit p = prologue p on the compact kernel; bit 4 + oct/8)
    }
    const int kper = a.K / KS;
    if (!old_kernel && ((g8_mask >> pro) & 1) && ((g8_mask >> (4 + (w.oct == 8 ? 1 : 0))) & 1) && g.M <= 8 && kper <= 2048 && kper % 128 == 0) {
#define G8_GO(OCT, PRO, EPI)                                                                                                             \
  do {                                                                                                                                   \
    static KKDevOnce attr;                                                                                                               \
    const size_t lds_ = (g8_lds_bytes<OCT>());                                                                                           \
    if (attr.first()) {                                                                                                                  \
      (void)hipFuncSetAttribute((const void*)gemv8_kernel<OCT, PRO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);        \
      attr.done();                                                                                                                       \
    }                                                                                                                                    \
    hipLaunchKernelGGL((gemv8_kernel<OCT, PRO, EPI>), dim3(nblk, KS), dim3(256), lds_, st, g);                                           \
  } while (0)
#define G8_PE(OCT)                                                          \
  do {                                                                       \
    if (pro == 1 && epi == 0) G8_GO(OCT, 1, 0);                              \
    else if (pro == 0 && epi == 1) G8_GO(OCT, 0, 1);                         \
    else if (pro == 2 && epi == 1) G8_GO(OCT, 2, 1);                         \
    else if (pro == 2 && epi == 2) G8_GO(OCT, 2, 2);                         \
    else if (pro == 3 && epi == 0) G8_GO(OCT, 3, 0);                         \
    else if (pro == 0 && epi == 0) G8_GO(OCT, 0, 0);                         \
    else return kk_fail("kk_csm: internal: fused GEMV form");                \
  } while (0)
      if (w.oct == 8) G8_PE(8); else G8_PE(1);
#undef G8_PE
#undef G8_GO
      KK_CHECK_LAUNCH();
      continue;
    }
    // MT depends on the rows per launch only through "fits in 8": a row's arithmetic is the same in both instantiations
    if (g.M <= 8) { if (w.oct == 8) FG_PE(8, 8); else FG_PE(8, 1); }
    else { if (w.oct == 8) FG_PE(16, 8); else FG_PE(16, 1); }
#undef FG_PE
#undef FG_GO
#undef FG_GO1
    KK_CHECK_LAUNCH();
  }
  return 0;
}

#define CS_TRY(x)          \
  do {                     \
    const int rc__ = (x);  \
    if (rc__ != 0) return rc__; \
  } while (0)

// h [B][S][D] (updated in place) -> out [B][S][D] = final norm; appends S positions to the stack's cache at st.offset
int stack_forward(Run& r, Stack& st, float* h, int S, int offset, float* out) {
  const kk_llama_args& a = st.a;
  const int B = r.B, H = a.num_heads, KV = a.num_kv_heads, hd = a.head_dim, D = a.hidden, I = a.intermediate;
  const int W = (H + 2 * KV) * hd;
  float* x = r.f32((size_t)B * S * D);
  float* qkv = r.f32((size_t)B * S * W);
  float* att = r.f32((size_t)B * S * H * hd);
  float* gu = r.f32((size_t)B * S * 2 * I);
  float* act = r.f32((size_t)B * S * I);
  if (r.oom) return kk_fail("kk_csm: workspace too small");
  if (!r.dry && offset + S > st.max_pos) return kk_fail("kk_csm: sequence exceeds the cache (max_seq_len)");
  // x = RMSNorm(h) of the CURRENT layer's input: stand-alone for layer 0, afterwards produced by the previous down projection's tail
  if (!r.dry) {
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(B * S), dim3(256), 0, r.st, h, st.layers[0].n1.p, D, a.rms_eps, x);
    KK_CHECK_LAUNCH();
  }
  for (int l = 0; l < a.num_layers; ++l) {
    const LlamaLayer& L = st.layers[l];
    float* kc = st.kc + (size_t)l * r.m->max_batch * st.max_pos * KV * hd;
    float* vc = st.vc + (size_t)l * r.m->max_batch * st.max_pos * KV * hd;
    CS_TRY(r.lin(L.qkv, x, (long long)S * D, S, qkv, (long long)S * W, nullptr));
    if (!r.dry) {
      if (S == 1) {  // single-token step: RoPE + cache append inside the attention kernel
        hipLaunchKernelGGL(attn_cache_kernel<true>, dim3(S, H, B), dim3(128), attn_lds_bytes(st.max_pos, hd), r.st, qkv, S, H, KV, hd, st.pos_dev,
                           st.pos_dev ? 0 : offset, kc, vc, st.max_pos, 1.0f / sqrtf((float)hd), att, 1, -1, st.rope.p, st.pad_dev);
        KK_CHECK_LAUNCH();
      } else {
        hipLaunchKernelGGL(rope_append_kernel, dim3(S, B), dim3(256), 0, r.st, qkv, S, H, KV, hd, st.rope.p, st.pos_dev, st.pos_dev ? 0 : offset, kc, vc, st.max_pos, st.pad_dev);
        KK_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_cache_kernel<false>, dim3(S, H, B), dim3(128), attn_lds_bytes(st.max_pos, hd), r.st, qkv, S, H, KV, hd, st.pos_dev,
                           st.pos_dev ? 0 : offset, kc, vc, st.max_pos, 1.0f / sqrtf((float)hd), att, 1, -1, (const float*)nullptr, st.pad_dev);
        KK_CHECK_LAUNCH();
      }
    }
    // h += o(att); x = RMSNorm(h) (post_attention_layernorm)
    CS_TRY(r.lin(L.o, att, (long long)S * H * hd, S, h, (long long)S * D, h, false, L.n2.p, x, a.rms_eps));
    CS_TRY(r.lin(L.gu, x, (long long)S * D, S, gu, (long long)S * 2 * I, nullptr));
    // h += down(silu(gate) * up); then the NEXT consumer's norm: the next layer's input_layernorm -> x, or the stack's final norm -> out
    const bool lastl = l + 1 == a.num_layers;
    const float* nw = lastl ? st.norm.p : st.layers[l + 1].n1.p;
    float* xn = lastl ? out : x;
    if (r.can_gate(L.down, S)) {  // bf16 weight mode, single-token step: SwiGLU is applied while the down projection stages its input
      CS_TRY(r.lin(L.down, gu, (long long)S * 2 * I, S, h, (long long)S * D, h, true, nw, xn, a.rms_eps));
    } else {
      if (!r.dry) {
        const long long n = (long long)B * S * I;
        hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, r.st, gu, I, n, act);
        KK_CHECK_LAUNCH();
      }
      CS_TRY(r.lin(L.down, act, (long long)S * I, S, h, (long long)S * D, h, false, nw, xn, a.rms_eps));
    }
  }
  return 0;
}

// The single-token step of a Llama stack in bf16 weight mode: FIVE launches per layer, no split-K partials, no stand-alone reduce / norm /
// SwiGLU kernels -- qkv = W(rms(h) n1); attention (RoPE + cache append inside for one-row items); h += Wo att; gu = W(rms(h) n2);
// h += Wdown(silu(gate) up).  h [B][rows][D] is updated in place; the consumer applies the final norm (a head's prologue).
bool stack_can_step(const Stack& st) {
  for (const auto& L : st.layers)
    for (const Lin* w : {&L.qkv, &L.o, &L.gu, &L.down})
      if (!(w->wm || w->wb)) return false;
  return !st.layers.empty();
}
// `gather` (optional): layer 0 reads its input rows from a table by code instead of from h, and writes them to h (FGArgs codes / cstride / cb / V / emb)
int stack_step(Run& r, Stack& st, float* h, int rows, int offset, const FGArgs* gather = nullptr) {
  const kk_llama_args& a = st.a;
  const int B = r.B, H = a.num_heads, KV = a.num_kv_heads, hd = a.head_dim, D = a.hidden, I = a.intermediate;
  const int W = (H + 2 * KV) * hd, M = B * rows;
  float* qkv = r.f32((size_t)M * W);
  float* att = r.f32((size_t)M * H * hd);
  float* gu = r.f32((size_t)M * 2 * I);
  float* part = r.f32((size_t)16 * M * D);  // partial tiles of a split-K down projection (<= 16 slices)
  float* attp = r.f32((size_t)8 * M * H * (hd + 2));  // key-split partials of the long-cache attention (<= 8 splits)
  if (r.oom) return kk_fail("kk_csm: workspace too small");
  if (r.dry) return 0;
  if (offset + rows > st.max_pos) return kk_fail("kk_csm: sequence exceeds the cache (max_seq_len)");
  for (int l = 0; l < a.num_layers; ++l) {
    const LlamaLayer& L = st.layers[l];
    float* kc = st.kc + (size_t)l * r.m->max_batch * st.max_pos * KV * hd;
    float* vc = st.vc + (size_t)l * r.m->max_batch * st.max_pos * KV * hd;
    FGArgs g;
    memset(&g, 0, sizeof g);
    g.x = h; g.xrs = D; g.nw = L.n1.p; g.eps = a.rms_eps; g.out = qkv; g.ors = W;
    if (l == 0 && gather) { g.codes = gather->codes; g.cstride = gather->cstride; g.cb = gather->cb; g.V = gather->V; g.emb = gather->emb; g.gather_out = h; }
    if (!(g_skip & 1)) CS_TRY(launch_gemv(L.qkv, 1, 0, g, M, r.st));
    if (g_skip & 2) {
    } else if (rows == 1 && st.max_pos <= 64 && H / KV <= 8 && !(ab_switches() & 2)) {
      const size_t lds = attn_step_lds_bytes(st.max_pos, hd, H / KV);
      static KKDevOnce attr;
      if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)attn_step_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_step_lds_bytes(64, 128, 8));
        (void)hipFuncSetAttribute((const void*)attn_step_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_step_lds_bytes(64, 64, 8));
        attr.done();
      }
      if (hd == 128)
        hipLaunchKernelGGL(attn_step_kernel<128>, dim3(KV, B), dim3(256), lds, r.st, qkv, H, KV, st.pos_dev, st.pos_dev ? 0 : offset, kc, vc, st.max_pos,
                           1.0f / sqrtf((float)hd), att, st.rope.p, st.pad_dev, ts_slot());
      else
        hipLaunchKernelGGL(attn_step_kernel<64>, dim3(KV, B), dim3(256), lds, r.st, qkv, H, KV, st.pos_dev, st.pos_dev ? 0 : offset, kc, vc, st.max_pos,
                           1.0f / sqrtf((float)hd), att, st.rope.p, st.pad_dev, ts_slot());
      KK_CHECK_LAUNCH();
    } else if (rows == 1 && H / KV <= 8 && !(ab_switches() & 2)) {
      const size_t lds = attn_decode_lds_bytes(hd, H / KV);
      static KKDevOnce attr;
      if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)attn_decode_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_decode_lds_bytes(128, 8));
        (void)hipFuncSetAttribute((const void*)attn_decode_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_decode_lds_bytes(64, 8));
        attr.done();
      }
      // key splits: one per 2 chunks of the longest cache, at most 8 (the backbone's 2048 positions: 8 splits of 2 chunks; short test stacks: 1-2)
      const int CHk = 8192 / hd, nchunks = (st.max_pos + CHk - 1) / CHk;
      int nsplit = (nchunks + 1) / 2;
      nsplit = nsplit < 1 ? 1 : (nsplit > 8 ? 8 : nsplit);
      if (nsplit > 1 && !attp) return kk_fail("kk_csm: internal: attention partials");
      if (hd == 128)
        hipLaunchKernelGGL(attn_decode_kernel<128>, dim3(KV, B, nsplit), dim3(256), lds, r.st, qkv, H, KV, st.pos_dev, st.pos_dev ? 0 : offset, kc, vc, st.max_pos,
                           1.0f / sqrtf((float)hd), att, st.rope.p, st.pad_dev, nsplit, attp);
      else
        hipLaunchKernelGGL(attn_decode_kernel<64>, dim3(KV, B, nsplit), dim3(256), lds, r.st, qkv, H, KV, st.pos_dev, st.pos_dev ? 0 : offset, kc, vc, st.max_pos,
                           1.0f / sqrtf((float)hd), att, st.rope.p, st.pad_dev, nsplit, attp);
      KK_CHECK_LAUNCH();
      if (nsplit > 1) {
        if (hd == 128) hipLaunchKernelGGL(attn_merge_kernel<128>, dim3(H, B), dim3(128), 0, r.st, attp, H, H / KV, nsplit, att);
        else hipLaunchKernelGGL(attn_merge_kernel<64>, dim3(H, B), dim3(64), 0, r.st, attp, H, H / KV, nsplit, att);
      }
      KK_CHECK_LAUNCH();
    } else if (rows == 1) {
      hipLaunchKernelGGL(attn_cache_kernel<true>, dim3(1, H, B), dim3(128), attn_lds_bytes(st.max_pos, hd), r.st, qkv, 1, H, KV, hd, st.pos_dev,
                         st.pos_dev ? 0 : offset, kc, vc, st.max_pos, 1.0f / sqrtf((float)hd), att, 1, -1, st.rope.p, st.pad_dev);
      KK_CHECK_LAUNCH();
    } else {
      hipLaunchKernelGGL(rope_append_kernel, dim3(rows, B), dim3(256), 0, r.st, qkv, rows, H, KV, hd, st.rope.p, st.pos_dev, st.pos_dev ? 0 : offset, kc, vc,
                         st.max_pos, st.pad_dev);
      KK_CHECK_LAUNCH();
      hipLaunchKernelGGL(attn_cache_kernel<false>, dim3(rows, H, B), dim3(128), attn_lds_bytes(st.max_pos, hd), r.st, qkv, rows, H, KV, hd, st.pos_dev,
                         st.pos_dev ? 0 : offset, kc, vc, st.max_pos, 1.0f / sqrtf((float)hd), att, 1, -1, (const float*)nullptr, st.pad_dev);
      KK_CHECK_LAUNCH();
    }
    memset(&g, 0, sizeof g);
    g.x = att; g.xrs = (long long)H * hd; g.res = h; g.rrs = D; g.out = h; g.ors = D;
    if (!(g_skip & 4)) CS_TRY(launch_gemv(L.o, 0, 1, g, M, r.st));
    memset(&g, 0, sizeof g);
    g.x = h; g.xrs = D; g.nw = L.n2.p; g.eps = a.rms_eps; g.out = gu; g.ors = 2 * I;
    if (!(g_skip & 8)) CS_TRY(launch_gemv(L.gu, 1, 0, g, M, r.st));
    memset(&g, 0, sizeof g);
    const int KS = gemv_slices(L.down);
    if (KS > 1) {  // deep projection: K slices over workgroups, then one small combine (h += sum of the slices)
      g.x = gu; g.xrs = 2 * I; g.out = part; g.ors = D; g.pss = (long long)M * D;
      if (!(g_skip & 16)) CS_TRY(launch_gemv(L.down, 2, 2, g, M, r.st, KS));
      const long long n = (long long)M * D;
      if (!(g_skip & 32)) {
        hipLaunchKernelGGL(combine_slices_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, r.st, part, KS, n, n, h, ts_slot());
        KK_CHECK_LAUNCH();
      }
    } else {
      g.x = gu; g.xrs = 2 * I; g.res = h; g.rrs = D; g.out = h; g.ors = D;
      if (!(g_skip & 16)) CS_TRY(launch_gemv(L.down, 2, 1, g, M, r.st));
    }
  }
  return 0;
}

int run_frame(Run& r, int S, const int* tokens, const float* mask, float temp, int top_k, const float* uniforms, int* codes) {
  kk_csm* m = r.m;
  const kk_csm_config& c = m->cfg;
  const int B = r.B, ncb = c.audio_num_codebooks, V = c.audio_vocab_size, D = c.backbone.hidden, Dd = c.decoder.hidden;
  const size_t mark = r.used;
  float* h = r.f32((size_t)B * S * D);
  float* hn = r.f32((size_t)B * S * D);
  float* logits = r.f32((size_t)B * V);
  float* curr = r.f32((size_t)B * 2 * D);
  float* pin = r.f32((size_t)B * 2 * Dd);
  float* dn = r.f32((size_t)B * 2 * Dd);
  {  // skinny-GEMM partials: at most 1024 workgroups of 256 columns -> KS * N <= 1024 * 256 (+ slack for the rounding of the K slices)
    r.skinny_floats = (size_t)2 * 1024 * 256 * SK_MAXM;
    r.skinny_scratch = r.f32(r.skinny_floats);
  }
  if (r.oom) return kk_fail("kk_csm_generate_frame: workspace too small");
  if (!r.dry) {
    if (ncb > 71) return kk_fail("kk_csm: more than 71 code books");
    hipLaunchKernelGGL(embed_sum_kernel, dim3(B * S, (D + 255) / 256), dim3(256), 0, r.st, tokens, mask, m->audio_emb.p, m->text_emb.p, ncb, V, c.text_vocab_size, D, h);
    KK_CHECK_LAUNCH();
  }
  const size_t inner = r.used;
  // bf16 weight mode: single-token frames (and every depth-decoder step) run on the fused five-launch layers
  const bool fast = m->wdt == KK_BF16 && stack_can_step(m->bb) && stack_can_step(m->dec) && (m->proj.wm || m->proj.wb) && (m->c0_head.wm || m->c0_head.wb);
  bool heads_fast = fast;
  for (const auto& l : m->audio_head) heads_fast = heads_fast && (l.wm || l.wb);
  const float* last_h;   // the backbone's final-normed last position of every item
  long long last_rs;     // its item pitch
  if (fast && heads_fast && S == 1) {
    CS_TRY(stack_step(r, m->bb, h, 1, m->bb.offset));
    if (!r.dry) {
      hipLaunchKernelGGL(rmsnorm_kernel, dim3(B), dim3(256), 0, r.st, h, m->bb.norm.p, D, c.backbone.rms_eps, hn);
      KK_CHECK_LAUNCH();
    }
    last_h = hn; last_rs = D;
  } else {
    CS_TRY(stack_forward(r, m->bb, h, S, m->bb.offset, hn));
    last_h = hn ? hn + (size_t)(S - 1) * D : nullptr;  // row S-1 of every item (pitch S*D)
    last_rs = (long long)S * D;
  }
  size_t peak = r.used;
  r.used = inner;  // the stack's scratch is free again
  // the heads write their logits straight into the slot kk_csm_debug_logits reads ([n_cb][maxB][V], first B rows): no copy per code book
  if (!r.dry && m->dbg_logits) logits = m->dbg_logits;
  if (fast && heads_fast) {
    if (!r.dry) {
      FGArgs g;
      memset(&g, 0, sizeof g);
      g.x = last_h; g.xrs = last_rs; g.out = logits; g.ors = V;
      if (!(g_skip & 64)) CS_TRY(launch_gemv(m->c0_head, 0, 0, g, B, r.st));
      if (!(g_skip & 128)) CS_TRY(launch_sample(logits, V, temp, top_k, uniforms, ncb, codes, ncb, B, r.st));
    }
    int rows = 2, dpos = 0;
    for (int i = 1; i < ncb; ++i) {
      r.used = inner;
      // curr = [last_h, embed(0, c0)] for the first step, [embed(i-1, c_{i-1})] afterwards (sesame.py:373-392): gathered by the projection's prologue
      // later steps (one row per item): the projection of an embedding row is a row of the table built at finalize -- no launch; the decoder's first
      // kernel gathers it and materialises the residual stream
      const bool tabled = rows == 1 && m->proj_table && m->dec.layers[0].qkv.wm && !(ab_switches() & 4);
      FGArgs gat;
      memset(&gat, 0, sizeof gat);
      gat.codes = codes + (i - 1); gat.cstride = ncb; gat.cb = i - 1; gat.V = V; gat.emb = m->proj_table;
      if (!r.dry && !tabled) {
        FGArgs g;
        memset(&g, 0, sizeof g);
        g.x = last_h; g.xrs = last_rs; g.codes = codes + (i - 1); g.cstride = ncb; g.cb = i - 1; g.V = V; g.rows = rows; g.emb = m->audio_emb.p;
        g.out = pin; g.ors = Dd;
        if (!(g_skip & 256)) CS_TRY(launch_gemv(m->proj, 3, 0, g, B * rows, r.st));
      }
      CS_TRY(stack_step(r, m->dec, pin, rows, dpos, tabled ? &gat : nullptr));
      if (r.used > peak) peak = r.used;
      dpos += rows;
      if (!r.dry) {
        if (m->dbg_logits) logits = m->dbg_logits + (size_t)i * m->max_batch * V;
        FGArgs g;
        memset(&g, 0, sizeof g);
        g.x = pin + (size_t)(rows - 1) * Dd; g.xrs = (long long)rows * Dd; g.nw = m->dec.norm.p; g.eps = c.decoder.rms_eps; g.out = logits; g.ors = V;
        if (!(g_skip & 64)) CS_TRY(launch_gemv(m->audio_head[i - 1], 1, 0, g, B, r.st));
        if (!(g_skip & 128)) CS_TRY(launch_sample(logits, V, temp, top_k, uniforms ? uniforms + i : nullptr, ncb, codes + i, ncb, B, r.st));
      }
      rows = 1;
    }
  } else {
  CS_TRY(r.lin(m->c0_head, last_h, last_rs, 1, logits, V, nullptr));
  if (!r.dry) {
    CS_TRY(launch_sample(logits, V, temp, top_k, uniforms, ncb, codes, ncb, B, r.st));
    // curr = [last_h, embed_audio(0, c0)]
    hipLaunchKernelGGL(copy_rows_kernel, dim3(B), dim3(256), 0, r.st, last_h, last_rs, curr, (long long)2 * D, D);
    KK_CHECK_LAUNCH();
    hipLaunchKernelGGL(embed_audio_kernel, dim3(B), dim3(256), 0, r.st, codes, ncb, m->audio_emb.p, 0, V, D, curr, 2, 1);
    KK_CHECK_LAUNCH();
  }
  int rows = 2, dpos = 0;
  for (int i = 1; i < ncb; ++i) {
    r.used = inner;
    CS_TRY(r.lin(m->proj, curr, (long long)rows * D, rows, pin, (long long)rows * Dd, nullptr));
    CS_TRY(stack_forward(r, m->dec, pin, rows, dpos, dn));
    if (r.used > peak) peak = r.used;
    dpos += rows;
    const float* dl = dn ? dn + (size_t)(rows - 1) * Dd : nullptr;
    if (!r.dry && m->dbg_logits) logits = m->dbg_logits + (size_t)i * m->max_batch * V;
    CS_TRY(r.lin(m->audio_head[i - 1], dl, (long long)rows * Dd, 1, logits, V, nullptr));
    if (!r.dry) {
      CS_TRY(launch_sample(logits, V, temp, top_k, uniforms ? uniforms + i : nullptr, ncb, codes + i, ncb, B, r.st));
      hipLaunchKernelGGL(embed_audio_kernel, dim3(B), dim3(256), 0, r.st, codes + i, ncb, m->audio_emb.p, i, V, D, curr, 1, 0);
      KK_CHECK_LAUNCH();
    }
    rows = 1;
  }
  }
  if (!r.dry) {
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, r.st, m->bb.pos_dev, S);
    KK_CHECK_LAUNCH();
  }
  r.used = peak;
  (void)mark;
  return 0;
}

int check_llama(const kk_llama_args& a) {
  if (a.num_layers < 1 || a.num_heads < 1 || a.num_kv_heads < 1 || a.num_heads % a.num_kv_heads != 0) return kk_fail("kk_csm_create: bad head counts");
  if (a.hidden < 1 || a.intermediate < 1) return kk_fail("kk_csm_create: bad sizes");
  if (a.head_dim != 64 && a.head_dim != 128) return kk_fail("kk_csm_create: head_dim must be 64 or 128 (llama-1B / llama-100M, sesame.py:225-273)");
  return 0;
}

}  // namespace

// the cache kernels on their own (Mimi's streaming transformer, kk_mimi.hip): interleaved-pair RoPE from a [max_pos][hd/2][2] cos|sin
// table on q (in place) and k, k / v appended to the caches at `offset`; attention of the S new queries over the cache
int kk_launch_rope_append(float* qkv, int S, int H, int KV, int hd, const float* rope, int offset, float* kc, float* vc, int max_pos, int B, hipStream_t st) {
  if (hd != 64 && hd != 128) return kk_fail("rope_append: head_dim must be 64 or 128");
  hipLaunchKernelGGL(rope_append_kernel, dim3(S, B), dim3(256), 0, st, qkv, S, H, KV, hd, rope, (const int*)nullptr, offset, kc, vc, max_pos, (const int*)nullptr);
  KK_CHECK_LAUNCH();
  return 0;
}
int kk_launch_attn_cache(const float* qkv, int S, int H, int KV, int hd, int offset, const float* kc, const float* vc, int max_pos, float scale, float* out,
                         int causal, int ctx, int B, hipStream_t st) {
  if (hd != 64 && hd != 128) return kk_fail("attn_cache: head_dim must be 64 or 128");
  hipLaunchKernelGGL(attn_cache_kernel<false>, dim3(S, H, B), dim3(128), attn_lds_bytes(max_pos, hd), st, qkv, S, H, KV, hd, (const int*)nullptr, offset,
                     const_cast<float*>(kc), const_cast<float*>(vc), max_pos, scale, out, causal, ctx, (const float*)nullptr, (const int*)nullptr);
  KK_CHECK_LAUNCH();
  return 0;
}

// make_sampler(temp, top_k) on its own (tests): logits [B][V] fp32 -> codes [B] int32, uniforms [B] (NULL or temp == 0: argmax)
extern "C" int kk_op_csm_sample(void* stream, int B, int V, const float* logits, float temperature, int top_k, const float* uniforms, int32_t* codes_out) {
  if (!logits || !codes_out || B < 1 || V < 1) return kk_fail("kk_op_csm_sample: bad argument");
  return launch_sample(logits, V, temperature, top_k, uniforms, 1, codes_out, 1, B, (hipStream_t)stream);
}

extern "C" int kk_csm_create(const kk_csm_config* cfg, kk_csm** out) {
  if (!cfg || !out) return kk_fail("kk_csm_create: null argument");
  if (cfg->audio_num_codebooks < 1 || cfg->audio_vocab_size < 1 || cfg->audio_vocab_size > 8192 || cfg->text_vocab_size < 1 || cfg->max_seq_len < 2)
    return kk_fail("kk_csm_create: bad configuration");
  CS_TRY(check_llama(cfg->backbone));
  CS_TRY(check_llama(cfg->decoder));
  kk_csm* m = new kk_csm();
  m->cfg = *cfg;
  m->bb.a = cfg->backbone;
  m->dec.a = cfg->decoder;
  *out = m;
  return 0;
}

// A second generator on the SAME device weights: the packed matrices of a finalized generator are immutable, everything a frame mutates (KV
// caches, positions, padding, logits, captured graphs) is per generator.  Concurrent jobs (own stream / thread each) share one copy of the
// 1.6 B parameters this way.  `m` must outlive the generators made from it.  Call kk_csm_setup_caches on the new one.
extern "C" int kk_csm_share(const kk_csm* m, kk_csm** out) {
  if (!m || !out || !m->finalized) return kk_fail("kk_csm_share: needs a finalized generator");
  kk_csm* c = new (std::nothrow) kk_csm(*m);  // descriptors + resolved device pointers of the weights
  if (!c) return kk_fail("kk_csm_share: out of memory");
  c->weights_of = m->weights_of ? m->weights_of : m;
  for (Stack* s : {&c->bb, &c->dec}) {
    s->kc = s->vc = nullptr;
    s->offset = 0;
    s->pos_dev = s->pad_dev = nullptr;
  }
  c->max_batch = 0;
  c->dbg_logits = nullptr;
  c->graphs.clear();
  c->reset_pending = c->pad_pending = false;
  c->pad_host.clear();
  c->cap_stream = nullptr;
  *out = c;
  return 0;
}

extern "C" void kk_csm_destroy(kk_csm* m) {
  if (!m) return;
  if (m->dev && !m->weights_of) (void)hipFree(m->dev);
  if (m->devb && !m->weights_of) (void)hipFree(m->devb);
  if (m->proj_table && !m->weights_of) (void)hipFree(m->proj_table);
  for (Stack* s : {&m->bb, &m->dec}) {
    if (s->kc) (void)hipFree(s->kc);
    if (s->vc) (void)hipFree(s->vc);
  }
  if (m->dbg_logits) (void)hipFree(m->dbg_logits);
  if (m->bb.pos_dev) (void)hipFree(m->bb.pos_dev);
  if (m->bb.pad_dev) (void)hipFree(m->bb.pad_dev);
  for (auto& g : m->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
  delete m;
}

// Weight storage of the linears in the single-token steps.  KK_DTYPE_BF16: every Linear matrix is rounded to bf16 once (lossless for a
// bf16 checkpoint -- the reference keeps the checkpoint's dtype, tts/utils.py:217-262) and the skinny GEMM streams 2-byte weights (half
// the bytes per frame; the 212 MB depth decoder then fits the 256 MB Infinity Cache across its 31 steps); activations, accumulation,
// KV cache and logits stay fp32.  Call before kk_csm_finalize.
extern "C" int kk_csm_set_weight_dtype(kk_csm* m, int dtype) {
  if (!m) return kk_fail("kk_csm_set_weight_dtype: null model");
  if (m->finalized) return kk_fail("kk_csm_set_weight_dtype: call before kk_csm_finalize");
  if (dtype != KK_DTYPE_F32 && dtype != KK_DTYPE_BF16) return kk_fail("kk_csm_set_weight_dtype: F32 or BF16");
  m->wdt = dtype == KK_DTYPE_BF16 ? KK_BF16 : KK_F32;
  return 0;
}

extern "C" int kk_csm_load_tensor(kk_csm* m, const char* name, const int64_t* shape, int ndim, const float* data) {
  if (!m || !name || !shape || !data || ndim < 1) return kk_fail("kk_csm_load_tensor: bad argument");
  if (m->finalized) return kk_fail("kk_csm_load_tensor: model already finalized");
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  m->host[name].assign(data, data + n);
  return 0;
}

extern "C" int kk_csm_finalize(kk_csm* m, void* stream) {
  if (!m) return kk_fail("kk_csm_finalize: null model");
  if (m->finalized) return kk_fail("kk_csm_finalize: already finalized");
  const kk_csm_config& c = m->cfg;
  Packer P{m, ""};
  const int D = c.backbone.hidden, Dd = c.decoder.hidden, V = c.audio_vocab_size, ncb = c.audio_num_codebooks;
  pack_stack(P, "backbone", m->bb, c.max_seq_len);
  pack_stack(P, "decoder", m->dec, ncb + 1);
  m->text_emb = P.vec("text_embeddings.weight", (size_t)c.text_vocab_size * D);
  m->audio_emb = P.vec("audio_embeddings.weight", (size_t)V * ncb * D);
  m->proj = P.linear({"projection.weight"}, {Dd}, D);
  m->c0_head = P.linear({"codebook0_head.weight"}, {V}, D);
  m->audio_head.resize(ncb > 1 ? ncb - 1 : 0);
  if (ncb > 1) {
    const std::vector<float>* ah = P.get("audio_head", (size_t)(ncb - 1) * Dd * V);
    if (ah)
      for (int i = 0; i < ncb - 1; ++i) m->audio_head[i] = P.linear({}, {V}, Dd, ah->data() + (size_t)i * Dd * V);  // [Dd][V] used as x @ W
  }
  if (!P.err.empty()) return kk_fail(("kk_csm_finalize: " + P.err).c_str());
  {  // KK_CSM_NT (A/B, measured equal within noise: 5.62 ms per frame each): "all" (default) nontemporal weight loads everywhere, "none" plain loads, "bb": the backbone
    // streams and the depth decoder (222 MB re-read 31 times per frame, inside the reach of the 256-MB memory-side cache) is cacheable
    const char* e = getenv("KK_CSM_NT");
    const std::string mode = e ? e : "all";
    const int dec_cached = mode != "all", bb_cached = mode == "none";
    for (auto& L : m->dec.layers) L.qkv.cached = L.o.cached = L.gu.cached = L.down.cached = dec_cached;
    for (auto& L : m->bb.layers) L.qkv.cached = L.o.cached = L.gu.cached = L.down.cached = bb_cached;
    for (auto& l : m->audio_head) l.cached = dec_cached;
    m->proj.cached = dec_cached;
    m->c0_head.cached = bb_cached;
  }
  if (hipMalloc((void**)&m->dev, m->pack.size() * sizeof(float)) != hipSuccess) return kk_fail("kk_csm_finalize: hipMalloc failed");
  if (hipMemcpyAsync(m->dev, m->pack.data(), m->pack.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return kk_fail("kk_csm_finalize: upload failed");
  if (!m->packb.empty()) {
    if (hipMalloc((void**)&m->devb, m->packb.size() * 2) != hipSuccess) return kk_fail("kk_csm_finalize: hipMalloc failed");
    if (hipMemcpyAsync(m->devb, m->packb.data(), m->packb.size() * 2, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
      return kk_fail("kk_csm_finalize: upload failed");
  }
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return kk_fail("kk_csm_finalize: stream sync failed");
  resolve(m, m->bb); resolve(m, m->audio_emb); resolve(m, m->proj);
  if (m->proj.wm && ncb > 1) {
    // projection(audio_embeddings): every input the depth decoder's later steps can see (sesame.py:373-392: curr_h = projection(embed(c_{i-1})))
    const size_t rows = (size_t)V * ncb;
    if (hipMalloc((void**)&m->proj_table, rows * Dd * sizeof(float)) != hipSuccess) return kk_fail("kk_csm_finalize: hipMalloc failed");
    GPArgs g;
    memset(&g, 0, sizeof g);
    g.x = m->audio_emb.p; g.xrs = D; g.w = m->proj.wm; g.K = D; g.N = Dd; g.M = (int)rows; g.nsub = m->proj.nsub; g.out = m->proj_table; g.ors = Dd;
    if (launch_gemmp(g, (hipStream_t)stream) != 0 || hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return kk_fail("kk_csm_finalize: projection table failed");
  }
  resolve(m, m->bb); resolve(m, m->dec); resolve(m, m->text_emb); resolve(m, m->audio_emb); resolve(m, m->proj); resolve(m, m->c0_head);
  for (auto& l : m->audio_head) resolve(m, l);
  m->host.clear();
  std::vector<float>().swap(m->pack);
  std::vector<uint16_t>().swap(m->packb);
  m->finalized = true;
  return 0;
}

// SesameModel.setup_caches / reset_caches (sesame.py:320-345): device KV caches for `max_batch` items; positions restart at 0
extern "C" int kk_csm_setup_caches(kk_csm* m, int max_batch) {
  if (!m || !m->finalized || max_batch < 1) return kk_fail("kk_csm_setup_caches: bad argument");
  // captured frame steps have the OLD cache / logits pointers baked in: drop every graph before the buffers they point at are freed
  // (a replay after a second setup_caches would read and write freed memory)
  for (auto& g : m->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  m->graphs.clear();
  (void)hipDeviceSynchronize();  // nothing in flight may still use the caches about to be freed
  for (Stack* s : {&m->bb, &m->dec}) {
    if (s->kc) (void)hipFree(s->kc);
    if (s->vc) (void)hipFree(s->vc);
    s->kc = s->vc = nullptr;
    const size_t n = (size_t)s->a.num_layers * max_batch * s->max_pos * s->a.num_kv_heads * s->a.head_dim * 4;
    if (hipMalloc((void**)&s->kc, n) != hipSuccess || hipMalloc((void**)&s->vc, n) != hipSuccess) return kk_fail("kk_csm_setup_caches: hipMalloc failed");
    s->offset = 0;
  }
  if (m->dbg_logits) (void)hipFree(m->dbg_logits);
  m->dbg_logits = nullptr;
  if (hipMalloc((void**)&m->dbg_logits, (size_t)m->cfg.audio_num_codebooks * max_batch * m->cfg.audio_vocab_size * 4) != hipSuccess)
    return kk_fail("kk_csm_setup_caches: hipMalloc failed");
  if (!m->bb.pos_dev && hipMalloc((void**)&m->bb.pos_dev, 4) != hipSuccess) return kk_fail("kk_csm_setup_caches: hipMalloc failed");
  if (hipMemset(m->bb.pos_dev, 0, 4) != hipSuccess) return kk_fail("kk_csm_setup_caches: memset failed");
  if (m->bb.pad_dev) (void)hipFree(m->bb.pad_dev);
  m->bb.pad_dev = nullptr;
  if (hipMalloc((void**)&m->bb.pad_dev, (size_t)max_batch * 4) != hipSuccess || hipMemset(m->bb.pad_dev, 0, (size_t)max_batch * 4) != hipSuccess)
    return kk_fail("kk_csm_setup_caches: hipMalloc failed");
  m->max_batch = max_batch;
  m->reset_pending = m->pad_pending = false;  // freshly zeroed above
  return 0;
}
// Ragged prompts: the streams of a batch are LEFT-padded to the longest prompt (padding frames carry an all-zero mask); pad[b] = number of
// padding frames of item b.  Item b's token in cache slot p then has position p - pad[b] (RoPE) and sees the slots >= pad[b] only, so its
// results are bit-identical to running it alone.  Call on an empty cache (after setup / reset), before the prompt block; reset clears it.
extern "C" int kk_csm_set_padding(kk_csm* m, int B, const int32_t* pad_host) {
  if (!m || !m->bb.pad_dev || B < 1 || B > m->max_batch || !pad_host) return kk_fail("kk_csm_set_padding: bad argument (call kk_csm_setup_caches first)");
  if (m->bb.offset != 0) return kk_fail("kk_csm_set_padding: the cache is not empty");
  for (int b = 0; b < B; ++b)
    if (pad_host[b] < 0 || pad_host[b] >= m->bb.max_pos) return kk_fail("kk_csm_set_padding: padding out of range");
  m->pad_host.assign((size_t)m->max_batch, 0);
  for (int b = 0; b < B; ++b) m->pad_host[b] = pad_host[b];
  m->pad_pending = true;  // uploaded on the next frame's stream
  return 0;
}
extern "C" int kk_csm_reset_caches(kk_csm* m) {
  if (!m) return kk_fail("kk_csm_reset_caches: null model");
  m->bb.offset = 0;
  m->dec.offset = 0;
  m->reset_pending = true;  // position counter and padding are cleared on the next frame's stream
  m->pad_pending = false;
  return 0;
}
extern "C" int kk_csm_position(const kk_csm* m) { return m ? m->bb.offset : -1; }

extern "C" size_t kk_csm_workspace_bytes(kk_csm* m, int B, int S) {
  if (!m || !m->finalized || B <= 0 || S <= 0) return 0;
  Run r{m, nullptr, B, nullptr, 0, 0, true, false};
  if (run_frame(r, S, nullptr, nullptr, 0.f, 1, nullptr, nullptr) != 0) return 0;
  return r.used + 256;
}

extern "C" int kk_csm_generate_frame(kk_csm* m, void* stream, int B, int S, const int32_t* tokens, const float* tokens_mask, float temperature,
                                     int top_k, const float* uniforms, void* workspace, size_t workspace_bytes, int32_t* codes_out) {
  if (!m || !m->finalized) return kk_fail("kk_csm_generate_frame: model not finalized");
  if (m->max_batch < 1) return kk_fail("kk_csm_generate_frame: call kk_csm_setup_caches first");
  if (B <= 0 || B > m->max_batch || S <= 0 || !tokens || !tokens_mask || !workspace || !codes_out) return kk_fail("kk_csm_generate_frame: bad argument");
  if (m->bb.offset + S > m->bb.max_pos) return kk_fail("kk_csm_generate_frame: sequence exceeds max_seq_len");
  if (S > 1 && m->bb.offset != 0) return kk_fail("kk_csm_generate_frame: a multi-token block must start an empty cache (sesame.py:41-48)");
  if (workspace_bytes < kk_csm_workspace_bytes(m, B, S)) return kk_fail("kk_csm_generate_frame: workspace too small");
  if (m->reset_pending) {  // deferred kk_csm_reset_caches, in stream order before this frame (never inside a capture)
    if (hipMemsetAsync(m->bb.pos_dev, 0, 4, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(m->bb.pad_dev, 0, (size_t)m->max_batch * 4, (hipStream_t)stream) != hipSuccess)
      return kk_fail("kk_csm_generate_frame: cache reset failed");
    m->reset_pending = false;
  }
  if (m->pad_pending) {  // deferred kk_csm_set_padding
    if (hipMemcpyAsync(m->bb.pad_dev, m->pad_host.data(), (size_t)m->max_batch * 4, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
      return kk_fail("kk_csm_generate_frame: padding upload failed");
    m->pad_pending = false;
  }
  auto eager = [&](void* on_stream) -> int {
    Run r{m, (hipStream_t)on_stream, B, (char*)workspace, workspace_bytes, 0, false, false};
    return run_frame(r, S, tokens, tokens_mask, temperature, top_k, uniforms, codes_out);
  };
  int rc;
  if (!m->graph_mode || S != 1) {
    rc = eager(stream);
  } else {
    // the single-token step (~1400 launches) as one hipGraphLaunch: every position-dependent kernel reads the device counter
    unsigned tbits;
    memcpy(&tbits, &temperature, 4);
    const std::vector<unsigned long long> key = {(unsigned long long)B, (unsigned long long)(uintptr_t)tokens, (unsigned long long)(uintptr_t)tokens_mask,
        (unsigned long long)tbits, (unsigned long long)top_k, (unsigned long long)(uintptr_t)uniforms, (unsigned long long)(uintptr_t)workspace,
        (unsigned long long)workspace_bytes, (unsigned long long)(uintptr_t)codes_out, (unsigned long long)g_skip, (unsigned long long)(uintptr_t)g_ts};
    kk_csm::GraphEntry* ge = nullptr;
    for (auto& g : m->graphs)
      if (g.key == key) ge = &g;
    if (!ge) {
      if (m->graphs.size() >= 8) {
        if (m->graphs.front().exec) (void)hipGraphExecDestroy(m->graphs.front().exec);
        if (m->graphs.front().graph) (void)hipGraphDestroy(m->graphs.front().graph);
        m->graphs.erase(m->graphs.begin());
      }
      m->graphs.emplace_back();
      ge = &m->graphs.back();
      ge->key = key;
    }
    if (ge->seen == 0) {
      ge->seen = 1;
      rc = eager(stream);
    } else {
      rc = 0;
      if (ge->seen == 1) {
        if (!m->cap_stream && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess) return kk_fail("kk_csm: hipStreamCreate failed");
        if (hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return kk_fail("kk_csm: hipStreamBeginCapture failed");
        rc = eager((void*)m->cap_stream);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(m->cap_stream, &g);
        if (rc != 0) {
          if (g) (void)hipGraphDestroy(g);
          return rc;
        }
        if (e != hipSuccess || !g) return kk_fail("kk_csm: hipStreamEndCapture failed");
        hipGraphExec_t ex = nullptr;
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
          (void)hipGraphDestroy(g);
          return kk_fail("kk_csm: hipGraphInstantiate failed");
        }
        ge->graph = g;
        ge->exec = ex;
        ge->seen = 2;
      }
      if (hipGraphLaunch(ge->exec, (hipStream_t)stream) != hipSuccess) return kk_fail("kk_csm: hipGraphLaunch failed");
    }
  }
  if (rc == 0) m->bb.offset += S;
  return rc;
}

extern "C" int kk_csm_debug_timestamps(unsigned long long* buf, int capacity) {
  g_ts = buf;
  g_ts_cap = buf ? capacity : 0;
  g_ts_next = 0;
  return 0;
}

extern "C" int kk_csm_debug_skip(int mask) {
  g_skip = mask & 0xffff;
  return 0;
}

extern "C" int kk_csm_set_graph_mode(kk_csm* m, int on) {
  if (!m) return kk_fail("kk_csm_set_graph_mode: null model");
  m->graph_mode = on != 0;
  return 0;
}

// logits of the last frame (tests): [n_cb][B][V] float32, device to device
extern "C" int kk_csm_debug_logits(kk_csm* m, void* stream, int B, float* dst) {
  if (!m || !m->dbg_logits || !dst || B < 1 || B > m->max_batch) return kk_fail("kk_csm_debug_logits: bad argument");
  const int V = m->cfg.audio_vocab_size;
  for (int i = 0; i < m->cfg.audio_num_codebooks; ++i)
    if (hipMemcpyAsync(dst + (size_t)i * B * V, m->dbg_logits + (size_t)i * m->max_batch * V, (size_t)B * V * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) !=
        hipSuccess)
      return kk_fail("kk_csm_debug_logits: copy failed");
  return 0;
}
