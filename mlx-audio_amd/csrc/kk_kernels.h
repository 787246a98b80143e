// Launch wrappers of every device kernel (host-callable, enqueue on the given stream, never sync).
#pragma once
#include "kk_common.h"

// ---- convolution family (kk_conv.hip, kk_conv_mfma.hip)
int kk_launch_conv_generic(const KKConvArgs& a, int B, int in_dtype, int out_dtype, hipStream_t st);

// bf16 MFMA implicit-GEMM convolution (kk_conv_mfma.hip)
struct KKMfmaArgs {
  const bf16_t* x;  // [B][rows][ldx], ldx >= CinP, pad channels finite (zero)
  long long xbs;
  int ldx;
  const bf16_t* w;  // packed [Kw][CoutP][CinP], zero padded
  const bf16_t* wf; // the same weights in MFMA fragment order (kk_mfma4_pack_index); enables the variant-4 kernel
  int CinP, CoutP;
  int Cin;  // channels of x that carry data; [Cin, CinP) are masked to zero while staging (pad channels may hold anything); 0 = CinP
  const float* bias;  // [CoutP] or null
  void* out;          // bf16 or fp32
  long long obs;
  int ldo;
  const void* res;  // same dtype as out
  long long rbs;
  int ldr;
  int Cout;  // channels actually written (multiple of 8)
  int Kw, mode, stride, pad, dil, in_shift;
  int Q, Lo_rows;
  KKLen lin, lout;
  float in_slope, scale;
  int accumulate, act;
  float act_slope;
  int in_act;  // variant 4 only: KK_ACT_ELU = elu(x, 1) applied to the input while it is staged (Mimi SEANet); 0 = leaky-relu(in_slope)
  int dbg;  // timing experiments only (KK_MFMA_DBG): bit0 skip W reloads, bit1 skip X reloads
  // fused input transform (AdaIN apply + activation while staging X):  y = act(x * nrm_a[b][c] + nrm_b[b][c])
  const float* nrm_a;  // [B][nrm_stride], zero for pad channels; null = no transform
  const float* nrm_b;
  int nrm_stride;
  int nrm_act;            // KK_ACT_NONE / LRELU / SNAKE
  float nrm_slope;
  const float* nrm_alpha; // [nrm_C] Snake alpha
  int nrm_C;
  // fused output statistics: per-tile column sums of the STORED values, part[((b*stat_ntiles + tile)*2 + {0,1})*Cout + n]
  float* stat_part;
  int stat_ntiles;
  // FLAT rows (variants 2 / 4, k = 1 only): the launch covers the B x flat_T rows of a dense [B][flat_T][C] tensor as ONE item (B = 1, Q = B * flat_T),
  // so that short utterances share 192-row tiles (Albert at T = 130: 22 row tiles instead of 32); row r belongs to item r / flat_T and is
  // stored as zeros past that item's length flat_len.  0 = off.
  int flat_T;
  KKLen flat_len;
  float post_slope;  // variants 4 / 5 (bf16 out): LeakyReLU(post_slope) of the final stored value; 0 or 1 = none
};
// Streaming matrix-core Linear over token rows (kk_linear_rows.hip): out[b][t][:] = act(x[b][t][:] W + bias), zeros past an utterance's length
struct KKLinMfmaArgs {
  const bf16_t* x;   // [items][rows][ldx] bf16; the first K channels of a row are read
  long long xbs;
  int ldx;
  const bf16_t* wl;  // fragment pack [ceil(N / 16)][K / 32][64 lanes][8] (kk_linear_pack_index)
  const float* bias; // [Nb] or null
  int Nb;
  bf16_t* out;       // [items][rows][ldo]
  long long obs;
  int ldo;
  int K, N;          // N: channels written (even; columns >= the layer's real width carry zero weights and zero bias)
  int rows, items;   // rows per item; flat: the items are dense and form ONE row axis of items * rows rows
  int flat;
  KKLen len;         // valid rows per item
  int act;           // KK_ACT_NONE or KK_ACT_GELU (exact erf)
};
__host__ __device__ inline long long kk_linear_pack_index(int o, int i, int K) {  // weight (output column o, input channel i) of a [N][K] Linear
  return (((long long)(o >> 4) * (K >> 5) + (i >> 5)) * 64 + ((i & 31) >> 3) * 16 + (o & 15)) * 8 + (i & 7);
}
int kk_launch_linear_rows_mfma(const KKLinMfmaArgs& a, hipStream_t st);
bool kk_mfma_eligible(int Cin, int Cout, int Kw, int mode, int stride, int dil);
int kk_mfma_tile_rows(int Q);  // 128 or 256 output rows per workgroup for a launch covering Q rows per phase
int kk_launch_conv_mfma(const KKMfmaArgs& a, int B, int out_dtype, hipStream_t st);
// variant 4 (kk_conv_mfma4.hip): W fragments straight from global memory into the MFMA operand registers; 192-row tiles, bf16 out
int kk_launch_conv_mfma4(const KKMfmaArgs& a, int B, int out_dtype, hipStream_t st);
// variant 5 (kk_conv_mfma5.hip): wave-specialised persistent kernel (4 MFMA waves + 4 service waves per CU) for stride-1 convolutions
bool kk_mfma5_eligible(const KKMfmaArgs& a, int out_dtype);
int kk_launch_conv_mfma5(const KKMfmaArgs& a, int B, int out_dtype, hipStream_t st);
long long kk_mfma4_pack_index(int tap, int cout, int k, int CoutP, int CinP);
int kk_launch_pack_w_frag(const void* w, void* wf, int Kw, int CoutP, int CinP, hipStream_t st);
// rows per statistics tile of the kernel kk_launch_conv_mfma will pick for these arguments
int kk_mfma_stat_tile_rows(const KKMfmaArgs& a, int out_dtype);

// ---- normalisation family (kk_norm.hip)
struct KKStatsArgs {
  // optional AdaIN folding: pa = rstd*(1+gamma), pb = beta - mean*pa  (gamma/beta from the style projection gb)
  const float* gb;
  int gbs;
  float* pa;
  float* pb;
  int pstride, Cp;  // pa/pb row pitch and number of channels written (pads get 0)
  int fused;        // 1: `partial` holds un-shifted per-tile sums written by conv epilogues (sum ALL nchunk tiles)
  const void* x;
  long long xbs;
  int ldx;
  int C;
  int Lmax;
  KKLen len;
  float* partial;  // scratch: kk_stats_partial_floats()
  int nchunk;      // filled by the launcher
  int rows_per_chunk;
  float* mean;  // [B][C]
  float* rstd;  // [B][C]
  float eps;
};
size_t kk_stats_partial_floats(int B, int C, int Lmax, int rows_per_chunk);
int kk_launch_instnorm_stats(KKStatsArgs a, int B, int dtype, hipStream_t st);
// adds ONE extra row of a bf16 tensor (row pointer per utterance = x + b*xbs) to tile 0 of the fused statistics partials
int kk_launch_stat_add_row(const void* x, long long xbs, float* part, int ntiles, int C, int B, hipStream_t st);
int kk_launch_norm_finalize(KKStatsArgs a, int B, hipStream_t st);  // fused partials -> mean/rstd (+ pa/pb)

struct KKAdainArgs {
  const void* x;
  long long xbs;
  int ldx;
  void* out;
  long long obs;
  int ldo;
  int C, Cpad;   // channels computed / channels written (pad channels are zero-filled)
  int Lmax_out;  // rows of the output buffer to cover
  KKLen len_in;  // valid input rows; valid output rows = pool ? 2*Lin : Lin
  const float* mean;
  const float* rstd;  // [B][C]
  const float* gb;    // style projection: gamma = gb[b*gbs + c], beta = gb[b*gbs + C + c]
  int gbs;
  int act;  // KK_ACT_NONE / LRELU / SNAKE
  float slope;
  const float* alpha;  // [C], Snake
  int pool;            // 1: depth-wise convT k3 s2 p1 + front pad after the activation
  const float* pool_w; // [C][3] (weight-norm folded)
  const float* pool_b; // [C]
  int fast;            // 1: hardware sin approximation (bf16 mode)
};
int kk_launch_adain_act(const KKAdainArgs& a, int B, int dtype, hipStream_t st);

struct KKLnArgs {
  const void* x;
  long long xbs;
  int ldx;
  const void* res;  // optional: LN(x + res)
  long long rbs;
  int ldr;
  void* out;
  long long obs;
  int ldo;
  int C;
  int Lmax;
  KKLen len;
  const float* w;
  const float* bias;  // affine (when gb == null)
  const float* gb;    // AdaLayerNorm: (1 + gamma) * xhat + beta
  int gbs;
  float eps;
  int act;
  float slope;
};
int kk_launch_layernorm(const KKLnArgs& a, int B, int dtype, hipStream_t st);

// ---- bidirectional LSTM recurrence (kk_lstm.hip)
struct KKLstmArgs {
  const float* xproj;  // [B][Lmax][2][4H] : x @ Wx^T + (b_ih + b_hh), gate order i,f,g,o
  const float* whT;    // [2][H][4H]       : Wh transposed per direction
  void* out;           // [B][Lmax][ldo], forward at channel 0, backward at channel H
  long long obs;
  int ldo;
  int H;
  int Lmax;
  KKLen len;
};
int kk_launch_lstm(const KKLstmArgs& a, int B, int dtype, hipStream_t st);
// bf16 mode, H = 256: Wh [2][4H][H] bf16 kept in registers + LDS for the whole sequence
int kk_launch_lstm_h256_bf16(const KKLstmArgs& a, const void* whb, int B, int dtype, hipStream_t st);

// ---- Albert pieces (kk_albert.hip)
struct KKEmbedArgs {
  const int* ids;  // [B][Tmax]
  const float* word;
  const float* pos;
  const float* type;  // embedding tables [*][E]
  const float* ln_w;
  const float* ln_b;
  void* out;  // [B][Tmax][ldo]
  long long obs;
  int ldo;
  int E;
  int Tmax;
  KKLen len;
  float eps;
};
int kk_launch_albert_embed(const KKEmbedArgs& a, int B, int dtype, hipStream_t st);

struct KKAttnArgs {
  const void* qkv;  // [B][Tmax][ld]: q at h*64, k at hs + h*64, v at 2*hs + h*64
  long long bs;
  int ld;
  void* out;  // [B][Tmax][ldo] context at h*64
  long long obs;
  int ldo;
  int heads, hs;
  int Tmax;
  KKLen len;
  float scale;  // 1/sqrt(64)
};
int kk_launch_attention(const KKAttnArgs& a, int B, int dtype, hipStream_t st);

// ---- misc (kk_misc.hip)
// out[b][o] = sum_j wT[j][o] * s[b][soff + j] + bias[o]   (all style projections of one style half at once)
int kk_launch_style_fc(const float* ref_s, int soff, const float* wT, const float* bias, float* out, int N, int B, hipStream_t st);
// cat[b][t][coff + j] = s[b][soff + j] for t < len[b], else 0
int kk_launch_fill_style(const float* ref_s, int soff, void* out, long long obs, int ldo, int coff, int S, int Lmax, KKLen len, int B,
                         int dtype, hipStream_t st);
// out[b][t][c] = table[ids[b][t]][c]  (TextEncoder embedding, modules.py:42)
int kk_launch_embedding(const int* ids, const float* table, void* out, long long obs, int ldo, int C, int Tmax, KKLen len, int B,
                        int dtype, hipStream_t st);
// duration head (kokoro.py:148-150): dur[b][t] = max(1, rint(sum_o sigmoid(x.W[o] + bias[o]) / speed[b])), raw sum kept in dur_f
int kk_launch_duration(const void* x, long long xbs, int ldx, const float* W, const float* bias, int Cin, int nout, const float* speed,
                       int* dur, float* dur_f, int Tmax, KKLen len, int B, int dtype, hipStream_t st);
// alignment (kokoro.py:151-156): frame_idx[b][f] = token index of frame f, lenF[b] = min(sum dur, Fmax)
int kk_launch_alignment(const int* dur, int Tmax, const int* lenT, int* frame_idx, int* lenF, int Fmax, int B, hipStream_t st);
// length regulation (kokoro.py:157,162): out[b][f][coff + c] = in[b][frame_idx[b][f]][c]
int kk_launch_gather_rows(const void* in, long long ibs, int ldi, const int* frame_idx, int Fmax, const int* lenF, void* out,
                          long long obs, int ldo, int coff, int C, int B, int dtype, hipStream_t st);
// strided channel-slice copy: out[b][l][coff + c] = in[b][l][c] (zero past len)
int kk_launch_copy_slice(const void* in, long long ibs, int ldi, void* out, long long obs, int ldo, int coff, int C, int Lmax, KKLen len,
                         int B, int dtype, hipStream_t st);
int kk_launch_lens(const int* lenF, int* lens_out /*[4][B]: 2F, 20F, 120F+1, 600F*/, int B, int up0, int up1, int hop, hipStream_t st);

// ---- harmonic source + STFT + iSTFT head (kk_source.hip)
struct KKSourceArgs {
  const float* f0;  // [B][L2max] predicted F0 curve (2 frames per 25 ms frame)
  int L2max;
  const int* len2;  // valid entries per utterance (2F)
  float* phase;     // scratch [B][9][L2max]: 300 * 2*pi*cumsum(rad)
  const float* lin_w;  // [9]
  float lin_b;
  const float* noise;  // optional injected N(0,1) [B][Nmax][9]; null -> Philox (seed) or zero
  unsigned long long seed;
  const unsigned long long* seed_dev;  // graph replay: the seed lives in device memory (null = use `seed`)
  int noise_mode;  // 0 zero, 1 injected, 2 philox
  float* har_source;  // [B][Nmax]
  int Nmax;           // 300 * L2max
  int upsample;       // 300
};
int kk_launch_source(const KKSourceArgs& a, int B, hipStream_t st);
// har[b][t][0..10] = |STFT|, [11..21] = angle  (n_fft 20, hop 5, symmetric Hann, reflect pad; istftnet.py:463-495)
int kk_launch_stft20(const float* har_source, int Nmax, const int* lenN, void* har, long long obs, int ldo, int Tfmax, int B, int dtype,
                     hipStream_t st);
// iSTFT head (istftnet.py:804-806,497-523; utils.py:104-158): x[b][t][22] -> wav[b][5*(frames-1)]
int kk_launch_istft_head(const void* x, long long xbs, int ldx, const int* len_frames, int Tfmax, float* wav, long long wbs, int B,
                         int dtype, int fast, hipStream_t st);
// fused vocoder head (kk_head.hip, bf16 mode): LeakyReLU -> conv_post (128 -> 22, k 7) -> exp / sin -> inverse STFT -> overlap-add
struct KKHeadArgs {
  const bf16_t* x;   // [B][Tfmax][ldx] stage-1 output of the generator (128 channels)
  long long xbs;
  int ldx;
  const bf16_t* wf;  // conv_post weights in the head kernel's fragment order (kk_head_pack_index), output channels 22..31 zero
  const float* bias; // [>= 22]
  float in_slope;    // LeakyReLU on the input (istftnet.py:798: 0.01)
  const int* len_frames;  // valid frames per utterance (null: Tfmax)
  int Tfmax;
  float* wav;        // [B][wbs] fp32, 5 * (Tfmax - 1) samples written per utterance
  long long wbs;
  bf16_t* cp_out;    // optional (debug): the conv_post tensor [B][Tfmax][cp_ld] as the stand-alone conv would have stored it
  long long cp_bs;
  int cp_ld;
  float hann_per[20];  // periodic Hann (utils.py:121), for the partial window sums at utterance edges
  int dbg;             // timing experiments only (KK_HEAD_DBG): bit 0 one k-step, bit 1 no frame arithmetic, bit 2 cache-resident input
};
bool kk_head_eligible(int Cin, int Cout, int Kw, int n_fft, int hop);
long long kk_head_pack_index(int tap, int cout, int cin);
size_t kk_head_pack_elems();
int kk_launch_conv_post_istft(const KKHeadArgs& a, int B, hipStream_t st);
int kk_launch_pack_head_w(const bf16_t* w, bf16_t* wf, hipStream_t st);  // device-side re-layout [7][22][128] -> fragment order (tests)
// dtype-converting strided copy (debug hooks)
int kk_launch_set_u64(unsigned long long* dst, unsigned long long v, hipStream_t st);  // one 8-byte device store (graph-replay seed)
int kk_launch_convert(const void* src, int sdt, long long sbs, int lds, void* dst, int ddt, long long dbs, int ldd, int C, int rows, int B,
                      hipStream_t st);

// ---- MX-fp8 linears (kk_mxfp8.hip).  Operands in MFMA fragment order, see the header of kk_mxfp8.hip.
struct KKFp8Args {
  const uint4* aq;          // activation fragments (kk_launch_mxfp8_quant_rows)
  const unsigned char* as;  // activation scale bytes
  const uint4* wq;          // weight fragments (kk_mxfp8_pack_weight_host, uploaded)
  const unsigned char* ws;
  int M, N, K;              // flat rows, outputs (N % 64 == 0), inputs (K % 64 == 0)
  const float* bias;        // [N] or null
  bf16_t* out;              // out[m * ldo + n]
  int ldo;
  int rows_per_item;        // row m belongs to utterance m / rows_per_item at position m % rows_per_item
  KKLen lout;               // rows at positions past the utterance's length are written as zeros
  int act;                  // KK_ACT_NONE or KK_ACT_GELU (exact erf)
};
size_t kk_mxfp8_q_bytes(int rows, int K);
size_t kk_mxfp8_s_bytes(int rows, int K);
bool kk_mxfp8_eligible(int K, int N);
// host: fp32 [N][K] -> e4m3 fragments + E8M0 scale bytes, one scale per `group` inputs (a multiple of 32)
int kk_mxfp8_pack_weight_host(const float* w, int N, int K, int group, unsigned char* q, unsigned char* s);
int kk_launch_mxfp8_quant_rows(const void* x_bf16, int ldx, int M, int K, void* aq, void* as, hipStream_t st);
int kk_launch_linear_mxfp8(const KKFp8Args& a, hipStream_t st);

// ---- KV-cache kernels of kk_csm.hip, shared with Mimi's streaming transformer (kk_mimi.hip)
int kk_launch_rope_append(float* qkv, int S, int H, int KV, int hd, const float* rope, int offset, float* kc, float* vc, int max_pos, int B, hipStream_t st);
int kk_launch_attn_cache(const float* qkv, int S, int H, int KV, int hd, int offset, const float* kc, const float* vc, int max_pos, float scale, float* out,
                         int causal, int ctx, int B, hipStream_t st);
