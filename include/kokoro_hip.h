/* kokoro_hip.h -- C ABI of the MI355X-native Kokoro-82M acoustic path (libkokoro_hip.so).
 *
 * The reference (stevenmiller888/mlx-audio) has no FFI: its hot path sits behind plain Python
 * callables that dispatch every op to the MLX runtime.  This header is the boundary a maintainer
 * would bind instead (ctypes stub in INTEGRATION.md).  Each entry point names the reference
 * interface it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; kk_last_error() gives the message.
 *     No C++ exception crosses this boundary.
 *   - all data pointers are DEVICE pointers (hipMalloc / torch.cuda tensors) unless the parameter is
 *     documented as host memory.  The caller owns every buffer; the library never frees or keeps one
 *     beyond the call, except the weights it copied in kk_load_tensor/kk_finalize.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*) and is asynchronous; no entry
 *     point synchronises the device except kk_finalize.
 *   - a kk_model is IMMUTABLE after kk_finalize and may be shared by any number of threads / streams (one copy of the weights).  Everything a
 *     forward mutates -- the cache of captured graphs (kk_set_graph_mode), the side stream + fork / join events of a forward, the debug hooks
 *     and switches, the profile brackets -- lives in a kk_context, made by kk_context_create: ONE CONTEXT PER STREAM / THREAD in flight, each with its
 *     own caller-owned workspace (bench.py --streams, TTSService(replicas=...) share one model).  A context is not thread safe; the model is.
 *   - tensors are "frames-major": [B][L][C] with C contiguous.
 */
#ifndef KOKORO_HIP_H
#define KOKORO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KK_ABI_VERSION 2 /* 2: forward / graph / debug / profile entry points take a kk_context: round 3 */

enum { KK_DTYPE_F32 = 0, KK_DTYPE_BF16 = 1, KK_DTYPE_I32 = 2, KK_DTYPE_F16 = 3 };
enum { KK_NOISE_ZERO = 0, KK_NOISE_INJECTED = 1, KK_NOISE_PHILOX = 2 };

typedef struct kk_model kk_model;
typedef struct kk_context kk_context;

/* Hyper-parameters: mlx_audio/tts/models/kokoro/kokoro.py:47-63 (ModelConfig), values pinned by
 * mlx_audio/tts/tests/test_models.py:92-122; Albert defaults mlx_audio/tts/models/kokoro/modules.py:418-435. */
typedef struct kk_config {
  int32_t n_token, hidden_dim, style_dim, n_layer, max_dur, text_encoder_kernel_size;
  int32_t plbert_hidden, plbert_heads, plbert_intermediate, plbert_max_pos, plbert_layers, plbert_embedding;
  int32_t decoder_hidden;             /* 1024, hard-coded at istftnet.py:917-932 */
  int32_t upsample_initial_channel;
  int32_t n_upsamples;                /* 2 */
  int32_t upsample_rates[4];          /* 10, 6 */
  int32_t upsample_kernel_sizes[4];   /* 20, 12 */
  int32_t n_resblock_kernels;         /* 3 */
  int32_t resblock_kernel_sizes[4];   /* 3, 7, 11 */
  int32_t resblock_dilations[4][3];   /* {1,3,5} x3 */
  int32_t gen_istft_n_fft, gen_istft_hop_size; /* 20, 5 */
  int32_t compute_dtype;              /* KK_DTYPE_F32 (exact path) or KK_DTYPE_BF16 (MFMA path) */
} kk_config;

/* Model(config)  --  kokoro.py:83-113 */
int kk_create(const kk_config* cfg, kk_model** out);
void kk_destroy(kk_model* m);

/* model.load_weights(...) after Model.sanitize  --  mlx_audio/tts/utils.py:217-262, kokoro.py:172-252,
 * istftnet.py:965-979.  `name` is the MLX-side parameter name; `data` is HOST memory in `dtype`
 * (F32 / BF16 / F16).  Conv weights may arrive in either layout ([O,K,I] MLX or [O,I,K] PyTorch): the
 * library decides by the expected shape, not by the reference's check_array_shape heuristic
 * (mlx_audio/tts/models/base.py:21-34).  PyTorch-side LSTM / gamma / beta names are accepted too. */
int kk_load_tensor(kk_model* m, const char* name, int dtype, const int64_t* shape, int ndim, const void* host_data);

/* Folds weight_norm once (istftnet.py:53-93,130: g*v/(||v||+1e-7), recomputed per call in the
 * reference), packs every matrix for the kernels and uploads.  Fails if a parameter is missing.
 * Synchronises `stream`. */
int kk_finalize(kk_model* m, void* stream);

/* Bytes of scratch kk_forward* needs for a batch of B utterances of at most Tmax tokens (BOS/EOS
 * included) and Fmax frames (Fmax = 0: text stage only). */
size_t kk_workspace_bytes(const kk_model* m, int B, int Tmax, int Fmax);

/* The per-stream run state of a finalized model (the reference's Model is one object because its forward is single threaded:
 * kokoro.py:83-113 holds module state such as `_pipelines`, istftnet.py:526 caches the STFT).  Any number of contexts may share one model;
 * use one per stream / thread in flight.  kk_context_workspace_bytes is kk_workspace_bytes under THIS context's debug switches (some of them
 * materialise extra intermediates).  Destroy every context before kk_destroy(model). */
int kk_context_create(kk_model* m, kk_context** out);
void kk_context_destroy(kk_context* cx);
kk_model* kk_context_model(kk_context* cx);
size_t kk_context_workspace_bytes(const kk_context* cx, int B, int Tmax, int Fmax);

/* Text stage of Model.__call__  --  kokoro.py:135-150 and :159-161:
 *   Albert -> bert_encoder -> DurationEncoder -> duration LSTM/proj -> pred_dur ; TextEncoder.
 * ids      [B][Tmax] int32, zero padded, row b = [0, ids..., 0] (kokoro.py:135)
 * lens     [B] int32, tokens per utterance including the two zeros
 * ref_s    [B][256] float32 style rows (pipeline.py:236; [:128] decoder, [128:] prosody, kokoro.py:145,165)
 * speed    [B] float32
 * pred_dur_out [B][Tmax] int32  = clip(round(sum(sigmoid(.))/speed), 1) (kokoro.py:149-150), 0 past lens[b]
 * The stage's results stay in `workspace` for kk_forward_audio. */
int kk_forward_text(kk_context* cx, void* stream, int B, int Tmax, const int32_t* ids, const int32_t* lens, const float* ref_s,
                    const float* speed, void* workspace, size_t workspace_bytes, int32_t* pred_dur_out);

/* Audio stage of Model.__call__  --  kokoro.py:151-165: alignment, F0Ntrain, Decoder, Generator, iSTFT.
 * dur       [B][Tmax] int32 durations to realise (pred_dur_out, or forced ones)
 * Fmax      frame capacity per utterance; frames past it are dropped (nframes_out reports min(sum, Fmax))
 * noise_mode / sine_noise / seed : the reference draws N(0,1) at istftnet.py:620; KK_NOISE_INJECTED reads
 *           sine_noise [B][600*Fmax][9] float32, KK_NOISE_PHILOX generates it on the fly from `seed`.
 * wav_out   [B][600*Fmax] float32 (zeros past 600*nframes)   -- Output.audio, kokoro.py:165-170
 * nframes_out [B] int32 */
int kk_forward_audio(kk_context* cx, void* stream, int B, int Tmax, const int32_t* lens, const float* ref_s, const int32_t* dur, int Fmax,
                     int noise_mode, const float* sine_noise, uint64_t seed, void* workspace, size_t workspace_bytes, float* wav_out,
                     int32_t* nframes_out);

/* Model.__call__ in one call (kokoro.py:120-170) without the reference's mid-forward host sync
 * (kokoro.py:151-153): durations are `forced_dur` if non-null, else the predicted ones, realised on
 * device and truncated at Fmax frames. */
int kk_forward(kk_context* cx, void* stream, int B, int Tmax, const int32_t* ids, const int32_t* lens, const float* ref_s, const float* speed,
               const int32_t* forced_dur, int Fmax, int noise_mode, const float* sine_noise, uint64_t seed, void* workspace,
               size_t workspace_bytes, float* wav_out, int32_t* pred_dur_out, int32_t* nframes_out);

/* Graph replay of kk_forward (off by default).  With it on, the SECOND call with an identical argument tuple (every pointer,
 * B, Tmax, Fmax, noise_mode; the seed may differ) is captured into a hipGraph on `stream` and that call and all later ones
 * are ONE hipGraphLaunch instead of ~450 kernel launches; results are bit-identical to the eager call.  Calls with debug
 * overrides or an open profile run eagerly.  The reference has no counterpart (MLX builds its own lazy graph per call,
 * kokoro.py:120-170 is re-traced every time). */
int kk_set_graph_mode(kk_context* cx, int on);

/* load_model's quantization branch  --  mlx_audio/tts/utils.py:241-260 (nn.quantize with the class predicate of :349-369; BASELINE
 * config 5).  Call between kk_create and kk_finalize when config["quantization"] is present; the weights handed to kk_load_tensor are
 * the DEQUANTISED ones (scale * q + bias per group, mlx-audio_amd/quant.py).  With compute_dtype bf16 the quantised Linear set with
 * eligible shapes (inputs % 64 == 0, outputs % 64 == 0: Albert's embedding map / QKV / dense / ffn / ffn_output and bert_encoder,
 * 99.9 % of the quantised FLOPs) is re-quantised to OCP e4m3 with one power-of-two (E8M0) scale per `group_size` inputs and runs on
 * the block-scaled fp8 matrix instruction; activations are quantised per (row, 32 inputs) on the fly.  The remaining members of the
 * set (style `fc` layers, duration_proj, embeddings: M = batch rows or table lookups) use the dequantised weights in fp32.
 * bits must be 8.  kk_quantized_layers reports how many of the 6 linears got an fp8 pack (after kk_finalize). */
int kk_set_quantization(kk_model* m, int group_size, int bits);
int kk_quantized_layers(const kk_model* m);

const char* kk_last_error(void);
int kk_abi_version(void);

/* ---- single-kernel entry points (used by the parity tests; same kernels kk_forward launches) ---- */

/* mx.conv1d / mx.conv_transpose1d / nn.Linear  --  istftnet.py:137-157.  w_packed [K][Cin][ldw] fp32 (device). */
int kk_op_conv1d(void* stream, int B, const void* x, int ldx, int Lin_rows, const int32_t* lin, const float* w_packed, int ldw,
                 const float* bias, int Cin, int Cout, int Kw, int transposed, int stride, int pad, int dil, int in_shift, float in_slope,
                 int act, float act_slope, const void* res, int ldr, float scale, int accumulate, void* out, int ldo, int Lout_rows,
                 const int32_t* lout, int in_dtype, int out_dtype);
/* the same ops on the bf16 MFMA kernel.  w_bf16 packed [Kw][CoutP][CinP] (CinP % 64 == 0, CoutP % 128 == 0, zero padded),
 * x bf16 with ldx >= CinP, out bf16 or fp32 (out_dtype), bias [CoutP]. */
int kk_op_conv1d_bf16(void* stream, int B, const void* x, int ldx, int Lin_rows, const int32_t* lin, const void* w_bf16, int CinP, int CoutP,
                      const float* bias, int Cout, int Kw, int transposed, int stride, int pad, int dil, int in_shift, float in_slope,
                      int act, float act_slope, const void* res, int ldr, float scale, int accumulate, void* out, int ldo, int Lout_rows,
                      const int32_t* lout, int out_dtype);
/* the fused resblock form of the bf16 path: y = act(x * nrm_a[b][c] + nrm_b[b][c]) applied while the input is staged
 * (AdaIN1d + Snake / LeakyReLU, istftnet.py:333-337,382) and per-tile column sums / sums of squares of the stored output
 * (stat_part [B][ntiles][2][Cout], the next InstanceNorm's statistics, istftnet.py:229-230) */
int kk_op_conv1d_bf16_fused(void* stream, int B, const void* x, int ldx, int L_rows, const int32_t* len, const void* w_bf16, int CinP,
                            int CoutP, const float* bias, int Cin, int Cout, int Kw, int pad, int dil, const float* nrm_a,
                            const float* nrm_b, int nrm_stride, int nrm_act, float nrm_slope, const float* nrm_alpha, const void* res,
                            int ldr, float scale, void* out, int ldo, float* stat_part, int* stat_ntiles_out);
/* InstanceNorm statistics + AdaIN1d + activation (+ pool)  --  istftnet.py:216-268,327-338,382,874-882 */
int kk_op_adain(void* stream, int B, const void* x, int ldx, int L_rows, const int32_t* len, int C, const float* gamma_beta, int gbs,
                int act, float slope, const float* alpha, int pool, const float* pool_w, const float* pool_b, void* out, int ldo, int Cpad,
                int Lout_rows, float* scratch, size_t scratch_floats, int dtype, int fast);
/* nn.LayerNorm / AdaLayerNorm  --  modules.py:33,71-90 */
int kk_op_layernorm(void* stream, int B, const void* x, int ldx, const void* res, int ldr, int L_rows, const int32_t* len, int C,
                    const float* w, const float* b, const float* gamma_beta, int gbs, float eps, int act, float slope, void* out, int ldo,
                    int dtype);
/* LSTM recurrence  --  modules.py:152-239 */
int kk_op_lstm(void* stream, int B, const float* xproj, const float* whT, int H, int L_rows, const int32_t* len, void* out, int ldo,
               int dtype);
/* the same recurrence for H = 256 with Wh given as bf16 [2][4H][H] (row = gate row, i|f|g|o) and kept on chip */
/* The streaming matrix-core Linear of the text side (kk_linear_rows.hip; Albert / bert_encoder / map_in Linears, modules.py:414-512, while few rows are in
 * flight) on its own: out[b][t][:] = act(x[b][t][:K] W^T + bias) for t < len[b], zeros past it.  x / out bf16 at item pitches xbs / obs elements and row
 * pitches ldx / ldo; w_bf16 [N][K] row-major bf16; pack_scratch: ceil(N / 16) * 16 * K bf16 of device memory; act 0 none, 2 exact-erf GELU. */
int kk_op_linear_rows(void* stream, int B, const void* x_bf16, long long xbs, int ldx, int rows, const int32_t* len, const void* w_bf16, int N, int K,
                      const float* bias, int act, void* pack_scratch, void* out_bf16, long long obs, int ldo);
int kk_op_lstm_bf16(void* stream, int B, const float* xproj, const void* wh_bf16, int L_rows, const int32_t* len, void* out, int ldo,
                    int dtype);
/* AlbertSelfAttention core  --  modules.py:497-512 */
int kk_op_attention(void* stream, int B, const void* qkv, int ld, int T_rows, const int32_t* len, int heads, void* out, int ldo, int dtype);
/* SourceModuleHnNSF + MLXSTFT.transform  --  istftnet.py:606-680,463-495 */
int kk_op_source_stft(void* stream, int B, const float* f0, int L2_rows, const int32_t* len2, const float* lin_w9, float lin_b,
                      int noise_mode, const float* noise, uint64_t seed, float* phase_scratch, float* har_source, void* har, int ldhar,
                      int dtype);
/* exp/sin + MLXSTFT.inverse + istft  --  istftnet.py:804-806,497-523; mlx_audio/utils.py:104-158 */
int kk_op_istft_head(void* stream, int B, const void* x, int ldx, int Tf_rows, const int32_t* len_frames, float* wav, int dtype, int fast);
/* The fused vocoder head of the bf16 mode: LeakyReLU(in_slope) -> conv_post (128 -> 22 channels, k = 7, pad 3) -> exp / sin -> inverse STFT ->
 * overlap-add, one kernel (istftnet.py:798-806,497-523; mlx_audio/utils.py:104-158).  x: bf16 [B][Tf_rows][ldx], 128 channels.
 * w_frag: conv_post in the kernel's fragment order, made by kk_op_pack_head_w from bf16 W[tap 7][cout 22][cin 128] (both DEVICE pointers,
 * w_frag holds 7 * 8 * 64 * 8 bf16).  cp_out (nullable): the conv_post tensor, bf16 [B][Tf_rows][cp_ld >= 22]. */
int kk_op_pack_head_w(void* stream, const void* w_bf16, void* w_frag);
int kk_op_conv_post_istft(void* stream, int B, const void* x, int ldx, int Tf_rows, const int32_t* len_frames, const void* w_frag,
                          const float* bias, float in_slope, float* wav, void* cp_out, int cp_ld);

/* MX-fp8 linear  --  the quantised nn.Linear of the reference's 8-bit checkpoints (mx.quantized_matmul, tts/utils.py:255-260).
 * kk_mxfp8_pack_weight (HOST to HOST): fp32 [N][K] -> e4m3 fragments + E8M0 scale bytes in MFMA fragment order, buffer sizes from
 * kk_mxfp8_bytes(N, K, ..).  kk_op_linear_mxfp8: x bf16 [M][ldx] (M = items * rows_per_item; len[item] valid rows, NULL = all) is
 * quantised into the caller's aq / as scratch (sizes kk_mxfp8_bytes(M, K, ..)), then out[m][n] = act(sum_k x[m][k] w[n][k] + bias[n])
 * in bf16, zero for rows past len.  act: 0 none, 2 exact GELU. */
int kk_mxfp8_bytes(int rows, int K, size_t* q_bytes, size_t* s_bytes);
int kk_mxfp8_pack_weight(const float* w_host, int N, int K, int group, uint8_t* q_host, uint8_t* s_host);
int kk_op_linear_mxfp8(void* stream, const void* x_bf16, int ldx, int M, int rows_per_item, const int32_t* len, int K, const void* wq,
                       const void* ws, int N, const float* bias, int act, void* aq, void* as, void* out_bf16, int ldo);

/* ---- debug hooks (tests only; not thread safe) ----
 * Named intermediates of the last kk_forward*: "bert_dur" "d" "t_en" "en" "asr" "F0_pred" "N_pred" "dec_out"
 * "har_source" "har" "gen_pre_res0" "gen_stage0" "gen_pre_res1" "gen_stage1" "conv_post".
 * Data format: dense float32 [B][rows][C] on the device. */
int kk_debug_info(kk_context* cx, const char* name, int64_t* rows, int64_t* channels);
int kk_debug_fetch(kk_context* cx, void* stream, const char* name, float* dst);
int kk_debug_override(kk_context* cx, const char* name, const float* src); /* src must stay valid until kk_debug_clear */
void kk_debug_clear(kk_context* cx);
/* variant 4 of the bf16 conv kernel reads its weights in MFMA fragment order: kk_op_pack_w_frag re-lays a [Kw][CoutP][CinP] bf16
 * pack out (device to device, same size); kk_debug_set_op_wfrag(wf) makes the kk_op_conv1d_bf16* calls that follow run variant 4
 * with that pack (NULL = back to the LDS-staged kernel).  The model packs both layouts in kk_finalize. */
int kk_op_pack_w_frag(void* stream, const void* w_bf16, void* w_frag, int Kw, int CoutP, int CinP);
void kk_debug_set_op_wfrag(const void* w_frag);
void kk_debug_set_op_variant(int v); /* 4 (default) or 5: which fragment-order kernel those calls use (5 = wave-specialised persistent, kk_conv_mfma5.hip) */
void kk_debug_force_generic(kk_context* cx, int flags); /* A/B tests in bf16 mode: bit0 no MFMA kernel, bit1 MFMA without norm fusion, bit2 LDS-staged MFMA kernel instead of variant 4, bit3 quantised model without the fp8 kernel, bit4 also materialise tensors fused kernels skip, bit5 stand-alone conv_post + iSTFT kernels, bit6 variant-5 (wave-specialised persistent) conv kernel wherever eligible, bit7 never (default: the layers with >= 9 taps), bit8 no side stream (the TextEncoder / harmonic-source branches of a forward on the caller's stream too) */

/* ---- per-kernel-class timing (bench.py): HIP events around every launch on the forward's stream ----
 * classes: 0 conv_generic 1 conv_mfma 2 instnorm_stats 3 adain_act 4 lstm 5 istft_head 6 layernorm 7 attention
 *          8 source 9 stft 10 linear_mxfp8.  kk_profile_end returns summed milliseconds, algorithmic flops / bytes and launch counts. */
int kk_profile_begin(kk_context* cx, int max_launches);
int kk_profile_end(kk_context* cx, int ncls, double* ms, double* flops, double* bytes, int64_t* count);

/* =====================================================================================================================
 * Mimi codec, decode path (CSM row C4): Mimi.decode, mlx_audio/codec/models/mimi/mimi.py:147-154.
 * Same conventions as above: device pointers, caller-owned buffers, work enqueued on `stream`, 0 = OK.
 * ===================================================================================================================== */
typedef struct kk_mimi kk_mimi;

/* mimi_202407 (mimi.py:41-101): dim 512, nq 32, bins 2048, qdim 256, 8 heads, 8 layers, ff 2048, nfilters 64,
 * ratios {8,6,5,4}, ksize 7, residual_ksize 3, last_ksize 3, upsample_stride 2, compress 2, rope_base 10000 */
typedef struct kk_mimi_config {
  int32_t dim, nq, bins, qdim, num_heads, num_layers, dim_feedforward, nfilters;
  int32_t n_ratios, ratios[8];
  int32_t ksize, residual_ksize, last_ksize, upsample_stride, compress;
  float rope_base;
  int32_t compute_dtype; /* KK_F32: parity path (generic fp32 kernels); KK_BF16: bf16 activations, variant-4 MFMA convolutions */
} kk_mimi_config;

int kk_mimi_create(const kk_mimi_config* cfg, kk_mimi** out);
void kk_mimi_destroy(kk_mimi* m);
/* parameters by their MLX-side names (after Mimi.load_pytorch_weights' remap, mimi.py:184-249), host fp32, MLX layouts:
 * conv / conv-transpose weights [O][K][I], code books as embedding_sum [bins][qdim] + cluster_usage [bins] */
int kk_mimi_load_tensor(kk_mimi* m, const char* name, const int64_t* shape, int ndim, const float* data);
int kk_mimi_finalize(kk_mimi* m, void* stream); /* code books divided by max(usage, 1e-5) (quantization.py:25-28), LayerScale folded, upload */
int64_t kk_mimi_samples_per_frame(const kk_mimi* m); /* 1920 for mimi_202407 */
size_t kk_mimi_workspace_bytes(kk_mimi* m, int B, int Nf);
/* codes [B][nq][Nf] int32 (device) -> pcm [B][samples_per_frame * Nf] float32 (device).  Like the reference's non-streaming
 * decode the transformer attends over the WHOLE sequence (no mask reaches the attention call, transformer.py:171). */
int kk_mimi_decode(kk_mimi* m, void* stream, int B, int Nf, const int32_t* codes, void* workspace, size_t workspace_bytes, float* pcm_out);
/* Mimi.encode (mimi.py:138-145): pcm [B][N] float32 -> codes [B][nq][kk_mimi_encode_frames(N)] int32.  Needs the
 * encoder.*, encoder_transformer.*, downsample.*, quantizer.*.input_proj parameters; always runs on the fp32 kernels (the
 * code-book search is an argmin). */
int kk_mimi_encode_frames(const kk_mimi* m, int N); /* ceil chain over the ratios and the resampler: 120000 -> 63 */
size_t kk_mimi_encode_workspace_bytes(kk_mimi* m, int B, int N);
int kk_mimi_encode(kk_mimi* m, void* stream, int B, int N, const float* pcm, void* workspace, size_t workspace_bytes, int32_t* codes_out);
/* Streaming: Mimi.decode_step / Mimi.encode_step / MimiStreamingDecoder (mimi.py:156-168,264-306; conv.py:265-351; seanet.py:219-223,277-283).
 * A stream owns the state the reference's streaming modules own, in device memory: per causal convolution the last k - stride input rows
 * (StreamableConv1d._prev_xs), per transposed convolution (k = 2 stride) the previous input row -- one row that determines the stride rows
 * of partial sums StreamableConvTranspose1d._prev_ys keeps --, the resampler's previous frame, and the KV caches of the transformer (last
 * 250 cached positions + the step's own, no mask: transformer.py:79-104).  Each step runs every layer over the rows of `chunk_frames` code
 * frames only.  decode: codes [B][nq][chunk] int32 -> pcm [B][chunk * samples_per_frame]; encode: the reverse.  fp32 kernels; B is fixed
 * between resets; at most max_frames frames per reset; a stream is created for one direction. */
typedef struct kk_mimi_stream kk_mimi_stream;
int kk_mimi_stream_create(kk_mimi* m, int max_batch, int max_frames, kk_mimi_stream** out); /* decode, one frame per step */
int kk_mimi_stream_create_chunked(kk_mimi* m, int encoder, int max_batch, int max_frames, int chunk_frames, kk_mimi_stream** out);
void kk_mimi_stream_destroy(kk_mimi_stream* s);
int kk_mimi_stream_reset(kk_mimi_stream* s); /* MimiStreamingDecoder.reset / Mimi.reset_state (mimi.py:131-137) */
int kk_mimi_stream_frames(const kk_mimi_stream* s);
int kk_mimi_stream_chunk_frames(const kk_mimi_stream* s);
/* the reference's step functions accept any number of frames per call and continue the state (mimi.py:156-168): frames per step of the
   following calls, 1 .. the chunk_frames the stream was created with */
int kk_mimi_stream_set_chunk(kk_mimi_stream* s, int chunk_frames);
int kk_mimi_stream_max_chunk_frames(const kk_mimi_stream* s);
int kk_mimi_stream_set_context(kk_mimi_stream* s, int context); /* TransformerConfig.context, default 250 (mimi.py:55-77); fresh / reset stream only */
size_t kk_mimi_stream_workspace_bytes(kk_mimi_stream* s, int B);
int kk_mimi_decode_step(kk_mimi_stream* s, void* stream, int B, const int32_t* codes, void* workspace, size_t workspace_bytes, float* pcm_out);
int kk_mimi_encode_step(kk_mimi_stream* s, void* stream, int B, const float* pcm, void* workspace, size_t workspace_bytes, int32_t* codes_out); /* Mimi.encode_step (mimi.py:156-161) */
/* intermediates of the last decode / encode (tests): "quantized", "upsampled", "transformer", "layer0".."layer3" (decode), "seanet", "transformer", "downsampled" (encode); [B][rows][channels] fp32 */
int kk_mimi_debug_info(kk_mimi* m, const char* name, int64_t* rows, int64_t* channels);
int kk_mimi_debug_fetch(kk_mimi* m, void* stream, const char* name, float* dst);

/* =====================================================================================================================
 * CSM-1B frame generator (rows C1-C3): SesameModel.generate_frame, mlx_audio/tts/models/sesame/sesame.py:349-395, with the
 * Llama stacks of mlx_lm (LlamaModel + the reference's Attention / Llama3ScaledRoPE, attention.py).  fp32 arithmetic; in bf16 weight
 * mode the single-token steps run as five launches per layer on the fused GEMV (no split-K partials, norm / SwiGLU / residual fused).
 * ===================================================================================================================== */
typedef struct kk_csm kk_csm;
typedef struct kk_llama_args { /* sesame.py:225-273 */
  int32_t num_layers, num_heads, num_kv_heads, head_dim, hidden, intermediate;
  float rope_theta, rope_factor, rms_eps; /* 500000, 32 (llama3 scaling: low 1, high 4, old context 8192), 1e-5 */
} kk_llama_args;
typedef struct kk_csm_config {
  int32_t text_vocab_size, audio_vocab_size, audio_num_codebooks, max_seq_len; /* 128256, 2051, 32, 2048 */
  kk_llama_args backbone, decoder;                                              /* llama-1B, llama-100M */
} kk_csm_config;

int kk_csm_create(const kk_csm_config* cfg, kk_csm** out);
void kk_csm_destroy(kk_csm* m);
/* MLX-side names: {backbone,decoder}.layers.N.{self_attn.{q,k,v,o}_proj,mlp.{gate,up,down}_proj,input_layernorm,
 * post_attention_layernorm}.weight, {backbone,decoder}.norm.weight, text_embeddings.weight, audio_embeddings.weight,
 * projection.weight, codebook0_head.weight, audio_head [n_cb-1][decoder_dim][audio_vocab]; host fp32 */
int kk_csm_load_tensor(kk_csm* m, const char* name, const int64_t* shape, int ndim, const float* data);
/* weight storage of the Linear layers: KK_DTYPE_F32 (default) or KK_DTYPE_BF16 -- matrices rounded to bf16 once (lossless for the
 * bf16 checkpoints load_model keeps in their own dtype, tts/utils.py:217-262), 2-byte weight stream in the single-token steps with
 * SwiGLU applied while the down projection stages its input; activations, accumulation, KV cache, logits stay fp32.  Before finalize. */
int kk_csm_set_weight_dtype(kk_csm* m, int dtype);
int kk_csm_finalize(kk_csm* m, void* stream);
/* A second generator on the SAME device weights (immutable after kk_csm_finalize): own KV caches, positions, logits and graph cache, for another
   stream / thread in flight; `m` must outlive it; kk_csm_setup_caches before its first frame.  (The reference's SesameModel owns its caches,
   sesame.py:320-333, and is single threaded.) */
int kk_csm_share(const kk_csm* m, kk_csm** out);
int kk_csm_setup_caches(kk_csm* m, int max_batch); /* SesameModel.setup_caches (sesame.py:320-333): library-owned KV caches */
int kk_csm_reset_caches(kk_csm* m);                /* sesame.py:338-345: positions restart at 0 */
int kk_csm_position(const kk_csm* m);              /* tokens in the backbone cache */
/* Ragged prompts in one batch (the reference's generate is batch 1, sesame.py:689-817): the prompts are LEFT-padded to the longest one
 * (padding frames: all-zero mask) and pad[b] (HOST array, B entries) says how many padding frames item b has.  Item b's token in cache
 * slot p then sits at position p - pad[b] and attends to slots >= pad[b] only: bit-identical to running the item alone.  Empty cache only. */
int kk_csm_set_padding(kk_csm* m, int B, const int32_t* pad_host);
size_t kk_csm_workspace_bytes(kk_csm* m, int B, int S);
/* One audio frame.  tokens [B][S][n_cb+1] int32 and tokens_mask (same shape, float 0/1) on the device; the S new positions continue
 * the backbone cache (a block of S > 1 must start an empty cache, as index_causal_mask implies, sesame.py:41-48).  Sampling:
 * temperature == 0 or uniforms == NULL -> argmax (make_sampler's rule for temp 0); otherwise inverse CDF over the top_k (<= 64)
 * logits of softmax(logit / temperature) in descending order with the injected uniforms [B][n_cb] -- the distribution of
 * make_sampler(temp, top_k) with a reproducible draw.  codes_out [B][n_cb] int32. */
int kk_csm_generate_frame(kk_csm* m, void* stream, int B, int S, const int32_t* tokens, const float* tokens_mask, float temperature, int top_k,
                          const float* uniforms, void* workspace, size_t workspace_bytes, int32_t* codes_out);
/* graph replay of the single-token frame step: the third call with identical pointers / B / sampler settings and every later one is ONE
 * hipGraphLaunch (the backbone position is a device counter, so the captured step is position-independent); results are unchanged */
int kk_csm_set_graph_mode(kk_csm* m, int on);
int kk_csm_debug_logits(kk_csm* m, void* stream, int B, float* dst); /* logits of the last frame, [n_cb][B][audio_vocab] */
/* TIMING ONLY (results are wrong while a bit is set): kernel classes of the single-token step that are not launched; process-wide.
   bit 0 q|k|v, 1 attention, 2 o, 3 gate|up, 4 down, 5 split-K combine, 6 heads, 7 sampler, 8 projection (tools/csm_skip_sweep.py) */
int kk_csm_debug_skip(int mask);
/* In-kernel wall-clock marks (100 MHz counter) of the single-token step's instrumented kernels, 8 uint64 per launch in launch order:
   class id, earliest workgroup start, latest workgroup end, workgroup 0 after its input loads, then workgroup 0's own start, shader-clock
   count at start, end, shader-clock count at end (core clock = cycles / wall x 100 MHz).  buf: device memory for `capacity`
   launches, the caller fills words 1 with ~0 and 2 with 0 before a frame; NULL switches the marks off.  Process-wide, debugging only
   (tools/csm_timeline.py). */
int kk_csm_debug_timestamps(unsigned long long* buf, int capacity);
/* the sampler of generate_frame on its own (mlx_lm make_sampler(temp, top_k), sesame.py:335-336,719): logits [B][V] -> codes [B];
 * uniforms [B] or NULL (argmax).  Radix select of the top_k set + one-wave sort; V <= 8192. */
int kk_op_csm_sample(void* stream, int B, int V, const float* logits, float temperature, int top_k, const float* uniforms, int32_t* codes_out);

#ifdef __cplusplus
}
#endif
#endif /* KOKORO_HIP_H */
