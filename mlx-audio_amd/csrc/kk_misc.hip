// Small glue kernels of the Kokoro forward (all memory-bound, coalesced along channels).
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

// out[b][o] = sum_j wT[j][o] * s[b][soff + j] + bias[o]  -- every AdaIN / AdaLayerNorm `fc` (istftnet.py:334,
// modules.py:80) of one style half evaluated in one launch; wT is [S=128][N] so lanes read consecutive o.
__global__ __launch_bounds__(256) void style_fc_kernel(const float* ref_s, int soff, const float* wT, const float* bias, float* out, int N) {
  __shared__ float s[128];
  const int b = blockIdx.y;
  if (threadIdx.x < 128) s[threadIdx.x] = ref_s[(long long)b * 256 + soff + threadIdx.x];
  __syncthreads();
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= N) return;
  float acc = 0.f;
#pragma unroll 8
  for (int j = 0; j < 128; ++j) acc = __builtin_fmaf(wT[(long long)j * N + o], s[j], acc);
  out[(long long)b * N + o] = acc + bias[o];
}

template <typename T>
__global__ __launch_bounds__(256) void fill_style_kernel(const float* ref_s, int soff, T* out, long long obs, int ldo, int coff, int S,
                                                         int Lmax, KKLen len) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)Lmax * S) return;
  const int t = (int)(e / S), j = (int)(e - (long long)t * S);
  const float v = t < kk_len(len, b) ? ref_s[(long long)b * 256 + soff + j] : 0.f;
  kk_st(out + (long long)b * obs + (long long)t * ldo + coff + j, v);
}

template <typename T>
__global__ __launch_bounds__(256) void embedding_kernel(const int* ids, const float* table, T* out, long long obs, int ldo, int C, int Tmax,
                                                        KKLen len) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)Tmax * C) return;
  const int t = (int)(e / C), c = (int)(e - (long long)t * C);
  float v = 0.f;
  if (t < kk_len(len, b)) v = table[(long long)ids[(long long)b * Tmax + t] * C + c];
  kk_st(out + (long long)b * obs + (long long)t * ldo + c, v);
}

// one wave per token: lane o < nout computes sigmoid(x . W[:, o] + bias[o]) (W transposed: [Cin][nout]); wave-sum; /speed; rint (half to even); >= 1
template <typename T>
__global__ __launch_bounds__(256) void duration_kernel(const T* x, long long xbs, int ldx, const float* W, const float* bias, int Cin,
                                                       int nout, const float* speed, int* dur, float* dur_f, int Tmax, KKLen len) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wv, b = blockIdx.y;
  if (t >= Tmax) return;
  const int L = kk_len(len, b);
  if (t >= L) {
    if (lane == 0) {
      dur[(long long)b * Tmax + t] = 0;
      if (dur_f) dur_f[(long long)b * Tmax + t] = 0.f;
    }
    return;
  }
  const T* xr = x + (long long)b * xbs + (long long)t * ldx;
  float total = 0.f;
  for (int o0 = 0; o0 < nout; o0 += 64) {
    const int o = o0 + lane;
    float sg = 0.f;
    if (o < nout) {
      float acc = 0.f;  // same summation order as before (c ascending); W is [Cin][nout]
      const float* wr = W + o;
#pragma unroll 8
      for (int c = 0; c < Cin; ++c) acc = __builtin_fmaf(kk_ld(xr + c), wr[(long long)c * nout], acc);
      acc += bias[o];
      sg = 1.0f / (1.0f + expf(-acc));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) sg += __shfl_xor(sg, s);
    total += sg;
  }
  if (lane == 0) {
    const float d = total / speed[b];
    float r = rintf(d);
    if (r < 1.0f) r = 1.0f;
    dur[(long long)b * Tmax + t] = (int)r;
    if (dur_f) dur_f[(long long)b * Tmax + t] = d;
  }
}

// one workgroup per utterance: inclusive scan of the durations, then each token writes its frame range
__global__ __launch_bounds__(512) void alignment_kernel(const int* dur, int Tmax, const int* lenT, int* frame_idx, int* lenF, int Fmax) {
  __shared__ int sc[512];
  const int b = blockIdx.x, t = threadIdx.x;
  const int L = lenT[b];
  const int d = (t < L && t < Tmax) ? max(dur[(long long)b * Tmax + t], 0) : 0;
  sc[t] = d;
  __syncthreads();
  for (int off = 1; off < 512; off <<= 1) {
    const int v = t >= off ? sc[t - off] : 0;
    __syncthreads();
    sc[t] += v;
    __syncthreads();
  }
  const int end = sc[t], start = end - d;
  const int total = min(sc[511], Fmax);
  for (int f = start; f < end && f < Fmax; ++f) frame_idx[(long long)b * Fmax + f] = t;
  for (int f = total + t; f < Fmax; f += 512) frame_idx[(long long)b * Fmax + f] = 0;
  if (t == 0) lenF[b] = total;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* in, long long ibs, int ldi, const int* frame_idx, int Fmax, const int* lenF,
                                                          T* out, long long obs, int ldo, int coff, int C) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)Fmax * C) return;
  const int f = (int)(e / C), c = (int)(e - (long long)f * C);
  T v = (T)0.f;
  if (f < lenF[b]) v = in[(long long)b * ibs + (long long)frame_idx[(long long)b * Fmax + f] * ldi + c];
  out[(long long)b * obs + (long long)f * ldo + coff + c] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void copy_slice_kernel(const T* in, long long ibs, int ldi, T* out, long long obs, int ldo, int coff, int C,
                                                         int Lmax, KKLen len) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)Lmax * C) return;
  const int l = (int)(e / C), c = (int)(e - (long long)l * C);
  T v = (T)0.f;
  if (l < kk_len(len, b)) v = in[(long long)b * ibs + (long long)l * ldi + c];
  out[(long long)b * obs + (long long)l * ldo + coff + c] = v;
}

__global__ void lens_kernel(const int* lenF, int* o, int B, int up0, int up1, int hop) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int F = lenF[b];
  o[b] = 2 * F;
  o[B + b] = 2 * F * up0;
  o[2 * B + b] = F > 0 ? 2 * F * up0 * up1 + 1 : 0;
  o[3 * B + b] = 2 * F * up0 * up1 * hop;
}

}  // namespace

#define KK_DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
  do {                                            \
    if ((dtype) == KK_F32) { CALL_F32; }          \
    else { CALL_BF16; }                           \
  } while (0)

int kk_launch_style_fc(const float* ref_s, int soff, const float* wT, const float* bias, float* out, int N, int B, hipStream_t st) {
  if (B <= 0 || N <= 0) return 0;
  hipLaunchKernelGGL(style_fc_kernel, dim3(kk_cdiv(N, 256), B), dim3(256), 0, st, ref_s, soff, wT, bias, out, N);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_fill_style(const float* ref_s, int soff, void* out, long long obs, int ldo, int coff, int S, int Lmax, KKLen len, int B,
                         int dtype, hipStream_t st) {
  if (B <= 0 || Lmax <= 0) return 0;
  dim3 grid((unsigned)(((long long)Lmax * S + 255) / 256), B);
  KK_DISPATCH_T(dtype,
                hipLaunchKernelGGL(fill_style_kernel<float>, grid, dim3(256), 0, st, ref_s, soff, (float*)out, obs, ldo, coff, S, Lmax, len),
                hipLaunchKernelGGL(fill_style_kernel<bf16_t>, grid, dim3(256), 0, st, ref_s, soff, (bf16_t*)out, obs, ldo, coff, S, Lmax, len));
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_embedding(const int* ids, const float* table, void* out, long long obs, int ldo, int C, int Tmax, KKLen len, int B, int dtype,
                        hipStream_t st) {
  if (B <= 0 || Tmax <= 0) return 0;
  dim3 grid((unsigned)(((long long)Tmax * C + 255) / 256), B);
  KK_DISPATCH_T(dtype, hipLaunchKernelGGL(embedding_kernel<float>, grid, dim3(256), 0, st, ids, table, (float*)out, obs, ldo, C, Tmax, len),
                hipLaunchKernelGGL(embedding_kernel<bf16_t>, grid, dim3(256), 0, st, ids, table, (bf16_t*)out, obs, ldo, C, Tmax, len));
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_duration(const void* x, long long xbs, int ldx, const float* W, const float* bias, int Cin, int nout, const float* speed,
                       int* dur, float* dur_f, int Tmax, KKLen len, int B, int dtype, hipStream_t st) {
  if (B <= 0 || Tmax <= 0) return 0;
  dim3 grid(kk_cdiv(Tmax, 4), B);
  KK_DISPATCH_T(dtype,
                hipLaunchKernelGGL(duration_kernel<float>, grid, dim3(256), 0, st, (const float*)x, xbs, ldx, W, bias, Cin, nout, speed, dur,
                                   dur_f, Tmax, len),
                hipLaunchKernelGGL(duration_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, xbs, ldx, W, bias, Cin, nout, speed, dur,
                                   dur_f, Tmax, len));
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_alignment(const int* dur, int Tmax, const int* lenT, int* frame_idx, int* lenF, int Fmax, int B, hipStream_t st) {
  if (B <= 0) return 0;
  if (Tmax > 512) return kk_fail("alignment: Tmax > 512 (kokoro.py:131-134 caps the context at 512 tokens)");
  hipLaunchKernelGGL(alignment_kernel, dim3(B), dim3(512), 0, st, dur, Tmax, lenT, frame_idx, lenF, Fmax);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_gather_rows(const void* in, long long ibs, int ldi, const int* frame_idx, int Fmax, const int* lenF, void* out, long long obs,
                          int ldo, int coff, int C, int B, int dtype, hipStream_t st) {
  if (B <= 0 || Fmax <= 0) return 0;
  dim3 grid((unsigned)(((long long)Fmax * C + 255) / 256), B);
  KK_DISPATCH_T(dtype,
                hipLaunchKernelGGL(gather_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)in, ibs, ldi, frame_idx, Fmax, lenF,
                                   (float*)out, obs, ldo, coff, C),
                hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)in, ibs, ldi, frame_idx, Fmax, lenF,
                                   (bf16_t*)out, obs, ldo, coff, C));
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_copy_slice(const void* in, long long ibs, int ldi, void* out, long long obs, int ldo, int coff, int C, int Lmax, KKLen len,
                         int B, int dtype, hipStream_t st) {
  if (B <= 0 || Lmax <= 0) return 0;
  dim3 grid((unsigned)(((long long)Lmax * C + 255) / 256), B);
  KK_DISPATCH_T(dtype,
                hipLaunchKernelGGL(copy_slice_kernel<float>, grid, dim3(256), 0, st, (const float*)in, ibs, ldi, (float*)out, obs, ldo, coff, C,
                                   Lmax, len),
                hipLaunchKernelGGL(copy_slice_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)in, ibs, ldi, (bf16_t*)out, obs, ldo, coff,
                                   C, Lmax, len));
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_lens(const int* lenF, int* lens_out, int B, int up0, int up1, int hop, hipStream_t st) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(lens_kernel, dim3(kk_cdiv(B, 64)), dim3(64), 0, st, lenF, lens_out, B, up0, up1, hop);
  KK_CHECK_LAUNCH();
  return 0;
}

// ---- dtype-converting strided copy (debug fetch / override only)
namespace {
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS* src, long long sbs, int lds, TD* dst, long long dbs, int ldd, int C, int rows) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)rows * C) return;
  const int r = (int)(e / C), c = (int)(e - (long long)r * C);
  dst[(long long)b * dbs + (long long)r * ldd + c] = (TD)(float)src[(long long)b * sbs + (long long)r * lds + c];
}
}  // namespace

namespace {
__global__ void set_u64_kernel(unsigned long long* dst, unsigned long long v) { *dst = v; }
}  // namespace
int kk_launch_set_u64(unsigned long long* dst, unsigned long long v, hipStream_t st) {
  hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, st, dst, v);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_convert(const void* src, int sdt, long long sbs, int lds, void* dst, int ddt, long long dbs, int ldd, int C, int rows, int B,
                      hipStream_t st) {
  if (B <= 0 || rows <= 0 || C <= 0) return 0;
  dim3 grid((unsigned)(((long long)rows * C + 255) / 256), B);
  if (sdt == KK_F32 && ddt == KK_F32)
    hipLaunchKernelGGL((convert_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, sbs, lds, (float*)dst, dbs, ldd, C, rows);
  else if (sdt == KK_F32 && ddt == KK_BF16)
    hipLaunchKernelGGL((convert_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, sbs, lds, (bf16_t*)dst, dbs, ldd, C, rows);
  else if (sdt == KK_BF16 && ddt == KK_F32)
    hipLaunchKernelGGL((convert_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)src, sbs, lds, (float*)dst, dbs, ldd, C, rows);
  else
    return kk_fail("convert: unsupported dtype pair");
  KK_CHECK_LAUNCH();
  return 0;
}
