// Harmonic source, forward STFT(20) and the iSTFT vocoder head.
//
//  kk_launch_source  : SineGen + SourceModuleHnNSF (istftnet.py:531-680) with the 300x nearest F0
//                      up-sampling of Generator.__call__ (istftnet.py:770) folded in.
//  kk_launch_stft20  : MLXSTFT.transform (istftnet.py:463-495) + stft (utils.py:52-101).
//  kk_launch_istft_head : spec=exp / phase=sin (istftnet.py:804-805) + MLXSTFT.inverse
//                      (istftnet.py:497-523) + istft (utils.py:104-158).
//
// Float32 operation order follows the reference wherever it changes the result materially (the
// phase accumulator reaches 1e5..1e6 rad, where one float32 ulp is 0.01..0.1 rad): sequential
// cumsum, (cumsum*2)*pi*300, the un-clamped linear interpolation with its wrap to the LAST sample
// for the first 150 outputs (interpolate.py:88-106), un-fused multiply/add in the blend.
#include <stdlib.h>

#include "kk_common.h"
#include "kk_kernels.h"
#include "kk_istft_math.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

struct Tables {
  float hann_sym[20];  // utils.py:10-14, forward STFT window
  float hann_per[20];  // hanning(21)[:-1], utils.py:121, inverse window
  float cs[20];        // cos(2*pi*m/20)
  float sn[20];        // sin(2*pi*m/20)
};

Tables make_tables() {
  Tables t;
  const double pi = 3.14159265358979323846;
  for (int n = 0; n < 20; ++n) {
    t.hann_sym[n] = (float)(0.5 * (1.0 - cos(2.0 * pi * n / 19.0)));
    t.hann_per[n] = (float)(0.5 * (1.0 - cos(2.0 * pi * n / 20.0)));
    t.cs[n] = (float)cos(2.0 * pi * n / 20.0);
    t.sn[n] = (float)sin(2.0 * pi * n / 20.0);
  }
  // exact values where they are exact
  t.cs[0] = 1.f; t.cs[5] = 0.f; t.cs[10] = -1.f; t.cs[15] = 0.f;
  t.sn[0] = 0.f; t.sn[5] = 1.f; t.sn[10] = 0.f; t.sn[15] = -1.f;
  return t;
}

// ------------------------------------------------------------------ phase accumulator
// phase[b][h][i] = ((cumsum_i(rad) * 2) * pi) * up, rad = python_mod(f0*h/24000, 1)   (istftnet.py:561,573-575)
__global__ __launch_bounds__(64) void source_phase_kernel(KKSourceArgs a) {
  __shared__ float f0s[1024];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int L2 = a.len2 ? a.len2[b] : a.L2max;
  const float hmul = (float)(lane + 1);
  float cum = 0.f;
  float* ph = a.phase + ((long long)b * 9 + lane) * a.L2max;
  for (int i0 = 0; i0 < L2; i0 += 1024) {
    const int n = min(1024, L2 - i0);
    __syncthreads();
    for (int i = lane; i < n; i += 64) f0s[i] = a.f0[(long long)b * a.L2max + i0 + i];
    __syncthreads();
    if (lane < 9) {
      for (int i = 0; i < n; ++i) {
        const float fn = f0s[i] * hmul;
        float r = fmodf(fn / 24000.0f, 1.0f);
        if (r < 0.f) r += 1.0f;  // python-style modulo (result takes the sign of the divisor)
        cum = cum + r;
        ph[i0 + i] = ((cum * 2.0f) * 3.14159265358979323846f) * (float)a.upsample;
      }
    }
  }
}

// ------------------------------------------------------------------ Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t sub, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), sub, 0x4B4B5352u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ void box_muller(uint32_t u0, uint32_t u1, float& z0, float& z1) {
  const float a = ((float)(u0 >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  const float bq = ((float)(u1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * __logf(a));
  float s, c;
  __sincosf(6.28318530717958647692f * bq, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

// ------------------------------------------------------------------ sample synthesis
__global__ __launch_bounds__(256) void source_sample_kernel(KKSourceArgs a, float xs, float xh) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= a.Nmax) return;
  const int L2 = a.len2 ? a.len2[b] : a.L2max;
  const int N = L2 * a.upsample;
  float* o = a.har_source + (long long)b * a.Nmax;
  if (n >= N) {
    o[n] = 0.f;
    return;
  }
  // interpolate1d(linear, align_corners=None), interpolate.py:80-106
  const float x = ((float)n * xs + xh) - 0.5f;
  const float xl = floorf(x);
  const int lo = (int)xl;
  const int hi = min(lo + 1, L2 - 1);
  const float fr = x - xl;
  const int lo_w = lo < 0 ? lo + L2 : lo;  // negative index wraps to the end
  const float omf = 1.0f - fr;
  const float f0 = a.f0[(long long)b * a.L2max + n / a.upsample];
  const float uv = f0 > 10.0f ? 1.0f : 0.0f;
  const float namp = uv * 0.003f + ((1.0f - uv) * 0.1f) / 3.0f;
  float nz[12];
#pragma unroll
  for (int h = 0; h < 12; ++h) nz[h] = 0.f;
  if (a.noise_mode == 1) {
    const float* np_ = a.noise + ((long long)b * a.Nmax + n) * 9;
#pragma unroll
    for (int h = 0; h < 9; ++h) nz[h] = np_[h];
  } else if (a.noise_mode == 2) {
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    const uint64_t ctr = (uint64_t)b * (uint64_t)a.Nmax + (uint64_t)n;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      uint32_t r[4];
      philox4(seed, ctr, (uint32_t)g, r);
      box_muller(r[0], r[1], nz[4 * g], nz[4 * g + 1]);
      box_muller(r[2], r[3], nz[4 * g + 2], nz[4 * g + 3]);
    }
  }
  const float* pb = a.phase + (long long)b * 9 * a.L2max;
  float acc = 0.f;
#pragma unroll
  for (int h = 0; h < 9; ++h) {
    const float pl = pb[(long long)h * a.L2max + lo_w];
    const float phh = pb[(long long)h * a.L2max + hi];
    const float t0 = pl * omf;
    const float t1 = phh * fr;
    const float ph = t0 + t1;
    const float sw = (sinf(ph) * 0.1f) * uv + namp * nz[h];
    acc += sw * a.lin_w[h];
  }
  o[n] = tanhf(acc + a.lin_b);
}

// ------------------------------------------------------------------ forward STFT (n_fft 20, hop 5)
template <typename T>
__global__ __launch_bounds__(256) void stft20_kernel(const float* hs, int Nmax, const int* lenN, T* har, long long obs, int ldo,
                                                     int Tfmax, Tables tb) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Tfmax) return;
  const int N = lenN ? lenN[b] : Nmax;
  const int Tf = N > 0 ? N / 5 + 1 : 0;
  T* o = har + (long long)b * obs + (long long)t * ldo;
  if (t >= Tf) {
#pragma unroll
    for (int k = 0; k < 22; ++k) kk_st(o + k, 0.f);
    return;
  }
  const float* x = hs + (long long)b * Nmax;
  float fr[20];
#pragma unroll
  for (int j = 0; j < 20; ++j) {
    int m = 5 * t + j - 10;
    if (m < 0) m = -m;
    if (m >= N) m = 2 * (N - 1) - m;
    fr[j] = x[m] * tb.hann_sym[j];
  }
#pragma unroll
  for (int k = 0; k <= 10; ++k) {
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const int m = (k * j) % 20;
      re = __builtin_fmaf(fr[j], tb.cs[m], re);
      im = __builtin_fmaf(-fr[j], tb.sn[m], im);
    }
    if (k == 0 || k == 10) im = 0.f;  // real FFT: DC / Nyquist carry a +0 imaginary part
    kk_st(o + k, sqrtf(re * re + im * im));
    kk_st(o + 11 + k, atan2f(im, re));
  }
}

// ------------------------------------------------------------------ iSTFT head
// HBM-bound by design: 22 inputs and 5 fp32 outputs per frame column (64 B with bf16 input).
//   phase 0  the workgroup's [255 frames][ldx] input tile is ONE contiguous span: coalesced 16-byte loads into LDS
//   phase 1  one thread per frame: exp / sin / sincos, then the 20-point inverse real DFT using the o <-> 20-o symmetry
//            (cos terms even, sin terms odd: 11 x 18 FMAs instead of 20 x 18), x periodic Hann -> LDS
//   phase 2  one thread per hop block: <= 4 overlapping frames added in ascending frame order (the reference's
//            scatter-add order, utils.py:138-147), divided by the window sum accumulated in the same order
//   phase 3  the 1260 output samples of the workgroup leave as coalesced 8-byte stores
// 252 hop blocks + 3 halo frames = 255 frames: ONE frame per thread in phase 1 (with 256 + 3 the second trip of the frame loop
// ran a full pass for three frames and doubled the kernel's VALU time); 252 * 5 samples keeps every workgroup's first sample even
constexpr int IH_FR = 252;  // hop-blocks per workgroup
constexpr int IH_NF = IH_FR + 3;

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void istft_head_kernel(const T* x, long long xbs, int ldx, const int* len_frames, int Tfmax, float* wav,
                                                         long long wbs, int stage_off, Tables tb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ism[];
  // LDS: [ input tile [255][ldx]  ALIASED WITH  ys [255][21] fp32 ] [ stage [252][5] fp32 ] -- the tile is dead once every
  // thread holds its frame's 22 spectrum values in registers (barrier), so 26.5 KB (bf16 input) per workgroup instead of 34 KB:
  // 6 workgroups per CU instead of 4 to hide the tile-load latency
  float* ys = (float*)ism;   // [255][21]
  unsigned char* tile = ism;  // [255][ldx]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int Tf = len_frames ? len_frames[b] : Tfmax;
  const int g0 = blockIdx.x * IH_FR;  // first hop-block of this workgroup
  const int f0 = g0 - 3;              // first frame of the tile
  const T* xb = x + (long long)b * xbs;
  const int row_bytes = ldx * (int)sizeof(T);
  // ---- phase 0: coalesced tile load (frames clamped into [0, Tfmax): out-of-range frames are masked in phase 1)
  {
    const int fa = f0 < 0 ? 0 : f0;
    const int fb = min(f0 + IH_NF, Tfmax);
    const long long bytes = (long long)(fb - fa) * row_bytes;
    const unsigned char* src = (const unsigned char*)(xb + (long long)fa * ldx);
    unsigned char* dst = tile + (long long)(fa - f0) * row_bytes;
    if (bytes > 0) {
      if ((((uintptr_t)src) & 15) == 0 && (row_bytes & 15) == 0) {
        for (long long o = (long long)tid * 16; o < bytes; o += 256 * 16) *(uint4*)(dst + o) = *(const uint4*)(src + o);
      } else {
        for (long long o = (long long)tid * sizeof(T); o < bytes; o += 256 * sizeof(T)) *(T*)(dst + o) = *(const T*)(src + o);
      }
    }
  }
  __syncthreads();
  // ---- phase 1: windowed inverse real DFT of frames f0 .. f0+254
  {
    const int i = tid;  // IH_NF = 255 frames <= 256 threads
    const int f = f0 + i;
    float* yo = ys + i * 21;
    const bool fv = i < IH_NF && f >= 0 && f < Tf;
    float re[11], im[11];
    if (fv) {
      const T* xr = (const T*)(tile + (long long)i * row_bytes);
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float lm = kk_ld(xr + k), pr = kk_ld(xr + 11 + k);
        const float mag = FAST ? __expf(lm) : expf(lm);
        const float ph = FAST ? __sinf(pr) : sinf(pr);
        float s, c;
        if (FAST) __sincosf(ph, &s, &c); else sincosf(ph, &s, &c);
        re[k] = mag * c;
        im[k] = mag * s;
      }
    }
    __syncthreads();  // every thread has read its row: ys may now overwrite the tile
    if (fv) {
      // x[o] = (re0 + (-1)^o re10 + 2 (C[o] - S[o])) / 20,  x[20-o] = (.. + 2 (C[o] + S[o])) / 20  with
      //   C[o] = sum_k re[k] cos(2 pi k o / 20),  S[o] = sum_k im[k] sin(2 pi k o / 20),  k = 1..9.
      // Even / odd k split: cos(2 pi k (10-o)/20) = (-1)^k cos(..o..), sin(2 pi k (10-o)/20) = -(-1)^k sin(..o..), so
      // o and 10-o share their partial sums: 6 x 18 FMAs instead of 11 x 18.
#pragma unroll
      for (int o = 0; o <= 5; ++o) {
        float Ce = 0.f, Co = 0.f, Se = 0.f, So = 0.f;
#pragma unroll
        for (int k = 1; k < 10; ++k) {
          const int m = (k * o) % 20;
          if (k & 1) {
            Co = __builtin_fmaf(re[k], tb.cs[m], Co);
            So = __builtin_fmaf(im[k], tb.sn[m], So);
          } else {
            Ce = __builtin_fmaf(re[k], tb.cs[m], Ce);
            Se = __builtin_fmaf(im[k], tb.sn[m], Se);
          }
        }
        {
          const float C = Ce + Co, S = Se + So;
          const float dc = re[0] + ((o & 1) ? -re[10] : re[10]);
          yo[o] = (dc + 2.0f * (C - S)) * 0.05f * tb.hann_per[o];
          if (o > 0) yo[20 - o] = (dc + 2.0f * (C + S)) * 0.05f * tb.hann_per[20 - o];
        }
        if (o < 5) {
          const int p = 10 - o;
          const float C = Ce - Co, S = So - Se;
          const float dc = re[0] + ((p & 1) ? -re[10] : re[10]);
          yo[p] = (dc + 2.0f * (C - S)) * 0.05f * tb.hann_per[p];
          if (p < 10) yo[20 - p] = (dc + 2.0f * (C + S)) * 0.05f * tb.hann_per[20 - p];
        }
      }
    } else if (i < IH_NF) {
#pragma unroll
      for (int o = 0; o < 20; ++o) yo[o] = 0.f;
    }
  }
  __syncthreads();
  // ---- phase 2: overlap-add in ascending frame order, normalise by the window sum, trim 10 | 10
  float* stage = (float*)(ism + stage_off);  // [252][5]
  const int g = g0 + tid;
  const int nout = Tf > 0 ? 5 * (Tf - 1) : 0;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    if (tid >= IH_FR) break;
    const int n = 5 * g + r - 10;
    float v = 0.f;
    if (n >= 0 && n < nout) {
      float acc = 0.f, ws = 0.f;
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        const int f = g - j;
        if (f >= 0 && f < Tf) {
          acc += ys[(tid + 3 - j) * 21 + 5 * j + r];
          ws += tb.hann_per[5 * j + r];
        }
      }
      v = ws != 0.f ? (FAST ? acc * __builtin_amdgcn_rcpf(ws) : acc / ws) : acc;
    }
    stage[tid * 5 + r] = v;
  }
  __syncthreads();
  // ---- phase 3: samples n0 .. n0+1259 (n0 = 5*g0 - 10, even) as float2 stores
  float* wb = wav + (long long)b * wbs;
  const long long n0 = 5LL * g0 - 10;
  const long long ntot = 5LL * (Tfmax - 1);
  for (int e = tid; e < IH_FR * 5 / 2; e += 256) {
    const long long n = n0 + 2 * e;
    const float2 v = *(const float2*)(stage + 2 * e);
    if (n >= 0 && n + 1 < ntot && ((((uintptr_t)(wb + n)) & 7) == 0)) {
      *(float2*)(wb + n) = v;
    } else {
      if (n >= 0 && n < ntot) wb[n] = v.x;
      if (n + 1 >= 0 && n + 1 < ntot) wb[n + 1] = v.y;
    }
  }
}

// ------------------------------------------------------------------ iSTFT head, wave-local form (no LDS, no barriers)
// Lane l of a wave owns frame f = gw0 - 3 + l; the <= 4 frames that overlap hop block g = f live in lanes l-3 .. l, so the
// overlap-add is 15 __shfl_up and lanes 3..63 each produce the 5 samples of their hop block (61 hop blocks per wave; the 3
// halo frames per wave are recomputed, 5 %).  Occupancy is then bounded by VGPRs alone (8 waves per SIMD), which is what this
// transcendental-bound kernel needs to hide its input latency.  Same arithmetic and the same ascending-frame summation order
// as the tiled kernel above (the reference's scatter-add order, utils.py:138-147).
constexpr int IW_HB = 61;  // hop blocks per wave

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void istft_head_wave_kernel(const T* x, long long xbs, int ldx, const int* len_frames, int Tfmax, float* wav,
                                                              long long wbs, Tables tb) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int Tf = len_frames ? len_frames[b] : Tfmax;
  const int g0 = (blockIdx.x * 4 + wv) * IW_HB;  // first hop block of this wave
  if (g0 >= Tfmax + 3) return;                   // whole wave (no barriers in this kernel)
  const int f = g0 - 3 + lane;
  const bool fv = f >= 0 && f < Tf;
  float y[20];
  if (fv) {
    const T* xr = x + (long long)b * xbs + (long long)f * ldx;
    float in[22];
    if (sizeof(T) == 2 && (ldx & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)xbs * 2)) & 15) == 0) {
      // 22 bf16 = 44 bytes: three 16-byte loads (the row pitch is >= 24 elements)
      union { uint4 u[3]; bf16_t h[24]; } r;
#pragma unroll
      for (int q = 0; q < 3; ++q) r.u[q] = *((const uint4*)xr + q);
#pragma unroll
      for (int k = 0; k < 22; ++k) in[k] = (float)r.h[k];
    } else {
#pragma unroll
      for (int k = 0; k < 22; ++k) in[k] = kk_ld(xr + k);
    }
    float re[11], im[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float lm = in[k], pr = in[11 + k];
      const float mag = FAST ? __expf(lm) : expf(lm);
      const float ph = FAST ? __sinf(pr) : sinf(pr);
      float s, c;
      if (FAST) __sincosf(ph, &s, &c); else sincosf(ph, &s, &c);
      re[k] = mag * c;
      im[k] = mag * s;
    }
#pragma unroll
    for (int o = 0; o <= 5; ++o) {
      float Ce = 0.f, Co = 0.f, Se = 0.f, So = 0.f;
#pragma unroll
      for (int k = 1; k < 10; ++k) {
        const int m = (k * o) % 20;
        if (k & 1) {
          Co = __builtin_fmaf(re[k], tb.cs[m], Co);
          So = __builtin_fmaf(im[k], tb.sn[m], So);
        } else {
          Ce = __builtin_fmaf(re[k], tb.cs[m], Ce);
          Se = __builtin_fmaf(im[k], tb.sn[m], Se);
        }
      }
      {
        const float C = Ce + Co, S = Se + So;
        const float dc = re[0] + ((o & 1) ? -re[10] : re[10]);
        y[o] = (dc + 2.0f * (C - S)) * 0.05f * tb.hann_per[o];
        if (o > 0) y[20 - o] = (dc + 2.0f * (C + S)) * 0.05f * tb.hann_per[20 - o];
      }
      if (o < 5) {
        const int p = 10 - o;
        const float C = Ce - Co, S = So - Se;
        const float dc = re[0] + ((p & 1) ? -re[10] : re[10]);
        y[p] = (dc + 2.0f * (C - S)) * 0.05f * tb.hann_per[p];
        if (p < 10) y[20 - p] = (dc + 2.0f * (C + S)) * 0.05f * tb.hann_per[20 - p];
      }
    }
  } else {
#pragma unroll
    for (int o = 0; o < 20; ++o) y[o] = 0.f;
  }
  // overlap-add: sample 5 g + r - 10 = sum over j = 3..0 of frame (g - j)'s sample 5 j + r, ascending frame order
  const int g = f;
  const int nout = Tf > 0 ? 5 * (Tf - 1) : 0;
  const long long ntot = 5LL * (Tfmax - 1);
  float* wb = wav + (long long)b * wbs;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    float acc = 0.f, ws = 0.f;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const float v = j ? __shfl_up(y[5 * j + r], j) : y[r];  // lanes < j read their own value: they produce no output
      const int fj = g - j;
      if (fj >= 0 && fj < Tf) {
        acc += v;
        ws += tb.hann_per[5 * j + r];
      }
    }
    const long long n = 5LL * g + r - 10;
    if (lane >= 3 && g < g0 + IW_HB && n >= 0 && n < ntot) {
      float v = 0.f;
      if (n < nout) v = ws != 0.f ? (FAST ? acc * __builtin_amdgcn_rcpf(ws) : acc / ws) : acc;
      wb[n] = v;
    }
  }
}


// ------------------------------------------------------------------ iSTFT head, wave-local FAST form (bf16 mode)
// The same lane-per-frame structure as istft_head_wave_kernel<T, true>, with everything that does not depend on the data folded
// into literals: twiddles carry the factor 2 of the real inverse DFT (exact), the output scale is 0.5 * 0.05 * hann_per[o] -- the
// interior window sum of a periodic Hann at hop N/4 is exactly 2.0f in float32 for every r in this summation order, so interior
// samples need no division at all -- and invalid frames contribute exact zeros, so the overlap-add is four unconditional adds in
// ascending frame order.  Only waves that touch an utterance edge (first 3 hop blocks, the block after the last frame) take the
// branch that rebuilds the partial window sum.  472 -> ~330 VALU instructions per wave of 61 hop blocks; sin / cos of the phase's sine (|.| <= 1) are a
// packed polynomial; what remains quarter-rate is exp and the first sin: 22 transcendentals per frame.
template <typename T>
__global__ __launch_bounds__(256) void istft_head_wave_fast_kernel(const T* x, long long xbs, int ldx, const int* len_frames, int Tfmax, float* wav,
                                                                   long long wbs, Tables tb) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int Tf = len_frames ? len_frames[b] : Tfmax;
  const int g0 = (blockIdx.x * 4 + wv) * IW_HB;
  if (g0 >= Tfmax + 3) return;
  const int f = g0 - 3 + lane;
  const bool fv = f >= 0 && f < Tf;
  float y[20];
  {
    // every lane runs the arithmetic (no divergence, no zero-initialised outputs): frames outside [0, Tf) read a clamped row -- finite
    // data: rows past an utterance's length hold zeros -- and their magnitudes are multiplied by 0, which makes all 20 samples exact zeros
    const int fc = min(max(f, 0), Tfmax - 1);
    const float fmask = fv ? 1.0f : 0.0f;
    const T* xr = x + (long long)b * xbs + (long long)fc * ldx;
    float in[22];
    if (sizeof(T) == 2 && (ldx & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)xbs * 2)) & 15) == 0) {
      uint4 r[3];  // 22 bf16 = 44 bytes: three 16-byte loads (the row pitch is >= 24 elements)
#pragma unroll
      for (int q = 0; q < 3; ++q) r[q] = *((const uint4*)xr + q);
      const unsigned w[12] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w, r[2].x, r[2].y, r[2].z, r[2].w};
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        in[2 * k] = __uint_as_float(w[k] << 16);
        in[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 22; ++k) in[k] = kk_ld(xr + k);
    }
    kk_istft::frame_fast(in, fmask, y);
  }
  const int g = f;
  const bool interior = g >= 3 && g < Tf;  // frames g-3 .. g all exist
  const int nout = Tf > 0 ? 5 * (Tf - 1) : 0;
  const int ntot = 5 * (Tfmax - 1);
  const bool mine = lane >= 3 && g < g0 + IW_HB;
  float* wb = wav + (long long)b * wbs;
  float v[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const float s3 = __shfl_up(y[15 + r], 3), s2 = __shfl_up(y[10 + r], 2), s1 = __shfl_up(y[5 + r], 1);
    v[r] = ((s3 + s2) + s1) + y[r];  // ascending frame order (utils.py:138-147); frames outside [0, Tf) hold exact zeros
  }
  if (__ballot(mine && !interior) != 0ull) {  // an utterance edge inside this wave: partial window sums (utils.py:143-150)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      float ws = 0.f;
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        const int fj = g - j;
        if (fj >= 0 && fj < Tf) ws += tb.hann_per[5 * j + r];
      }
      if (!interior) {
        const float a2 = v[r] + v[r];  // the literals carry the interior 1/2
        v[r] = ws != 0.f ? a2 * __builtin_amdgcn_rcpf(ws) : a2;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int n = 5 * g + r - 10;
    if (mine && n >= 0 && n < ntot) wb[n] = n < nout ? v[r] : 0.f;
  }
}

const Tables g_tables = make_tables();

}  // namespace

int kk_launch_source(const KKSourceArgs& a, int B, hipStream_t st) {
  if (B <= 0 || a.L2max <= 0) return 0;
  hipLaunchKernelGGL(source_phase_kernel, dim3(B), dim3(64), 0, st, a);
  // interpolate.py:84-86: x = arange(size) * (in_width/size) + 0.5*(in_width/size) - 0.5 ; in_width/size is the
  // python double 2F/(600F) = fl64(1/up), converted to float32 when it meets the float32 array
  const double r = 1.0 / (double)a.upsample;
  const float xs = (float)r, xh = (float)(0.5 * r);
  hipLaunchKernelGGL(source_sample_kernel, dim3(kk_cdiv(a.Nmax, 256), B), dim3(256), 0, st, a, xs, xh);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_stft20(const float* har_source, int Nmax, const int* lenN, void* har, long long obs, int ldo, int Tfmax, int B, int dtype,
                     hipStream_t st) {
  if (B <= 0 || Tfmax <= 0) return 0;
  dim3 grid(kk_cdiv(Tfmax, 256), B);
  if (dtype == KK_F32)
    hipLaunchKernelGGL(stft20_kernel<float>, grid, dim3(256), 0, st, har_source, Nmax, lenN, (float*)har, obs, ldo, Tfmax, g_tables);
  else
    hipLaunchKernelGGL(stft20_kernel<bf16_t>, grid, dim3(256), 0, st, har_source, Nmax, lenN, (bf16_t*)har, obs, ldo, Tfmax, g_tables);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_istft_head(const void* x, long long xbs, int ldx, const int* len_frames, int Tfmax, float* wav, long long wbs, int B,
                         int dtype, int fast, hipStream_t st) {
  if (B <= 0 || Tfmax <= 0) return 0;
  if (ldx < 22 || ldx > 64) return kk_fail("istft_head: input pitch must be in [22, 64]");
  static int tiled = -1;
  if (tiled < 0) tiled = getenv("KK_ISTFT_TILED") ? 1 : 0;  // A/B switch: the LDS-tiled kernel
  static int oldfast = -1;
  if (oldfast < 0) oldfast = getenv("KK_ISTFT_OLDFAST") ? 1 : 0;  // A/B switch: the generic wave kernel's FAST instantiation
  if (!tiled && fast && !oldfast && 5LL * Tfmax < 0x7fffffffLL) {
    dim3 gw(kk_cdiv(Tfmax + 3, 4 * IW_HB), B);
    if (dtype == KK_F32)
      hipLaunchKernelGGL((istft_head_wave_fast_kernel<float>), gw, dim3(256), 0, st, (const float*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, g_tables);
    else
      hipLaunchKernelGGL((istft_head_wave_fast_kernel<bf16_t>), gw, dim3(256), 0, st, (const bf16_t*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, g_tables);
    KK_CHECK_LAUNCH();
    return 0;
  }
  if (!tiled) {
    dim3 gw(kk_cdiv(Tfmax + 3, 4 * IW_HB), B);
    if (dtype == KK_F32) {
      if (fast) hipLaunchKernelGGL((istft_head_wave_kernel<float, true>), gw, dim3(256), 0, st, (const float*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, g_tables);
      else hipLaunchKernelGGL((istft_head_wave_kernel<float, false>), gw, dim3(256), 0, st, (const float*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, g_tables);
    } else {
      if (fast) hipLaunchKernelGGL((istft_head_wave_kernel<bf16_t, true>), gw, dim3(256), 0, st, (const bf16_t*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, g_tables);
      else hipLaunchKernelGGL((istft_head_wave_kernel<bf16_t, false>), gw, dim3(256), 0, st, (const bf16_t*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, g_tables);
    }
    KK_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid(kk_cdiv(Tfmax + 3, IH_FR), B);
  const size_t esz = dtype == KK_F32 ? 4 : 2;
  size_t front = (size_t)IH_NF * ldx * esz;  // input tile, aliased with ys
  if (front < (size_t)IH_NF * 21 * 4) front = (size_t)IH_NF * 21 * 4;
  front = (front + 15) & ~(size_t)15;
  const int soff = (int)front;
  const size_t lds = front + (size_t)IH_FR * 5 * 4;
  if (dtype == KK_F32) {
    if (fast)
      hipLaunchKernelGGL((istft_head_kernel<float, true>), grid, dim3(256), lds, st, (const float*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, soff, g_tables);
    else
      hipLaunchKernelGGL((istft_head_kernel<float, false>), grid, dim3(256), lds, st, (const float*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, soff, g_tables);
  } else {
    if (fast)
      hipLaunchKernelGGL((istft_head_kernel<bf16_t, true>), grid, dim3(256), lds, st, (const bf16_t*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, soff, g_tables);
    else
      hipLaunchKernelGGL((istft_head_kernel<bf16_t, false>), grid, dim3(256), lds, st, (const bf16_t*)x, xbs, ldx, len_frames, Tfmax, wav, wbs, soff, g_tables);
  }
  KK_CHECK_LAUNCH();
  return 0;
}
