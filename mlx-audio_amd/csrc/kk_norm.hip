// Normalisation family on frames-major tensors:
//   * instance-norm statistics over the valid frames of each (utterance, channel)
//       -> InstanceNorm1d (istftnet.py:216-268, ddof 0, eps 1e-5)
//   * AdaIN apply + activation (Snake / LeakyReLU), optionally followed by the depth-wise
//     k3 s2 transposed conv + front zero pad of the up-sampling residual path
//       -> AdaIN1d (istftnet.py:327-338), Snake (istftnet.py:382,389), AdainResBlk1d._residual
//          (istftnet.py:874-882)
//   * LayerNorm / AdaLayerNorm over channels (modules.py:33,71-90,448,480,566)
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

// ---------------------------------------------------------------- instance-norm statistics
// partial[b][chunk][0|1][C]: sums of (x - shift) and (x - shift)^2 with shift = x[b][0][c]
template <typename T>
__global__ __launch_bounds__(256) void stats_partial_kernel(KKStatsArgs a, int cw) {
  __shared__ float red[2][256];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int L = kk_len(a.len, b);
  const int r0 = chunk * a.rows_per_chunk;
  const int r1 = min(L, r0 + a.rows_per_chunk);
  const int RL = 256 / cw;
  const int cl = tid % cw, rl = tid / cw;
  const T* xb = (const T*)a.x + (long long)b * a.xbs;
  float* pb = a.partial + ((long long)b * a.nchunk + chunk) * 2 * a.C;
  for (int cbase = 0; cbase < a.C; cbase += cw) {
    const int c = cbase + cl;
    float s = 0.f, ss = 0.f;
    if (c < a.C && L > 0) {
      const float shift = kk_ld(xb + c);
      for (int r = r0 + rl; r < r1; r += RL) {
        const float v = kk_ld(xb + (long long)r * a.ldx + c) - shift;
        s += v;
        ss = __builtin_fmaf(v, v, ss);
      }
    }
    red[0][tid] = s;
    red[1][tid] = ss;
    __syncthreads();
    if (rl == 0 && c < a.C) {
      for (int k = 1; k < RL; ++k) {
        s += red[0][k * cw + cl];
        ss += red[1][k * cw + cl];
      }
      pb[c] = s;
      pb[a.C + c] = ss;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stats_final_kernel(KKStatsArgs a) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (c >= a.C) {
    if (a.pa && c < a.Cp) {  // pad channels: transform to exactly zero
      a.pa[(long long)b * a.pstride + c] = 0.f;
      a.pb[(long long)b * a.pstride + c] = 0.f;
    }
    return;
  }
  const int L = kk_len(a.len, b);
  float mean = 0.f, rstd = 0.f;
  if (a.fused == 2) {  // statistics already known: only fold them with this layer's gamma / beta
    mean = a.mean[(long long)b * a.C + c];
    rstd = a.rstd[(long long)b * a.C + c];
  } else if (L > 0) {
    const int nch = a.fused ? a.nchunk : kk_cdiv(L, a.rows_per_chunk);
    double S = 0.0, SS = 0.0;
    const float* pb = a.partial + (long long)b * a.nchunk * 2 * a.C;
    for (int k = 0; k < nch; ++k) {
      S += (double)pb[(long long)k * 2 * a.C + c];
      SS += (double)pb[(long long)k * 2 * a.C + a.C + c];
    }
    const double shift = a.fused ? 0.0 : (double)kk_ld((const T*)a.x + (long long)b * a.xbs + c);
    const double m = S / L;
    double var = SS / L - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)(shift + m);
    rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  if (a.fused != 2) {
    if (a.mean) a.mean[(long long)b * a.C + c] = mean;
    if (a.rstd) a.rstd[(long long)b * a.C + c] = rstd;
  }
  if (a.pa) {
    const float g = 1.0f + a.gb[(long long)b * a.gbs + c], be = a.gb[(long long)b * a.gbs + a.C + c];
    const float A = rstd * g;
    a.pa[(long long)b * a.pstride + c] = A;
    a.pb[(long long)b * a.pstride + c] = be - mean * A;
  }
}

// fused partials (un-shifted per-tile column sums from the conv epilogues) -> mean / rstd / folded AdaIN parameters.
// Deterministic: fixed lane -> tile assignment and a fixed reduction tree.
__global__ __launch_bounds__(1024) void norm_finalize_tiles_kernel(KKStatsArgs a) {
  __shared__ double red[2][32][33];
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
  const int L = kk_len(a.len, b);
  double S = 0.0, SS = 0.0;
  if (c < a.C && L > 0) {
    const float* pb = a.partial + (long long)b * a.nchunk * 2 * a.C;
    for (int k = tl; k < a.nchunk; k += 32) {
      S += (double)pb[(long long)k * 2 * a.C + c];
      SS += (double)pb[(long long)k * 2 * a.C + a.C + c];
    }
  }
  red[0][tl][cl] = S;
  red[1][tl][cl] = SS;
  __syncthreads();
  if (tl != 0) return;
  if (c >= a.C) {
    if (a.pa && c < a.Cp) {
      a.pa[(long long)b * a.pstride + c] = 0.f;
      a.pb[(long long)b * a.pstride + c] = 0.f;
    }
    return;
  }
  for (int k = 1; k < 32; ++k) {
    S += red[0][k][cl];
    SS += red[1][k][cl];
  }
  float mean = 0.f, rstd = 0.f;
  if (L > 0) {
    const double m = S / L;
    double var = SS / L - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  if (a.mean) a.mean[(long long)b * a.C + c] = mean;
  if (a.rstd) a.rstd[(long long)b * a.C + c] = rstd;
  if (a.pa) {
    const float g = 1.0f + a.gb[(long long)b * a.gbs + c], be = a.gb[(long long)b * a.gbs + a.C + c];
    const float A = rstd * g;
    a.pa[(long long)b * a.pstride + c] = A;
    a.pb[(long long)b * a.pstride + c] = be - mean * A;
  }
}

// 16-byte vectorised statistics pass for bf16 tensors with C in {64, 128, 256, 512, 1024}: a thread owns 8 channels,
// C/8 threads cover a row, 256/(C/8) rows advance together.  Same partial layout as stats_partial_kernel.
__global__ __launch_bounds__(256) void stats_partial_bf16v_kernel(KKStatsArgs a) {
  __shared__ float red[2][256][8];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int L = kk_len(a.len, b);
  const int r0 = chunk * a.rows_per_chunk;
  const int r1 = min(L, r0 + a.rows_per_chunk);
  const int cwt = a.C >> 3, RL = 256 / cwt;  // threads per row, rows in flight
  const int cl = tid % cwt, rl = tid / cwt;
  const bf16_t* xb = (const bf16_t*)a.x + (long long)b * a.xbs;
  union { uint4 u; bf16_t h[8]; } t;
  float sh[8], s[8], q[8];
  t.u = *(const uint4*)(xb + cl * 8);
#pragma unroll
  for (int k = 0; k < 8; ++k) { sh[k] = (float)t.h[k]; s[k] = 0.f; q[k] = 0.f; }
  if (L > 0) {
    for (int r = r0 + rl; r < r1; r += RL) {
      t.u = *(const uint4*)(xb + (long long)r * a.ldx + cl * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = (float)t.h[k] - sh[k];
        s[k] += v;
        q[k] = __builtin_fmaf(v, v, q[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][tid][k] = s[k]; red[1][tid][k] = q[k]; }
  __syncthreads();
  if (rl == 0) {
    for (int j = 1; j < RL; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) { s[k] += red[0][j * cwt + cl][k]; q[k] += red[1][j * cwt + cl][k]; }
    float* pb = a.partial + ((long long)b * a.nchunk + chunk) * 2 * a.C + cl * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { pb[k] = s[k]; pb[a.C + k] = q[k]; }
  }
}


// The same pass for ANY channel count (the decoder's 514- and 1090-channel concatenations, pitch 576 / 1152): a thread owns the 8 channels of one
// 16-byte group of the row PITCH (pad channels are read and never reported), ldx / 8 threads cover a row, 4 rows are in flight per thread.
// Block = ldx / 8 threads (<= 256).  Same partial layout as stats_partial_kernel; the scalar kernel took ~80 us per decoder block on these.
__global__ __launch_bounds__(256) void stats_partial_bf16p_kernel(KKStatsArgs a) {
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int L = kk_len(a.len, b);
  const int r0 = chunk * a.rows_per_chunk;
  const int r1 = min(L, r0 + a.rows_per_chunk);
  const bf16_t* xb = (const bf16_t*)a.x + (long long)b * a.xbs + tid * 8;
  union U { uint4 u; bf16_t h[8]; };
  float sh[8], s[8], q[8];
  {
    U t;
    t.u = *(const uint4*)xb;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sh[k] = (float)t.h[k]; s[k] = 0.f; q[k] = 0.f; }
  }
  for (int r = r0; r < r1; r += 4) {
    U t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i].u = *(const uint4*)(xb + (long long)(r + i < r1 ? r + i : r1 - 1) * a.ldx);  // (clamped: no load under a condition)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float live = r + i < r1 ? 1.0f : 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = ((float)t[i].h[k] - sh[k]) * live;
        s[k] += v;
        q[k] = __builtin_fmaf(v, v, q[k]);
      }
    }
  }
  float* pb = a.partial + ((long long)b * a.nchunk + chunk) * 2 * a.C;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = tid * 8 + k;
    if (c < a.C) { pb[c] = s[k]; pb[a.C + c] = q[k]; }
  }
}

// ---------------------------------------------------------------- AdaIN apply (+ act, + pool)
__device__ __forceinline__ float act_apply(float y, int act, float slope, float alpha, bool fast) {
  if (act == KK_ACT_LRELU) return y > 0.f ? y : y * slope;
  if (act == KK_ACT_SNAKE) {
    const float s = fast ? __sinf(alpha * y) : sinf(alpha * y);
    return y + (1.0f / alpha) * (s * s);
  }
  return y;
}

constexpr int ADAIN_ROWS = 32;

template <typename T>
__global__ __launch_bounds__(256) void adain_act_kernel(KKAdainArgs a) {
  extern __shared__ float prm[];  // [5][C]: mean, rstd, 1+gamma, beta, alpha
  const int tid = threadIdx.x, b = blockIdx.y;
  const int C = a.C;
  float* p_mean = prm;
  float* p_rstd = prm + C;
  float* p_g = prm + 2 * C;
  float* p_b = prm + 3 * C;
  float* p_al = prm + 4 * C;
  for (int c = tid; c < C; c += 256) {
    p_mean[c] = a.mean[(long long)b * C + c];
    p_rstd[c] = a.rstd[(long long)b * C + c];
    p_g[c] = 1.0f + a.gb[(long long)b * a.gbs + c];
    p_b[c] = a.gb[(long long)b * a.gbs + C + c];
    p_al[c] = a.alpha ? a.alpha[c] : 1.0f;
  }
  __syncthreads();
  const int Lin = kk_len(a.len_in, b);
  const int Lout = a.pool ? 2 * Lin : Lin;
  const int row0 = blockIdx.x * ADAIN_ROWS;
  const T* xb = (const T*)a.x + (long long)b * a.xbs;
  T* ob = (T*)a.out + (long long)b * a.obs;
  const int total = ADAIN_ROWS * a.Cpad;
  for (int e = tid; e < total; e += 256) {
    const int rr = e / a.Cpad, c = e - rr * a.Cpad;
    const int row = row0 + rr;
    if (row >= a.Lmax_out) break;
    float v = 0.f;
    if (row < Lout && c < C) {
      if (!a.pool) {
        const float xn = (kk_ld(xb + (long long)row * a.ldx + c) - p_mean[c]) * p_rstd[c];
        v = act_apply(xn * p_g[c] + p_b[c], a.act, a.slope, p_al[c], a.fast != 0);
      } else if (row > 0) {
        // depth-wise ConvTranspose1d(k3, s2, p1) then one zero row in FRONT (istftnet.py:880-881):
        // out[j], j' = j-1:  even j'=2m -> y[m]*w1 ; odd j'=2m+1 -> y[m]*w2 + y[m+1]*w0 ; + bias
        const int jp = row - 1, m = jp >> 1;
        const float y0 = act_apply((kk_ld(xb + (long long)m * a.ldx + c) - p_mean[c]) * p_rstd[c] * p_g[c] + p_b[c], a.act,
                                   a.slope, p_al[c], a.fast != 0);
        const float* w3 = a.pool_w + 3 * c;
        if ((jp & 1) == 0) {
          v = y0 * w3[1] + a.pool_b[c];
        } else {
          float acc = y0 * w3[2];
          if (m + 1 < Lin) {
            const float y1 = act_apply(
                (kk_ld(xb + (long long)(m + 1) * a.ldx + c) - p_mean[c]) * p_rstd[c] * p_g[c] + p_b[c], a.act, a.slope, p_al[c],
                a.fast != 0);
            acc += y1 * w3[0];
          }
          v = acc + a.pool_b[c];
        }
      }
    }
    kk_st(ob + (long long)row * a.ldo + c, v);
  }
}


// 16-byte vectorised form for bf16 tensors (the three up-sampling AdainResBlk1d inputs: 1090 and 512 channels; the scalar kernel spends an
// integer division and two 2-byte memory operations per element: 93-190 us per launch).  A thread owns the 8 channels of one 16-byte group
// (parameters in registers), Cpad / 8 threads cover a row; the arithmetic of an element is the scalar kernel's, in the same order.
// act: KK_ACT_NONE / KK_ACT_LRELU.
__global__ __launch_bounds__(256) void adain_act_bf16v_kernel(KKAdainArgs a, int G) {
  const int tid = threadIdx.x, b = blockIdx.y;
  const int g = tid % G, rl = tid / G, RL = blockDim.x / G;
  const int C = a.C, c0 = g * 8;
  float mean[8], rstd[8], ga[8], be[8], w0[8], w1[8], w2[8], pbias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k < C ? c0 + k : C - 1;
    mean[k] = a.mean[(long long)b * C + c];
    rstd[k] = a.rstd[(long long)b * C + c];
    ga[k] = 1.0f + a.gb[(long long)b * a.gbs + c];
    be[k] = a.gb[(long long)b * a.gbs + C + c];
    w0[k] = a.pool ? a.pool_w[3 * c] : 0.f;
    w1[k] = a.pool ? a.pool_w[3 * c + 1] : 0.f;
    w2[k] = a.pool ? a.pool_w[3 * c + 2] : 0.f;
    pbias[k] = a.pool ? a.pool_b[c] : 0.f;
  }
  const int Lin = kk_len(a.len_in, b);
  const int Lout = a.pool ? 2 * Lin : Lin;
  const int row0 = blockIdx.x * ADAIN_ROWS;
  const bf16_t* xb = (const bf16_t*)a.x + (long long)b * a.xbs + c0;
  bf16_t* ob = (bf16_t*)a.out + (long long)b * a.obs + c0;
  union U { uint4 u; bf16_t h[8]; };
  const int lin_hi = Lin > 0 ? Lin - 1 : 0;
  for (int rr = rl; rr < ADAIN_ROWS; rr += RL) {
    const int row = row0 + rr;
    if (row >= a.Lmax_out) break;
    U o;
    o.u = make_uint4(0u, 0u, 0u, 0u);
    if (row < Lout) {
      if (!a.pool) {
        U t;
        t.u = *(const uint4*)(xb + (long long)row * a.ldx);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xn = ((float)t.h[k] - mean[k]) * rstd[k];
          float y = xn * ga[k] + be[k];
          if (a.act == KK_ACT_LRELU) y = y > 0.f ? y : y * a.slope;
          o.h[k] = (bf16_t)(c0 + k < C ? y : 0.f);
        }
      } else if (row > 0) {
        const int jp = row - 1, m = jp >> 1;
        const int m1 = m + 1 < Lin ? m + 1 : lin_hi;
        U t0, t1;
        t0.u = *(const uint4*)(xb + (long long)m * a.ldx);
        t1.u = *(const uint4*)(xb + (long long)m1 * a.ldx);  // (clamped; unused on even rows and past the end)
        const bool odd = (jp & 1) != 0, has1 = m + 1 < Lin;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float y0 = (((float)t0.h[k] - mean[k]) * rstd[k]) * ga[k] + be[k];
          float y1 = (((float)t1.h[k] - mean[k]) * rstd[k]) * ga[k] + be[k];
          if (a.act == KK_ACT_LRELU) {
            y0 = y0 > 0.f ? y0 : y0 * a.slope;
            y1 = y1 > 0.f ? y1 : y1 * a.slope;
          }
          float v;
          if (!odd) {
            v = y0 * w1[k] + pbias[k];
          } else {
            float acc = y0 * w2[k];
            if (has1) acc += y1 * w0[k];
            v = acc + pbias[k];
          }
          o.h[k] = (bf16_t)(c0 + k < C ? v : 0.f);
        }
      }
    }
    *(uint4*)(ob + (long long)row * a.ldo) = o.u;
  }
}

// ---------------------------------------------------------------- LayerNorm over channels (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(KKLnArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv, b = blockIdx.y;
  if (row >= a.Lmax) return;
  const int L = kk_len(a.len, b);
  T* orow = (T*)a.out + (long long)b * a.obs + (long long)row * a.ldo;
  if (row >= L) {
    for (int c = lane; c < a.C; c += 64) kk_st(orow + c, 0.f);
    return;
  }
  const T* xr = (const T*)a.x + (long long)b * a.xbs + (long long)row * a.ldx;
  const T* rr = a.res ? (const T*)a.res + (long long)b * a.rbs + (long long)row * a.ldr : nullptr;
  constexpr int MAXV = 32;  // C <= 2048
  float v[MAXV];
  // the scale / shift of the last loop are requested HERE, with the row: they depend on nothing, and read where they are used they are a second
  // exposed round trip behind the two reductions (the kernel is one wave per row: nothing else hides it)
  float pg[MAXV], pb[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    float t = 0.f;
    pg[i] = pb[i] = 0.f;
    if (c < a.C) {
      t = kk_ld(xr + c);
      if (rr) t += kk_ld(rr + c);
      if (a.gb) { pg[i] = a.gb[(long long)b * a.gbs + c]; pb[i] = a.gb[(long long)b * a.gbs + a.C + c]; }
      else { pg[i] = a.w[c]; pb[i] = a.bias[c]; }
    }
    v[i] = t;
    s += t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)a.C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < a.C) {
      const float d = v[i] - mean;
      ss = __builtin_fmaf(d, d, ss);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = 1.0f / sqrtf(ss / (float)a.C + a.eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < a.C) {
      float y = (v[i] - mean) * rstd;
      if (a.gb) y = (1.0f + pg[i]) * y + pb[i];
      else y = y * pg[i] + pb[i];
      if (a.act == KK_ACT_LRELU) y = y > 0.f ? y : y * a.slope;
      kk_st(orow + c, y);
    }
  }
}

}  // namespace

size_t kk_stats_partial_floats(int B, int C, int Lmax, int rows_per_chunk) {
  return (size_t)B * kk_cdiv(Lmax, rows_per_chunk) * 2 * C;
}

int kk_launch_instnorm_stats(KKStatsArgs a, int B, int dtype, hipStream_t st) {
  if (B <= 0 || a.C <= 0) return 0;
  a.nchunk = kk_cdiv(a.Lmax, a.rows_per_chunk);
  if (a.nchunk <= 0) a.nchunk = 1;
  const int cw = a.C >= 256 ? 256 : (a.C > 64 ? 128 : 64);
  a.fused = 0;
  dim3 g1(a.nchunk, B), g2(kk_cdiv(a.pa && a.Cp > a.C ? a.Cp : a.C, 256), B);
  if (dtype == KK_F32) {
    hipLaunchKernelGGL(stats_partial_kernel<float>, g1, dim3(256), 0, st, a, cw);
    hipLaunchKernelGGL(stats_final_kernel<float>, g2, dim3(256), 0, st, a);
  } else {
    const bool vec = (a.C == 64 || a.C == 128 || a.C == 256 || a.C == 512 || a.C == 1024) && a.ldx % 8 == 0 && !((uintptr_t)a.x & 15);
    const bool pitch = !vec && a.ldx % 8 == 0 && a.ldx / 8 <= 256 && a.C <= a.ldx && !((uintptr_t)a.x & 15) && a.C > 64;
    if (vec)
      hipLaunchKernelGGL(stats_partial_bf16v_kernel, g1, dim3(256), 0, st, a);
    else if (pitch)
      hipLaunchKernelGGL(stats_partial_bf16p_kernel, g1, dim3(a.ldx / 8), 0, st, a);
    else
      hipLaunchKernelGGL(stats_partial_kernel<bf16_t>, g1, dim3(256), 0, st, a, cw);
    hipLaunchKernelGGL(stats_final_kernel<bf16_t>, g2, dim3(256), 0, st, a);
  }
  KK_CHECK_LAUNCH();
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void stat_add_row_kernel(const bf16_t* x, long long xbs, float* part, int ntiles, int C) {
  const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const float v = (float)x[(long long)b * xbs + c];
  float* p = part + ((long long)b * ntiles * 2) * C + c;
  p[0] += v;
  p[C] = __builtin_fmaf(v, v, p[C]);
}
}  // namespace

int kk_launch_stat_add_row(const void* x, long long xbs, float* part, int ntiles, int C, int B, hipStream_t st) {
  if (B <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(stat_add_row_kernel, dim3(B, kk_cdiv(C, 256)), dim3(256), 0, st, (const bf16_t*)x, xbs, part, ntiles, C);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_norm_finalize(KKStatsArgs a, int B, hipStream_t st) {
  if (B <= 0 || a.C <= 0) return 0;
  const int cover = a.pa ? (a.Cp > a.C ? a.Cp : a.C) : a.C;
  if (a.fused != 2) {
    a.fused = 1;
    // hundreds of per-tile partials per channel: 32 channels x 32 tile lanes per workgroup, then an LDS tree
    hipLaunchKernelGGL(norm_finalize_tiles_kernel, dim3(kk_cdiv(cover, 32), B), dim3(1024), 0, st, a);
  } else {
    hipLaunchKernelGGL(stats_final_kernel<float>, dim3(kk_cdiv(cover, 256), B), dim3(256), 0, st, a);
  }
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_adain_act(const KKAdainArgs& a, int B, int dtype, hipStream_t st) {
  if (B <= 0 || a.Lmax_out <= 0) return 0;
  if (a.C > 2048) return kk_fail("adain_act: C > 2048");
  dim3 grid(kk_cdiv(a.Lmax_out, ADAIN_ROWS), B);
  const size_t sh = (size_t)5 * a.C * sizeof(float);
  const int G = a.Cpad / 8;
  const bool vec = dtype == KK_BF16 && (a.act == KK_ACT_NONE || a.act == KK_ACT_LRELU) && a.Cpad % 8 == 0 && G >= 1 && G <= 256 && a.Cpad <= a.ldx &&
                   a.Cpad <= a.ldo && a.ldx % 8 == 0 && a.ldo % 8 == 0 && !((uintptr_t)a.x & 15) && !((uintptr_t)a.out & 15) && a.C >= 8;
  if (dtype == KK_F32)
    hipLaunchKernelGGL(adain_act_kernel<float>, grid, dim3(256), sh, st, a);
  else if (vec)
    hipLaunchKernelGGL(adain_act_bf16v_kernel, grid, dim3(G * (256 / G)), 0, st, a, G);
  else
    hipLaunchKernelGGL(adain_act_kernel<bf16_t>, grid, dim3(256), sh, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_layernorm(const KKLnArgs& a, int B, int dtype, hipStream_t st) {
  if (B <= 0 || a.Lmax <= 0) return 0;
  if (a.C > 2048) return kk_fail("layernorm: C > 2048");
  dim3 grid(kk_cdiv(a.Lmax, 4), B);
  if (dtype == KK_F32)
    hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, dim3(256), 0, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}
