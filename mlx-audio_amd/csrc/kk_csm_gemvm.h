// The matrix-core GEMV of the CSM single-token steps (kk_csm.hip) in its own header, so that tools/gridbar/gemvm_bench.hip can time the
// SAME kernel in isolation.  Included inside kk_csm.hip's anonymous namespace.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float kk_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 kk_bf16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ void ts_begin(unsigned long long* ts, int id) {
  if (ts && threadIdx.x == 0) {
    atomicMin(ts + 1, (unsigned long long)wall_clock64());
#ifdef KK_TS_PER_WG  // (tools/gridbar/gemvm_bench.hip: every workgroup's own start / end behind the 8 summary words)
    ts[8 + 2 * (blockIdx.x + 32 * blockIdx.y)] = (unsigned long long)wall_clock64();
#endif
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
      ts[0] = (unsigned long long)id;
      ts[4] = (unsigned long long)wall_clock64();
    }
  }
}
__device__ __forceinline__ void ts_mid(unsigned long long* ts) {
  if (ts && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) ts[3] = (unsigned long long)wall_clock64();
}
__device__ __forceinline__ void ts_mark(unsigned long long* ts, int k) {  // k = 5, 7: further marks of workgroup 0
  if (ts && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) ts[k] = (unsigned long long)wall_clock64();
}
__device__ __forceinline__ void ts_end(unsigned long long* ts) {
  if (ts && threadIdx.x == 0) {
    atomicMax(ts + 2, (unsigned long long)wall_clock64());
#ifdef KK_TS_PER_WG
    ts[9 + 2 * (blockIdx.x + 32 * blockIdx.y)] = (unsigned long long)wall_clock64();
    ts[8 + 2 * 256 + (blockIdx.x + 32 * blockIdx.y)] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);  // XCC_ID | HW_ID
#endif
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
      ts[6] = (unsigned long long)wall_clock64();
    }
  }
}

// (ids are clamped into their tables on the device: an id outside -- a caller's mistake, or a code sampled from non-finite logits -- reads a
// valid row instead of faulting the GPU)
__device__ __forceinline__ int clamp_id(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }


struct FGArgs {
  const float* x; long long xrs;   // input row m at x + m * xrs (PRO 2: 2K floats, gate | up)
  const float* nw; float eps;      // PRO 1
  const int* codes; int cstride, cb, V, rows; const float* emb;  // PRO 3: row m is item m / rows; its LAST row is emb[(codes[item * cstride] + cb * V)], others come from x
  const uint16_t* w;
  int K, N, M;
  int kper;                         // gemvm_kernel: K rows per split-K slice (NOT K / gridDim.y in the kernel: gridDim is a dependent vector load from the
                                    // hidden kernel arguments ahead of every weight load, + an integer division)
  const float* res; long long rrs;  // EPI 1
  float* out; long long ors;
  long long pss;                    // EPI 2: floats between the partial tiles of consecutive K slices
  int dbg;                          // gemvm_kernel: bit 0 = plain instead of nontemporal weight loads (KK_CSM_NT); the round-2 kernels: KK_CSM_DBG phase switches
  float* gather_out;                // PRO 1 with `codes`: row m is emb[(codes[m * cstride] + cb * V)] (K floats, no x); column block 0 also writes the raw rows here (pitch K)
  unsigned long long* ts; int ts_id;  // kk_csm_debug_timestamps (null in production)
};


// ---------------------------------------------------------------------------------------------------------------------------------------
// gemvm_kernel (round 3): the single-token product on the MATRIX CORES, with every weight byte of the workgroup requested up front.
// What the per-kernel times of gemv8_kernel showed (profiles/r03_b_csm_bf16w_kernel_stats.csv): 15.5 us for gate|up (33.5 MB: 2.2 TB/s),
// 8.7 us for q|k|v (3 MB), 7.3 us for o (2 MB), while a chain of EMPTY kernels costs 1.55 us per launch in graph replay
// (tools/gridbar/launchfloor.hip): a frame is bound by the latency INSIDE its kernels.  gemv8 keeps 8 loads x 4 waves = 32 KB in flight
// per CU, so gate|up's 128 KB per workgroup are four exposed HBM round trips, and its 64 fp32 accumulators per lane (8 columns x 8 rows)
// leave no registers for a deeper ring and need a 3-stage LDS reduction.  Here:
//   * x (fp32, after the prologue) is split EXACTLY into three bf16 terms by truncation, x = x1 + x2 + x3 (8 + 8 + 8 significand bits:
//     x1 = top half of x, x2 = top half of x - x1, x3 = x - x1 - x2, all subtractions exact), and the bf16 weights meet them in
//     v_mfma_f32_16x16x32_bf16: products of two bf16 are exact in fp32 and the instruction accumulates in fp32, so the result is an
//     fp32-arithmetic dot product of the fp32 input with the bf16 matrix -- what the round-2 kernels computed with v_pk_fma_f32 -- in a
//     different summation order.  A (16 m x 32 k) carries the terms: m slot 4 (r / 2) + 2 t + r % 2 = term t of input row r, first
//     instruction [x1 | x2], second [x3 | 0]; B (32 k x 16 n) is one 1-KiB wave load of the fragment pack.  A lane of the 16 x 16 result
//     holds rows 2 g, 2 g + 1 (g = lane / 16) of column lane % 16 as acc[0] + acc[2], acc[1] + acc[3]: 4 accumulator registers per 16
//     columns instead of 64, no cross-lane reduction at all, one 4-KiB-per-16-columns exchange between the 8 waves at the end.
//     A row's result does not depend on the other rows of the launch (an output element of the instruction reads its own A row only).
//   * 512 threads; wave w owns the 32-row K chunks w, w + 8, ... and all NSUB 16-column sub-blocks of them: 4 chunks x NSUB loads of
//     16 bytes per lane go out BEFORE the input is staged (K = 1024, 64 columns: the whole 128 KB of the workgroup at once), later
//     rounds (K = 2048) refill a slot as it is consumed.
//   * staging: thread (row r = tid % 8, octet tid / 8 + 64 p) loads 8 consecutive k of its row, applies the prologue, splits, and writes
//     four 16-byte A operands; the k-octet pitch inside a fragment is 288 bytes, so the 8 rows x 2 octets of a 16-lane group cover all
//     64 banks once (the rows' m slots leave 32-byte holes that the next octet fills), and the MFMA-side read of 16 lanes is 256
//     contiguous bytes.  LDS: 2304 bytes per 32 k (72 KB for K = 1024, 144 KB for K = 2048).
//   * RMSNorm: sum of squares per row while staging (fixed order: lane butterfly, then waves 0..7), scale applied to the finished dot
//     products; EPI as before.  grid = (column blocks, K slices, 8-row chunks of M).
template <int NSUB, int PRO, int EPI, int ROUNDS>
__global__ __launch_bounds__(512) void gemvm_kernel(FGArgs a) {
  constexpr int KOP = 288, FRAG = 4 * KOP, CHB = 2 * FRAG;
  extern __shared__ __attribute__((aligned(16))) char smc[];
  ts_begin(a.ts, a.ts_id);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = blockIdx.x, K = a.K;
  const int Kper = a.kper, k_lo = blockIdx.y * Kper;
  const int nch = Kper >> 5;  // <= 32 ROUNDS (launcher)
  const int m0 = blockIdx.z * 8;
  const int M = a.M - m0 < 8 ? a.M - m0 : 8;
  const int mainb = (nch + 1) * CHB > NSUB * 4096 ? (nch + 1) * CHB : NSUB * 4096;
  char* xf = smc;                           // [nch + 1 spare][2 fragments][4 k octets at pitch 288][16 m slots][8 bf16]
  float* red = (float*)smc;                 // [8 waves][NSUB][2][64] (aliases xf after the main loop)
  float* rsq = (float*)(smc + mainb);       // [8 waves][8 rows] sums of squares (PRO 1)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* wblk = (const u32x4*)a.w + ((long long)nb * (K >> 5) + (k_lo >> 5)) * (NSUB * 64) + lane;  // chunk c, sub-block s at + (c NSUB + s) 64
  u32x4 ring[4][NSUB];
  // ---- the input rows' loads go out FIRST, the first round of weight loads right behind them: a wave's loads return in order, so the
  // input is there ahead of the weights and the prologue / split runs while the weights are in flight.  Every load below is unconditional
  // (a chunk or octet past the end is clamped to a valid one and its result dropped) and the code is straight-line per ROUNDS: the
  // compiler's s_waitcnt vmcnt then counts exactly.  (With the weights first, or with loads under a branch or in a rolled loop, the staging
  // waited for the whole weight stream: 6.5-8.5 of gate|up's 10.5 us in the in-kernel marks, tools/csm_timeline.py.)
  const int r = tid & 7, oi = tid >> 3;
  const float* row;
  {
    const int mg = m0 + (r < M ? r : 0);
    if (PRO == 3) {  // an item's last row comes from the audio embedding table (sesame.py:373-392), its other rows from x
      const int item = mg / a.rows, rr = mg - item * a.rows;
      row = rr == a.rows - 1 ? a.emb + (long long)(clamp_id(a.codes[(long long)item * a.cstride], a.V) + a.cb * a.V) * K : a.x + (long long)item * a.xrs;
    } else if (PRO == 1 && a.codes) {  // rows gathered from a table by code (the depth decoder's input from its projected-embedding table)
      row = a.emb + (long long)(clamp_id(a.codes[(long long)mg * a.cstride], a.V) + a.cb * a.V) * K;
    } else {
      row = a.x + (long long)mg * a.xrs;
    }
    row += k_lo;
  }
  const int noct = Kper >> 3;
  float4 g[ROUNDS][2][2], u[PRO == 2 ? ROUNDS : 1][2][2], nw[PRO == 1 ? ROUNDS : 1][2][2];
#pragma unroll
  for (int ps = 0; ps < ROUNDS; ++ps)
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // thread (row r, octets oi + 64 p + 128 ps): 8 consecutive k
      const int o = 128 * ps + oi + 64 * p, oc = o < noct ? o : 0;
      g[ps][p][0] = *(const float4*)(row + 8 * oc); g[ps][p][1] = *(const float4*)(row + 8 * oc + 4);
      if (PRO == 2) { u[ps][p][0] = *(const float4*)(row + K + 8 * oc); u[ps][p][1] = *(const float4*)(row + K + 8 * oc + 4); }
      if (PRO == 1) { nw[ps][p][0] = *(const float4*)(a.nw + k_lo + 8 * oc); nw[ps][p][1] = *(const float4*)(a.nw + k_lo + 8 * oc + 4); }
    }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0);
  if (a.dbg & 1) {  // matrices that are re-read within the reach of the memory-side cache (the depth decoder: 222 MB, 31 times per frame): plain loads
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = wave + 8 * j, cc = c < nch ? c : nch - 1;
#pragma unroll
      for (int s = 0; s < NSUB; ++s) ring[j][s] = *(wblk + (long long)(cc * NSUB + s) * 64);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = wave + 8 * j, cc = c < nch ? c : nch - 1;
#pragma unroll
      for (int s = 0; s < NSUB; ++s) ring[j][s] = __builtin_nontemporal_load(wblk + (long long)(cc * NSUB + s) * 64);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise issues most of the weight loads BEHIND the split arithmetic to save registers)
  // ---- prologue, exact three-way bf16 split, fragment order
  {
    const float live = r < M ? 1.0f : 0.0f;
    float ssq = 0.f;
    char* dst0 = xf + (4 * (r >> 1) + (r & 1)) * 16;
#pragma unroll
    for (int ps = 0; ps < ROUNDS; ++ps)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int o = 128 * ps + oi + 64 * p;
        float t[8] = {g[ps][p][0].x, g[ps][p][0].y, g[ps][p][0].z, g[ps][p][0].w, g[ps][p][1].x, g[ps][p][1].y, g[ps][p][1].z, g[ps][p][1].w};
        if (PRO == 2) {  // silu(gate) * up
          const float uu[8] = {u[ps][p][0].x, u[ps][p][0].y, u[ps][p][0].z, u[ps][p][0].w, u[ps][p][1].x, u[ps][p][1].y, u[ps][p][1].z, u[ps][p][1].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = t[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-t[e])) * uu[e];
        } else if (PRO == 1) {
          if (a.gather_out && nb == 0 && o < noct && r < M) {  // the gathered rows ARE the residual stream: materialised once, by column block 0
            float* go = a.gather_out + (long long)(m0 + r) * K + k_lo + 8 * o;
            *(float4*)go = g[ps][p][0];
            *(float4*)(go + 4) = g[ps][p][1];
          }
          const float ww[8] = {nw[ps][p][0].x, nw[ps][p][0].y, nw[ps][p][0].z, nw[ps][p][0].w, nw[ps][p][1].x, nw[ps][p][1].y, nw[ps][p][1].z, nw[ps][p][1].w};
          const float cnt = o < noct ? live : 0.0f;  // (a clamped duplicate past the end does not count)
#pragma unroll
          for (int e = 0; e < 8; ++e) { ssq = __builtin_fmaf(t[e] * cnt, t[e], ssq); t[e] *= ww[e]; }
        }
        unsigned x1[4], x2[4], x3[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float a0 = t[e] * live, a1 = t[e + 1] * live;
          const float b0 = a0 - __uint_as_float(__float_as_uint(a0) & 0xffff0000u), b1 = a1 - __uint_as_float(__float_as_uint(a1) & 0xffff0000u);
          const float c0 = b0 - __uint_as_float(__float_as_uint(b0) & 0xffff0000u), c1 = b1 - __uint_as_float(__float_as_uint(b1) & 0xffff0000u);
          x1[e >> 1] = __builtin_amdgcn_perm(__float_as_uint(a1), __float_as_uint(a0), 0x07060302u);  // high halves: lower k in the low half
          x2[e >> 1] = __builtin_amdgcn_perm(__float_as_uint(b1), __float_as_uint(b0), 0x07060302u);
          x3[e >> 1] = __builtin_amdgcn_perm(__float_as_uint(c1), __float_as_uint(c0), 0x07060302u);
        }
        {  // (an octet past the end lands in a spare chunk behind the last one: no branch, or the compiler sinks the LOADS into it)
          char* d = dst0 + (o < noct ? o >> 2 : nch) * CHB + (o & 3) * KOP;
          *(uint4*)d = make_uint4(x1[0], x1[1], x1[2], x1[3]);
          *(uint4*)(d + 32) = make_uint4(x2[0], x2[1], x2[2], x2[3]);
          *(uint4*)(d + FRAG) = make_uint4(x3[0], x3[1], x3[2], x3[3]);
          *(uint4*)(d + FRAG + 32) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    if (PRO == 1) {
      ssq += __shfl_xor(ssq, 8); ssq += __shfl_xor(ssq, 16); ssq += __shfl_xor(ssq, 32);
      if (lane < 8) rsq[wave * 8 + lane] = ssq;  // (its own LDS words: visible behind the barriers below)
    }
  }
  kk_f32x4 acc[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) acc[s] = kk_f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  ts_mid(a.ts);
  {
    const char* xl = xf + (lane >> 4) * KOP + (lane & 15) * 16;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = wave + 8 * (4 * rd + j);
        kk_bf16x8 b[NSUB];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) b[s] = __builtin_bit_cast(kk_bf16x8, ring[j][s]);
        if (rd + 1 < ROUNDS) {  // refill the slot for the next round
          const int cn = c + 32 < nch ? c + 32 : nch - 1;
#pragma unroll
          for (int s = 0; s < NSUB; ++s) ring[j][s] = __builtin_nontemporal_load(wblk + (long long)(cn * NSUB + s) * 64);
        }
        if (c < nch) {
          const kk_bf16x8 a1 = *(const kk_bf16x8*)(xl + c * CHB), a2 = *(const kk_bf16x8*)(xl + c * CHB + FRAG);
#pragma unroll
          for (int s = 0; s < NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[s], acc[s], 0, 0, 0);
#pragma unroll
          for (int s = 0; s < NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b[s], acc[s], 0, 0, 0);
        }
      }
    }
  }
  // ---- the waves' partial tiles meet in LDS (wave order); lane L of sub-block s: rows 2 (L / 16) + j, column 16 s + L % 16
  __syncthreads();  // xf is no longer read
  ts_mark(a.ts, 5);
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    red[((wave * NSUB + s) * 2 + 0) * 64 + lane] = acc[s][0] + acc[s][2];
    red[((wave * NSUB + s) * 2 + 1) * 64 + lane] = acc[s][1] + acc[s][3];
  }
  __syncthreads();
  ts_mark(a.ts, 7);
  if (tid < NSUB * 128) {
    const int s = tid >> 7, j = (tid >> 6) & 1;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[((w * NSUB + s) * 2 + j) * 64 + lane];
    const int rr = 2 * (lane >> 4) + j, n = nb * (16 * NSUB) + 16 * s + (lane & 15);
    if (rr < M && n < a.N) {
      const long long mg = m0 + rr;
      if (PRO == 1) {
        float q = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) q += rsq[w * 8 + rr];
        t *= 1.0f / sqrtf(q / (float)K + a.eps);
      }
      if (EPI == 1) t += a.res[mg * a.rrs + n];
      if (EPI == 2) a.out[(long long)blockIdx.y * a.pss + mg * a.ors + n] = t;  // K slice blockIdx.y of a split-K launch
      else a.out[mg * a.ors + n] = t;
    }
  }
  ts_end(a.ts);
}
static size_t gm_lds_bytes(int nsub, int kper) {
  const size_t xb = (size_t)(kper / 32 + 1) * 2304, rb = (size_t)nsub * 4096;
  return (xb > rb ? xb : rb) + 256;
}

