// Persistent bf16 MFMA implicit-GEMM convolution for LONG frames-major tensors (the generator's 20F / 120F+1 stages).
//
// Same math and the same epilogue options as kk_conv_mfma.hip; different machine mapping, driven by what the 128-row
// kernel measured (tools/bench_conv.py, PMC and ablations in DESIGN.md 3.1):
//   * W re-read from L2 for every 128-row tile moves 16 KiB per 16 MFMAs and wave: the L2 -> LDS path (28 B/clk/CU
//     achieved) takes as long as the MFMAs and the two do not overlap  ->  256-row tiles halve W bytes per MFMA
//   * prologue loads and epilogue stores of a tile ran at HBM speed but NOT under the MFMAs  ->  one persistent
//     workgroup per CU walks tiles; the next tile's X slab and W tiles are prefetched while the current tile computes,
//     the residual rows are requested before the last tap, stores are fire-and-forget
//   * W tiles are prefetched TWO taps ahead (two register sets, two LDS buffers) so an L2 round trip has two
//     iterations of MFMAs to land
// Geometry: 512 threads = 8 waves as 4 (rows) x 2 (cols), wave tile 64 x 64 (four 32x32 accumulators), block tile
// 256 rows x 128 output channels, K walked as (64-channel slab) x (tap).  LDS: X slab [256+50][72] bf16 (44 KB) +
// 2 W tiles [128][72] (37 KB) + AdaIN parameter table; the epilogue re-uses the X slab as a 64-row fp32 tile (4 passes).
#include <stdlib.h>

#include "kk_common.h"
#include "kk_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 128, CK = 64, NT = 512;
constexpr int XLD = CK + 8;
constexpr int MAX_HALO = 50;
constexpr int XROWS = BM + MAX_HALO;          // 306
constexpr int XS_BYTES = XROWS * XLD * 2;      // 44064
constexpr int WS_BYTES = BN * XLD * 2;         // 18432
constexpr int PS_BYTES = 2 * 3 * CK * 4;       // 1536
constexpr int LDS_BYTES = XS_BYTES + 2 * WS_BYTES + PS_BYTES;  // 82464 -> one workgroup per CU
constexpr int XREG = (XROWS * 8 + NT - 1) / NT;  // 5
constexpr int CLD = BN;                        // fp32 epilogue tile pitch; 64 rows x 128 x 4 B = 32 KiB <= XS_BYTES

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

union U16 {
  uint4 u;
  bf16_t h[8];
};

struct TileId {
  int b, phase, n0, q0, mt;
};

template <bool NRM>
__global__ __launch_bounds__(NT, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_mfma3_kernel(KKMfmaArgs a, int ntm, int ntn,
                                                                                                      int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xs = (bf16_t*)smem;
  bf16_t* Ws0 = (bf16_t*)(smem + XS_BYTES);
  bf16_t* Ws1 = (bf16_t*)(smem + XS_BYTES + WS_BYTES);
  float* Ps = (float*)(smem + XS_BYTES + 2 * WS_BYTES);  // [2][3][64]
  float* Cs = (float*)smem;                               // epilogue tile aliases the X slab only

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;

  auto decode = [&](int t) __attribute__((always_inline)) {
    TileId r;
    r.mt = t % ntm;
    int rest = t / ntm;
    const int nt = rest % ntn;
    rest /= ntn;
    r.b = rest / nphase;
    r.phase = rest - r.b * nphase;
    r.n0 = nt * BN;
    r.q0 = r.mt * BM;
    return r;
  };

  // tap geometry (depends on the phase of a transposed conv; identical for all tiles of a plain conv)
  struct Taps {
    int ntaps, off0, dstep, widx0, wstep, min_off, halo;
  };
  auto taps_of = [&](int phase) __attribute__((always_inline)) {
    Taps t;
    if (a.mode == KK_CONV) {
      t.ntaps = a.Kw; t.off0 = -a.pad; t.dstep = a.dil; t.widx0 = 0; t.wstep = 1;
    } else {
      const int k0 = (phase + a.pad) % a.stride;
      t.ntaps = (a.Kw - k0 + a.stride - 1) / a.stride;
      t.off0 = (phase + a.pad - k0) / a.stride; t.dstep = -1; t.widx0 = k0; t.wstep = a.stride;
    }
    t.min_off = t.dstep >= 0 ? t.off0 : t.off0 + (t.ntaps - 1) * t.dstep;
    t.halo = (t.ntaps - 1) * (t.dstep >= 0 ? t.dstep : -t.dstep);
    return t;
  };

  const int nchunk = a.CinP / CK;
  const int first_tile = blockIdx.x;
  if (first_tile >= total_tiles) return;

  uint4 xreg[XREG];
  uint4 wA0, wA1, wB0, wB1;  // two W register sets (2 x 16-byte chunks per thread each)
  float4 preg = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned xok = 0;

  auto load_x = [&](const TileId& T, const Taps& tp, int chunk) __attribute__((always_inline)) {
    const int Lin = kk_len(a.lin, T.b);
    const int lin_hi = Lin > 0 ? Lin - 1 : 0;
    const bf16_t* xb = a.x + (long long)T.b * a.xbs;
    const int xrows = BM + tp.halo;
    xok = 0;
    if (NRM && tid < 48) {
      const int which = tid >> 4, c = chunk * CK + (tid & 15) * 4;
      if (which == 0) preg = *(const float4*)(a.nrm_a + (long long)T.b * a.nrm_stride + c);
      else if (which == 1) preg = *(const float4*)(a.nrm_b + (long long)T.b * a.nrm_stride + c);
      else if (a.nrm_act == KK_ACT_SNAKE) {
        preg.x = (c + 0) < a.nrm_C ? a.nrm_alpha[c + 0] : 1.0f;
        preg.y = (c + 1) < a.nrm_C ? a.nrm_alpha[c + 1] : 1.0f;
        preg.z = (c + 2) < a.nrm_C ? a.nrm_alpha[c + 2] : 1.0f;
        preg.w = (c + 3) < a.nrm_C ? a.nrm_alpha[c + 3] : 1.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < XREG; ++i) {
      const int id = i * NT + tid;
      const int r = id >> 3, c8 = (id & 7) * 8;
      int row = T.q0 + tp.min_off + r;
      const bool ok0 = row >= 0 && r < xrows;
      if (a.in_shift) row >>= a.in_shift;
      if (ok0 && row < Lin) xok |= 1u << i;
      const int rc = row < 0 ? 0 : (row > lin_hi ? lin_hi : row);
      xreg[i] = *(const uint4*)(xb + (long long)rc * a.ldx + chunk * CK + c8);
    }
    asm volatile("" ::: "memory");
  };
  auto store_p = [&](int slab) __attribute__((always_inline)) {
    if (NRM && tid < 48) *(float4*)(Ps + (slab & 1) * 3 * CK + (tid >> 4) * CK + (tid & 15) * 4) = preg;
  };
  auto store_x = [&](const Taps& tp, int slab) __attribute__((always_inline)) {
    const int xrows = BM + tp.halo;
    float pa[8], pb[8], pl[8];
    if (NRM) {
      const float* pt = Ps + (slab & 1) * 3 * CK + (tid & 7) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        pa[k] = pt[k];
        pb[k] = pt[CK + k];
        pl[k] = pt[2 * CK + k];
      }
    }
#pragma unroll
    for (int i = 0; i < XREG; ++i) {
      const int id = i * NT + tid;
      const int r = id >> 3, c8 = (id & 7) * 8;
      if (r < xrows) {
        U16 t;
        const unsigned msk = (xok >> i) & 1u ? 0xFFFFFFFFu : 0u;
        t.u = xreg[i];
        if (NRM) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float y = __builtin_fmaf((float)t.h[k], pa[k], pb[k]);
            if (a.nrm_act == KK_ACT_SNAKE) {
              const float sn = __sinf(pl[k] * y);
              y = y + __builtin_amdgcn_rcpf(pl[k]) * (sn * sn);
            } else if (a.nrm_act == KK_ACT_LRELU) {
              y = y > 0.f ? y : y * a.nrm_slope;
            }
            t.h[k] = (bf16_t)y;
          }
        }
        t.u = make_uint4(t.u.x & msk, t.u.y & msk, t.u.z & msk, t.u.w & msk);
        if (a.in_slope != 1.0f) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float f = (float)t.h[k];
            t.h[k] = (bf16_t)(f > 0.f ? f : f * a.in_slope);
          }
        }
        *(uint4*)(Xs + r * XLD + c8) = t.u;
      }
    }
  };
  // W tile of (tile T, iteration it): [128 n][64 ci]; 1024 chunks of 16 B, two per thread
  auto w_ptr = [&](const TileId& T, const Taps& tp, int it) __attribute__((always_inline)) {
    const int chunk = it / tp.ntaps, tap = it - chunk * tp.ntaps;
    return a.w + ((long long)(tp.widx0 + tap * tp.wstep) * a.CoutP + T.n0) * a.CinP + chunk * CK + (long long)(tid >> 3) * a.CinP + (tid & 7) * 8;
  };
  const long long wstep64 = (long long)64 * a.CinP;
  const int wdst = (tid >> 3) * XLD + (tid & 7) * 8;

  // ---- pipeline state: the flattened sequence of (tile, iteration) pairs of this workgroup
  TileId T = decode(first_tile);
  Taps tp = taps_of(T.phase);
  int nit = nchunk * tp.ntaps;
  // helper to advance (tile, it) by one iteration in the flattened order
  auto next_pos = [&](int tile, int it, int& ntile, int& nitr) __attribute__((always_inline)) {
    const TileId Tt = decode(tile < total_tiles ? tile : first_tile);
    const Taps tt = taps_of(Tt.phase);
    const int n = nchunk * tt.ntaps;
    if (it + 1 < n) { ntile = tile; nitr = it + 1; }
    else { ntile = tile + gridDim.x; nitr = 0; }
  };

  // prologue: X slab 0, W(0) -> LDS ; W(1) in set B, W(2) in set A
  load_x(T, tp, 0);
  {
    const bf16_t* p0 = w_ptr(T, tp, 0);
    wA0 = *(const uint4*)p0;
    wA1 = *(const uint4*)(p0 + wstep64);
    asm volatile("" ::: "memory");
  }
  store_p(0);
  *(uint4*)(Ws0 + wdst) = wA0;
  *(uint4*)(Ws0 + wdst + 64 * XLD) = wA1;
  __syncthreads();
  store_x(tp, 0);
  __syncthreads();
  int t1, i1, t2, i2;  // positions of global iterations +1 and +2
  next_pos(first_tile, 0, t1, i1);
  next_pos(t1, i1, t2, i2);
  if (t1 < total_tiles) {
    const TileId Tn = decode(t1);
    const Taps tn = taps_of(Tn.phase);
    const bf16_t* p = w_ptr(Tn, tn, i1);
    wB0 = *(const uint4*)p;
    wB1 = *(const uint4*)(p + wstep64);
    asm volatile("" ::: "memory");
  }
  if (t2 < total_tiles) {
    const TileId Tn = decode(t2);
    const Taps tn = taps_of(Tn.phase);
    const bf16_t* p = w_ptr(Tn, tn, i2);
    wA0 = *(const uint4*)p;
    wA1 = *(const uint4*)(p + wstep64);
    asm volatile("" ::: "memory");
  }

  int t3, i3;  // position of global iteration gi + 3 (rolls forward by one per iteration)
  next_pos(t2, i2, t3, i3);

  const int arow = wr * 64 + (lane & 31);
  const int brow = wc * 64 + (lane & 31);
  const int kofs = 8 * (lane >> 5);
  int gi = 0;    // global iteration counter of this workgroup (parity selects W buffer / register set)
  int slab = 0;  // global slab counter (parity selects the parameter table)

  constexpr int NPASS = BM / 64;
  uint4 rres[2 * NPASS];  // residual rows of the current tile: 2 per 64-row pass

  for (int tile = first_tile; tile < total_tiles; tile += gridDim.x) {
    T = decode(tile);
    tp = taps_of(T.phase);
    nit = nchunk * tp.ntaps;
    const int Lout = kk_len(a.lout, T.b);
    const int n = T.n0 + (tid & 15) * 8;
    const int nc = n < a.Cout ? n : 0;
    const int lo_hi = a.Lo_rows - 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int it = 0; it < nit; ++it, ++gi) {
      const int chunk = it / tp.ntaps, tap = it - chunk * tp.ntaps;
      const bf16_t* Ws = (gi & 1) ? Ws1 : Ws0;
      const bool last_tap = tap == tp.ntaps - 1;
      const bool last_it = it == nit - 1;
      const int ntile = tile + gridDim.x;
      // prefetch the next X slab at the first tap of this slab (next chunk of this tile, or chunk 0 of the next tile)
      bool have_next_slab = false;
      if (tap == 0) {
        if (chunk + 1 < nchunk) {
          load_x(T, tp, chunk + 1);
          have_next_slab = true;
        } else if (ntile < total_tiles) {
          const TileId Tn = decode(ntile);
          load_x(Tn, taps_of(Tn.phase), 0);
          have_next_slab = true;
        }
      }
      (void)have_next_slab;
      if (last_it && a.res) {  // residual rows of this tile: requested now, consumed after the last MFMAs
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = (i * NT + tid) >> 4;
            const int q = T.q0 + p * 64 + row;
            const int op = a.mode == KK_CONV ? q : T.phase + a.stride * q;
            const int opc = op < 0 ? 0 : (op > lo_hi ? lo_hi : op);
            rres[p * 2 + i] = *(const uint4*)((const bf16_t*)a.res + (long long)T.b * a.rbs + (long long)opc * a.ldr + nc);
          }
        asm volatile("" ::: "memory");
      }

      const int shift = (tp.off0 + tap * tp.dstep) - tp.min_off;
      const bf16_t* xa = Xs + (arow + shift) * XLD + kofs;
      const bf16_t* wb = Ws + brow * XLD + kofs;
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        const bf16x8 b0 = *(const bf16x8*)(wb + ks * 16);
        const bf16x8 b1 = *(const bf16x8*)(wb + 32 * XLD + ks * 16);
        const bf16x8 a0 = *(const bf16x8*)(xa + ks * 16);
        const bf16x8 a1 = *(const bf16x8*)(xa + 32 * XLD + ks * 16);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      }

      // ---- end of iteration: W(gi+1) from its register set into the other LDS buffer, then request W(gi+3)
      {
        bf16_t* Wn = (gi & 1) ? Ws0 : Ws1;
        if (gi & 1) {  // W(gi+1) lives in set A when gi is odd (A holds even global iterations)
          *(uint4*)(Wn + wdst) = wA0;
          *(uint4*)(Wn + wdst + 64 * XLD) = wA1;
        } else {
          *(uint4*)(Wn + wdst) = wB0;
          *(uint4*)(Wn + wdst + 64 * XLD) = wB1;
        }
      }
      if (tap == 0) store_p(slab + 1);
      if (last_tap) {
        __syncthreads();  // every wave is done with the X slab (and, at the end of a tile, with all MFMAs of the tile)
        if (last_it) {
          // ================= epilogue of `tile`: 4 passes of 64 rows through the (dead) X slab =====================
          float bias8[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) bias8[k] = a.bias ? a.bias[T.n0 + (tid & 15) * 8 + k] : 0.f;
          float st_s[8], st_q[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) st_s[k] = st_q[k] = 0.f;
          bf16_t* ob = (bf16_t*)a.out + (long long)T.b * a.obs;
#pragma unroll
          for (int p = 0; p < NPASS; ++p) {
            if (p > 0) __syncthreads();
            if (wr == p) {
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                  const int col = wc * 64 + ni * 32 + (lane & 31);
                  const int rbase = mi * 32 + 4 * (lane >> 5);
#pragma unroll
                  for (int r = 0; r < 16; ++r) Cs[(rbase + (r & 3) + 8 * (r >> 2)) * CLD + col] = acc[mi][ni][r];
                }
            }
            __syncthreads();
            uint4 rold[2];
            int opv[2];
            bool wr_ok[2], live[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int row = (i * NT + tid) >> 4;
              const int q = T.q0 + p * 64 + row;
              const int op = a.mode == KK_CONV ? q : T.phase + a.stride * q;
              opv[i] = op < 0 ? 0 : (op > lo_hi ? lo_hi : op);
              wr_ok[i] = q < a.Q && op < a.Lo_rows && n < a.Cout;
              live[i] = op < Lout;
            }
            if (a.accumulate) {
#pragma unroll
              for (int i = 0; i < 2; ++i) rold[i] = *(const uint4*)(ob + (long long)opv[i] * a.ldo + nc);
              asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int row = (i * NT + tid) >> 4;
              float v[8];
              const float4 c0 = *(const float4*)(Cs + row * CLD + (tid & 15) * 8);
              const float4 c1 = *(const float4*)(Cs + row * CLD + (tid & 15) * 8 + 4);
              v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] += bias8[k];
              if (a.act == KK_ACT_LRELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * a.act_slope;
              } else if (a.act == KK_ACT_GELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gelu_exact(v[k]);
              }
              if (a.res) {
                U16 t;
                t.u = rres[p * 2 + i];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += (float)t.h[k];
              }
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] *= a.scale;
              if (a.accumulate) {
                U16 t;
                t.u = rold[i];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += (float)t.h[k];
              }
              if (!live[i]) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
              }
              if (wr_ok[i]) {
                U16 t;
#pragma unroll
                for (int k = 0; k < 8; ++k) t.h[k] = (bf16_t)v[k];
                *(uint4*)(ob + (long long)opv[i] * a.ldo + n) = t.u;
                if (a.stat_part) {
#pragma unroll
                  for (int k = 0; k < 8; ++k) {
                    const float r = (float)t.h[k];
                    st_s[k] += r;
                    st_q[k] = __builtin_fmaf(r, r, st_q[k]);
                  }
                }
              }
            }
          }
          if (a.stat_part) {
            __syncthreads();
            float* red = Cs;  // [8 waves][2][128]
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              st_s[k] += __shfl_xor(st_s[k], 16);
              st_s[k] += __shfl_xor(st_s[k], 32);
              st_q[k] += __shfl_xor(st_q[k], 16);
              st_q[k] += __shfl_xor(st_q[k], 32);
            }
            if (lane < 16) {
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                red[(wave * 2 + 0) * 128 + lane * 8 + k] = st_s[k];
                red[(wave * 2 + 1) * 128 + lane * 8 + k] = st_q[k];
              }
            }
            __syncthreads();
            if (tid < 256) {
              const int which = tid >> 7, col = tid & 127;
              if (T.n0 + col < a.Cout) {
                float v = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) v += red[(w8 * 2 + which) * 128 + col];
                const int stile = T.mt * nphase + T.phase;
                a.stat_part[(((long long)T.b * a.stat_ntiles + stile) * 2 + which) * a.Cout + T.n0 + col] = v;
              }
            }
          }
          __syncthreads();  // the X slab region is free again
        }
        // stage the prefetched slab (next chunk of this tile or chunk 0 of the next tile)
        if (chunk + 1 < nchunk) store_x(tp, slab + 1);
        else if (ntile < total_tiles) {
          const TileId Tn = decode(ntile);
          store_x(taps_of(Tn.phase), slab + 1);
        }
        ++slab;
      }
      __syncthreads();
      if (t3 < total_tiles) {
        const TileId Tn = decode(t3);
        const Taps tn = taps_of(Tn.phase);
        const bf16_t* p = w_ptr(Tn, tn, i3);
        if (gi & 1) {  // set A was just drained (it held W(gi+1)); it now receives W(gi+3)
          wA0 = *(const uint4*)p;
          wA1 = *(const uint4*)(p + wstep64);
        } else {
          wB0 = *(const uint4*)p;
          wB1 = *(const uint4*)(p + wstep64);
        }
        asm volatile("" ::: "memory");
      }
      {
        int tn, in_;
        next_pos(t3, i3, tn, in_);
        t3 = tn;
        i3 = in_;
      }
    }
  }
}

}  // namespace

// EXPERIMENTAL, off by default.  Measured on MI355X (tools/bench_conv.py, B=32): correct on every test, but 1.8-2.2x SLOWER
// than the 128-row kernel (k3 1.46 vs 0.67 ms, k11 2.15 vs 1.20 ms): 71-95 spilled VGPRs inside the persistent loop, a
// 4-pass epilogue behind 8-wave barriers and no second workgroup to cover it.  Kept for the next round (needs hand
// register budgeting); enable with kk_debug_set_mfma3(1) or KK_MFMA3=1.
static int g_mfma3_on = -1;
void kk_set_mfma3(int on) { g_mfma3_on = on ? 1 : 0; }
bool kk_mfma3_usable(const KKMfmaArgs& a, int out_dtype) {
  if (g_mfma3_on < 0) {
    const char* e = getenv("KK_MFMA3");
    g_mfma3_on = e ? (atoi(e) != 0) : 0;
  }
  return g_mfma3_on && out_dtype == KK_BF16 && a.Q >= 2048;
}

int kk_launch_conv_mfma3(const KKMfmaArgs& a, int B, hipStream_t st) {
  if (a.Q <= 0 || B <= 0) return 0;
  static bool attr_done = false;
  static int ncu = 256;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_mfma3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv_mfma3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
    attr_done = true;
  }
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  const int ntm = kk_cdiv(a.Q, BM), ntn = a.CoutP / BN;
  const long long total = (long long)ntm * ntn * B * nphase;
  if (total > 0x7fffffffLL) return kk_fail("conv_mfma3: too many tiles");
  const int grid = (int)(total < ncu ? total : ncu);
  if (a.nrm_a)
    hipLaunchKernelGGL(conv_mfma3_kernel<true>, dim3(grid), dim3(NT), LDS_BYTES, st, a, ntm, ntn, (int)total);
  else
    hipLaunchKernelGGL(conv_mfma3_kernel<false>, dim3(grid), dim3(NT), LDS_BYTES, st, a, ntm, ntn, (int)total);
  KK_CHECK_LAUNCH();
  return 0;
}
