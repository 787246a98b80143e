// Variant 5 of the bf16 MFMA implicit-GEMM convolution: WAVE-SPECIALISED and PERSISTENT (stride-1 convolutions; see kk_conv_mfma4.hip for the
// arithmetic, the fragment-order weight pack and the epilogue rules, which are the same here).
//
// What variants 2 / 4 cannot hide: a tile is three serial phases -- (1) fetch the X slab, apply AdaIN + Snake, store it to LDS; (2) the
// MFMA loop; (3) accumulators -> LDS -> bias / residual / statistics -> HBM -- and with two 4-wave workgroups per CU the matrix pipe idles
// whenever both are in (1) or (3): measured 12-23 % MFMA-busy per wave, prologue + epilogue 30-55 % of a tile.
//
// Here ONE 8-wave workgroup per CU walks a contiguous run of tiles:
//   * waves 0-3 (one per SIMD) do nothing but the MFMA loop: A fragments from the X slab in LDS, B fragments straight from global memory
//     in fragment order (one iteration ahead, as in variant 4); at the end of a tile they drop their accumulators into a full-tile fp32 C
//     buffer in LDS and start the next tile at once;
//   * waves 4-7 (the SIMDs' second waves) do everything else, one step ahead: they fetch the NEXT slab (next channel slab of the tile, or
//     the first slab of the next tile) into registers, apply the fused AdaIN + Snake / LeakyReLU there, and run the epilogue of the
//     PREVIOUS tile out of the C buffer (bias, activation, residual, scale, accumulate, bf16 rounding, per-tile statistics, coalesced
//     16-byte stores) -- VALU, LDS and memory instructions that issue beside the other wave's MFMAs on the same SIMD;
//   * two workgroup barriers per slab: [A] the MFMA waves are done with the slab in LDS and the next one is ready in the service waves'
//     registers -> the service waves store it (the MFMA waves dump their accumulators meanwhile if the tile is finished) -> [B].
//     Barriers are raw s_barrier + lgkmcnt(0): the weight loads in flight are NOT drained.
// LDS: X slab 34.8 KB + C tile 96 KB + statistics scratch 4 KB = 135 KB, one workgroup per CU, 2 waves per SIMD (<= 256 VGPRs each).
#include <stdlib.h>

#include "kk_common.h"
#include "kk_kernels.h"
#include "kk_conv_mfma_shared.h"

#ifdef KK_MFMA_TRACE
// phase timing (python mlx-audio_amd/build.py --trace; never compiled into the shipped library): cycles of wave 0 (MFMA role) and wave 4
// (service role) of every workgroup.  Slots: 0 MFMA loop total, 1 MFMA waiting at [A], 2 MFMA between [A] and [B] (dump + wait), 3 service
// loop total, 4 service waiting at [A], 5 service between [A] and [B] (slab store + wait), 6 service epilogue share, 7 service load + transform
__device__ unsigned long long kk_mfma5_trace_acc[256][8];
#define TR5_NOW() (__builtin_readcyclecounter())
#define TR5_ADD(slot, v) \
  do { if ((threadIdx.x & 255) == 0) atomicAdd(&kk_mfma5_trace_acc[blockIdx.x & 255][slot], (unsigned long long)(v)); } while (0)
extern "C" int kk_debug_mfma5_trace(unsigned long long* out8, int reset) {
  static unsigned long long h[256][8];
  if (out8) {
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kk_mfma5_trace_acc), sizeof(h)) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    for (int r = 0; r < 256; ++r)
      for (int k = 0; k < 8; ++k) out8[k] += h[r][k];
  }
  if (reset) {
    for (int r = 0; r < 256; ++r)
      for (int k = 0; k < 8; ++k) h[r][k] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(kk_mfma5_trace_acc), h, sizeof(h)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define TR5_NOW() 0ull
#define TR5_ADD(slot, v) do { } while (0)
#endif
namespace {
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128, CK = 64, BM = 192, WM = 96, MI16 = WM / 16;
#ifndef KK_MFMA16
constexpr int MI = 3;
#endif
constexpr int XLD = KK_XLD;   // elements per LDS row (160 B / 144 B: conflict-free ds_read_b128 fragments of the 16- / 32-row MFMA shape, kk_conv_mfma4.hip)
constexpr int MAX_HALO = 50;  // (Kw-1)*dil of the largest resblock conv (k 11, dilation 5)
constexpr int CLD = BN;       // fp32 C tile pitch
constexpr int XROWS = BM + MAX_HALO;
constexpr int XS_BYTES = XROWS * XLD * 2;
constexpr int CS_BYTES = BM * CLD * 4;
constexpr int RED_BYTES = 4 * 2 * BN * 4;
constexpr int MAX_B5 = 256;  // utterances per launch (the LDS table of output lengths)
constexpr int MAX_C5 = 1280;  // channels per side (the LDS tables of bias and Snake alpha)
constexpr int LDS5_BYTES = XS_BYTES + CS_BYTES + RED_BYTES + 2 * MAX_B5 * 4 + 2 * MAX_C5 * 4;
constexpr int XREG = (XROWS * 8 + 255) / 256;  // 16-byte chunks of the slab per service thread
constexpr int NTASK = BM * 16 / 256;           // epilogue row tasks per service thread and tile (12)


// raw workgroup barrier: LDS traffic of this wave is complete, global loads / stores stay in flight
#define KK_BAR5()                                        \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
  } while (0)

// where the epilogue's lanes outside the output store (one 16-byte slot per service thread of 256 workgroups; never read)
__device__ uint4 g_dump5[256 * 256];

struct Item {
  int b, q0, n0, nb;
  int Lin, Lout;
  bool live;
};

// NRM: 0 = raw input, 1 = AdaIN + Snake while staging, 2 = AdaIN + LeakyReLU(nrm_slope; 1 = identity) while staging
// TPP: epilogue row tasks of the previous tile that ride in one slab period (12 / min(slabs per tile, 4): the last 12 / TPP periods of a tile)
// ACC: the output rows are read and added to (Generator's sum over the three resblocks): its own kernel, the rows cost 4 * TPP registers
template <int NRM, int TPP, bool ACC>
__global__ __launch_bounds__(512, 2) void conv_mfma5_kernel(KKMfmaArgs a, int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xs = (bf16_t*)smem;
  float* Cs = (float*)(smem + XS_BYTES);
  float* red = (float*)(smem + XS_BYTES + CS_BYTES);  // [4 waves][2][128]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mfma_role = __builtin_amdgcn_readfirstlane(wave) < 4;
  const int ntaps = a.Kw, off0 = -a.pad, dstep = a.dil;
  const int halo = (ntaps - 1) * dstep, xrows = BM + halo;
  const int nchunk = a.CinP / CK, nit = nchunk * ntaps;
  const int ntx = kk_cdiv(a.Q, BM), nby = a.CoutP / BN;
  const int total = B * ntx * nby;
  // Tiles of this workgroup: v, v + G, v + 2G, ...  At any moment the grid works on G neighbouring tiles (the column blocks of a row tile and
  // the row tiles sharing a halo are in flight together -> one HBM read of the input rows), and tiles past an utterance's end (cheap: zero
  // stores only) spread evenly over the workgroups instead of emptying the range of a few.  v renumbers the workgroups so that the ones
  // of one XCD (blockIdx % 8) own neighbouring tiles: each XCD has its own L2.
  const int G = gridDim.x;
  const int v = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (v >= total) return;  // (whole workgroup; the launcher's grid is <= total)
  const int ntile = (total - v + G - 1) / G;
  // [B] output / input lengths: a tile is decoded from LDS (through global memory each decode was a dependent ~2 us load: 8.6 k cycles per tile)
  int* s_lout = (int*)(smem + XS_BYTES + CS_BYTES + RED_BYTES);
  int* s_lin = s_lout + MAX_B5;
  // bias [CoutP] and Snake alpha [CinP] (1 past the real channels) as well: every global load a service wave uses RIGHT AWAY waits for the
  // input rows it has just requested too (vmcnt retires in order), which is the latency the period is built to hide
  float* s_bias = (float*)(s_lin + MAX_B5);
  float* s_alpha = s_bias + MAX_C5;
  for (int i = tid; i < B; i += 512) {
    s_lout[i] = kk_len(a.lout, i);
    s_lin[i] = kk_len(a.lin, i);
  }
  for (int i = tid; i < a.CoutP; i += 512) s_bias[i] = a.bias ? a.bias[i] : 0.f;  // (bias has CoutP entries)
  if (NRM == 1)
    for (int i = tid; i < a.CinP; i += 512) s_alpha[i] = i < a.nrm_C ? a.nrm_alpha[i] : 1.0f;
  __syncthreads();

  auto decode = [&](int i) __attribute__((always_inline)) -> Item {
    Item t;
    t.nb = i % nby;
    const int r = i / nby;
    const int bx = r % ntx;
    t.b = r / ntx;
    t.q0 = bx * BM;
    t.n0 = t.nb * BN;
    t.Lin = s_lin[t.b];
    t.Lout = s_lout[t.b];
    t.live = t.q0 < t.Lout;
    return t;
  };

  if (mfma_role) {
    // ============================================================ MFMA waves
    const int wr = wave >> 1, wc = wave & 1;
#ifdef KK_MFMA16
    f32x4 acc[MI16][4];
#pragma unroll
    for (int i = 0; i < MI16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#endif
    // pack order of the weights: [tap][n block][chunk][wc][ks][ni][lane] x 16 bytes (kk_mfma4_pack_index; [wc][ni][ks] with -DKK_MFMA32)
    const uint4* wbase = (const uint4*)a.wf + (wc * 2) * 4 * 64 + lane;
    // The B fragments of iteration it + 2 are requested right behind the MFMAs of iteration it (two named register sets, even / odd it;
    // nit is even): with ONE MFMA wave per SIMD nothing else hides the L2 latency of a weight load -- one iteration ahead (variant 4's
    // distance, fine with two MFMA waves per SIMD) left this wave waiting ~700 cycles per iteration.  The request cursor (ptap, pchunk,
    // pnb) runs on into the next tile of this workgroup; no loads on any other path of the loop (hipcc's wait counts assume the fewest
    // loads in flight over all paths into a use: a load-free dead iteration between live ones made every iteration wait for the fragments
    // requested one iteration earlier).
    uint4 qa00, qa01, qa02, qa03, qa10, qa11, qa12, qa13, qb00, qb01, qb02, qb03, qb10, qb11, qb12, qb13;
    int ptap = 0, pchunk = 0, pnb = 0, pj = 0;
    auto nb_of = [&](int j) __attribute__((always_inline)) -> int { return (v + (j < ntile ? j : ntile - 1) * G) % nby; };
    auto cursor = [&]() __attribute__((always_inline)) -> const uint4* {
      return wbase + ((long long)(ptap * nby + pnb) * nchunk + pchunk) * 1024;
    };
    auto advance = [&]() __attribute__((always_inline)) {
      if (++ptap == ntaps) {
        ptap = 0;
        if (++pchunk == nchunk) {
          pchunk = 0;
          pnb = nb_of(++pj);
        }
      }
    };
#ifdef KK_MFMA16
    const int arow = wr * WM + (lane & 15);
    const int kofs = 8 * (lane >> 4);
#else
    const int arow = wr * WM + (lane & 31);
    const int kofs = 8 * (lane >> 5);
#endif
    KK_BAR5();  // [P] slab 0 of the first tile is in LDS (and s_lout is written)
    if (!(a.dbg & 16)) __builtin_amdgcn_s_setprio(3);  // beside a service wave on the same SIMD, the MFMA wave's LDS reads / weight loads issue first
    const unsigned long long trm0 = TR5_NOW();
    (void)trm0;
#ifdef KK_MFMA16
#define KK_KSTEP5(KS, B0, B1, B2, B3)                                                                           \
    {                                                                                                            \
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, B0), b1 = __builtin_bit_cast(bf16x8, B1);                   \
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, B2), b3 = __builtin_bit_cast(bf16x8, B3);                   \
      _Pragma("unroll") for (int mi = 0; mi < MI16; ++mi) {                                                      \
        const bf16x8 av = *(const bf16x8*)(xa + mi * 16 * XLD + (KS) * 32);                                     \
        acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b0, acc[mi][0], 0, 0, 0);                       \
        acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b1, acc[mi][1], 0, 0, 0);                       \
        acc[mi][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b2, acc[mi][2], 0, 0, 0);                       \
        acc[mi][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b3, acc[mi][3], 0, 0, 0);                       \
      }                                                                                                          \
      B0 = fn[((KS) * 4 + 0) * 64];                                                                              \
      B1 = fn[((KS) * 4 + 1) * 64];                                                                              \
      B2 = fn[((KS) * 4 + 2) * 64];                                                                              \
      B3 = fn[((KS) * 4 + 3) * 64];                                                                              \
    }
#define KK_STEPS5(Q00, Q01, Q02, Q03, Q10, Q11, Q12, Q13)                                                       \
      KK_KSTEP5(0, Q00, Q01, Q02, Q03)                                                                           \
      KK_KSTEP5(1, Q10, Q11, Q12, Q13)                                                                           \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                         \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
        _Pragma("unroll") for (int j = 0; j < MI16; ++j) {                                                       \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                     \
          if (ks * MI16 + j + 2 < 2 * MI16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   \
        }                                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                       \
      }
#define KK_CDUMP5()                                                                                              \
          _Pragma("unroll") for (int mi = 0; mi < MI16; ++mi)                                                    \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) {                                                   \
              const int col = wc * 64 + ni * 16 + (lane & 15);                                                   \
              const int rbase = wr * WM + mi * 16 + 4 * (lane >> 4);                                             \
              _Pragma("unroll") for (int r = 0; r < 4; ++r) Cs[(rbase + r) * CLD + col] = acc[mi][ni][r];       \
              acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};                                                           \
            }
#else
#define KK_KSTEP5(KS, B0, B1)                                                                                   \
    {                                                                                                            \
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, B0), b1 = __builtin_bit_cast(bf16x8, B1);                   \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                        \
        const bf16x8 av = *(const bf16x8*)(xa + mi * 32 * XLD + (KS) * 16);                                     \
        acc[mi][0] = kk_mfma32<(KS)>(av, b0, acc[mi][0]);                       \
        acc[mi][1] = kk_mfma32<(KS)>(av, b1, acc[mi][1]);                       \
      }                                                                                                          \
      B0 = fn[(KS) * 64];                                                                                        \
      B1 = fn[(4 + (KS)) * 64];                                                                                 \
    }
#define KK_STEPS5(Q00, Q01, Q02, Q03, Q10, Q11, Q12, Q13)                                                       \
      KK_KSTEP5(0, Q00, Q10)                                                                                     \
      KK_KSTEP5(1, Q01, Q11)                                                                                     \
      KK_KSTEP5(2, Q02, Q12)                                                                                     \
      KK_KSTEP5(3, Q03, Q13)                                                                                     \
      __builtin_amdgcn_sched_group_barrier(0x100, MI, 0);                                                        \
      _Pragma("unroll") for (int ks = 0; ks < CK / 16; ++ks) {                                                   \
        _Pragma("unroll") for (int j = 0; j < MI; ++j) {                                                         \
          __builtin_amdgcn_sched_group_barrier(0x008, KK_MFMA_PER, 0);                                           \
          if (ks < CK / 16 - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
        }                                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, MI * KK_MFMA_PER, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                       \
      }
#define KK_CDUMP5()                                                                                              \
          _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                      \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                   \
              const int col = wc * 64 + ni * 32 + (lane & 31);                                                   \
              const int rbase = wr * WM + mi * 32 + 4 * (lane >> 5);                                             \
              _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                   \
                Cs[(rbase + (r & 3) + 8 * (r >> 2)) * CLD + col] = acc[mi][ni][r];                               \
                acc[mi][ni][r] = 0.f;                                                                            \
              }                                                                                                  \
            }
#endif
#define KK_ITER5(Q00, Q01, Q02, Q03, Q10, Q11, Q12, Q13)                                                                                  \
    {                                                                                                                                     \
      const uint4* fn = cursor();                                                                                                         \
      const bf16_t* xa = Xs + (arow + tap * dstep) * XLD + kofs; /* row shift of this tap inside the slab */                              \
      KK_STEPS5(Q00, Q01, Q02, Q03, Q10, Q11, Q12, Q13)                                                                                  \
      asm volatile("" ::: "memory");                                                                                                      \
      advance();                                                                                                                          \
      if (++tap == ntaps) {                                                                                                               \
        tap = 0;                                                                                                                          \
        const unsigned long long ta0 = TR5_NOW();                                                                                         \
        KK_BAR5(); /* [A] done with this slab */                                                                                          \
        const unsigned long long ta1 = TR5_NOW();                                                                                         \
        TR5_ADD(1, ta1 - ta0);                                                                                                            \
        (void)ta0; (void)ta1;                                                                                                             \
        if (++chunk == nchunk) { /* the finished tile -> C buffer (the service waves finished the previous tile's C before [A]) */        \
          chunk = 0;                                                                                                                      \
          KK_CDUMP5()                                                                                                                     \
        }                                                                                                                                 \
        KK_BAR5(); /* [B] next slab in LDS (and, after a tile's last slab, its C complete) */                                             \
        TR5_ADD(2, TR5_NOW() - ta1);                                                                                                      \
      }                                                                                                                                   \
    }
    bool chain = false;  // the two register sets hold iterations 0 and 1 of the tile about to start
    int tap = 0, chunk = 0;
    for (int j = 0; j < ntile; ++j) {
      const int item = v + j * G;
      const int r = item / nby;
      const bool live = (r % ntx) * BM < s_lout[r / ntx];
      if (!live) {  // a tile past the utterance's end: nothing to multiply, the barriers only
        for (int c = 0; c < nchunk; ++c) {
          KK_BAR5();
          KK_BAR5();
        }
        chain = false;
        continue;
      }
      if (!chain) {  // the first tile, or the first one behind a dead tile: both sets, then wait (rare)
        pj = j; pnb = nb_of(j); ptap = 0; pchunk = 0;
        const uint4* f0 = cursor();
        advance();
        const uint4* f1 = cursor();
        advance();
        qa00 = f0[0 * 64]; qa01 = f0[1 * 64]; qa02 = f0[2 * 64]; qa03 = f0[3 * 64];
        qa10 = f0[4 * 64]; qa11 = f0[5 * 64]; qa12 = f0[6 * 64]; qa13 = f0[7 * 64];
        qb00 = f1[0 * 64]; qb01 = f1[1 * 64]; qb02 = f1[2 * 64]; qb03 = f1[3 * 64];
        qb10 = f1[4 * 64]; qb11 = f1[5 * 64]; qb12 = f1[6 * 64]; qb13 = f1[7 * 64];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        chain = true;
      }
      for (int it = 0; it < nit; it += 2) {
        KK_ITER5(qa00, qa01, qa02, qa03, qa10, qa11, qa12, qa13)
        KK_ITER5(qb00, qb01, qb02, qb03, qb10, qb11, qb12, qb13)
      }
    }
#undef KK_ITER5
#undef KK_KSTEP5
#undef KK_STEPS5
#undef KK_CDUMP5
    TR5_ADD(0, TR5_NOW() - trm0);
    if (a.stat_part) KK_BAR5();  // (the service waves' last statistics reduction has one more barrier: keep the counts equal)
    return;
  }

  // ================================================================ service waves
  const int stid = tid - 256, swave = wave - 4;
  const int cin_real = a.Cin > 0 ? a.Cin : a.CinP;
  // two slabs of raw input rows in flight: the set being transformed this period was requested a whole period ago
  uint4 xr0[XREG], xr1[XREG];
  unsigned xk0 = 0, xk1 = 0;
  float pa[8], pb[8], pal[8];  // AdaIN A, B and Snake alpha of this thread's 8 channels in the slab being staged

  auto load_params = [&](const Item& t, int chunk) __attribute__((always_inline)) {
    if (NRM) {
      const int c = chunk * CK + (stid & 7) * 8;
      const float4 a0 = *(const float4*)(a.nrm_a + (long long)t.b * a.nrm_stride + c), a1 = *(const float4*)(a.nrm_a + (long long)t.b * a.nrm_stride + c + 4);
      const float4 b0 = *(const float4*)(a.nrm_b + (long long)t.b * a.nrm_stride + c), b1 = *(const float4*)(a.nrm_b + (long long)t.b * a.nrm_stride + c + 4);
      pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
      pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
      if (NRM == 1) {
        const float4 l0 = *(const float4*)(s_alpha + c), l1 = *(const float4*)(s_alpha + c + 4);
        pal[0] = l0.x; pal[1] = l0.y; pal[2] = l0.z; pal[3] = l0.w; pal[4] = l1.x; pal[5] = l1.y; pal[6] = l1.z; pal[7] = l1.w;
      }
    }
  };
  auto load_x = [&](const Item& t, int chunk, uint4 (&xreg)[XREG], unsigned& xok) __attribute__((always_inline)) {
    xok = 0;
    const bf16_t* xb = a.x + (long long)t.b * a.xbs;
    const int lin_hi = t.Lin > 0 ? t.Lin - 1 : 0;
#pragma unroll
    for (int i = 0; i < XREG; ++i) {
      const int id = i * 256 + stid;
      const int r = id >> 3, c8 = (id & 7) * 8;
      int row = t.q0 + off0 + r;
      const bool ok0 = row >= 0 && r < xrows;
      if (a.in_shift) row >>= a.in_shift;
      if (ok0 && row < t.Lin) xok |= 1u << i;
      const int rc = row < 0 ? 0 : (row > lin_hi ? lin_hi : row);
      xreg[i] = *(const uint4*)(xb + (long long)rc * a.ldx + chunk * CK + c8);
    }
    asm volatile("" ::: "memory");
  };
  // the fused input transform, in registers (the slab is stored later, between the two barriers)
  auto transform_x = [&](int chunk, uint4 (&xreg)[XREG], const unsigned xok, const bool nomask) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XREG; ++i) asm volatile("" : "+v"(xreg[i].x), "+v"(xreg[i].y), "+v"(xreg[i].z), "+v"(xreg[i].w));
    if (NRM) {
      const float slope = a.nrm_slope;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float a0 = pa[2 * kk], a1 = pa[2 * kk + 1], b0 = pb[2 * kk], b1 = pb[2 * kk + 1];
        float l0 = 0.f, l1 = 0.f, i0 = 0.f, i1 = 0.f;
        if (NRM == 1) {  // snake: y + sin^2(alpha y) / alpha; v_sin_f32 takes revolutions
          l0 = pal[2 * kk] * 0.15915494309189535f;
          l1 = pal[2 * kk + 1] * 0.15915494309189535f;
          i0 = __builtin_amdgcn_rcpf(pal[2 * kk]);
          i1 = __builtin_amdgcn_rcpf(pal[2 * kk + 1]);
        }
#pragma unroll
        for (int i = 0; i < XREG; ++i) {
          const unsigned wd = kk == 0 ? xreg[i].x : kk == 1 ? xreg[i].y : kk == 2 ? xreg[i].z : xreg[i].w;
          float y0 = __builtin_fmaf(__uint_as_float(wd << 16), a0, b0);
          float y1 = __builtin_fmaf(__uint_as_float(wd & 0xFFFF0000u), a1, b1);
          if (NRM == 1) {
            const float s0 = __builtin_amdgcn_sinf(l0 * y0), s1 = __builtin_amdgcn_sinf(l1 * y1);
            y0 = __builtin_fmaf(i0 * s0, s0, y0);
            y1 = __builtin_fmaf(i1 * s1, s1, y1);
          } else {
            y0 = y0 > 0.f ? y0 : y0 * slope;
            y1 = y1 > 0.f ? y1 : y1 * slope;
          }
          const bf16x2 pk = {(bf16_t)y0, (bf16_t)y1};
          const unsigned o = __builtin_bit_cast(unsigned, pk);
          if (kk == 0) xreg[i].x = o;
          else if (kk == 1) xreg[i].y = o;
          else if (kk == 2) xreg[i].z = o;
          else xreg[i].w = o;
        }
      }
    }
    // padding rows / pad channels stay exactly zero (32-bit integer ops only, see variant 4).  A fused slab that lies inside the utterance and
    // inside the real channels (uniform over the workgroup: `nomask`) has nothing to mask -- all but the edge tiles -- and skips the ANDs.
    if (NRM != 0 && nomask) return;
    const int cfirst = chunk * CK + (stid & 7) * 8;
    unsigned cm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = (cfirst + 2 * j < cin_real ? 0x0000FFFFu : 0u) | (cfirst + 2 * j + 1 < cin_real ? 0xFFFF0000u : 0u);
#pragma unroll
    for (int i = 0; i < XREG; ++i) {
      const unsigned msk = (xok >> i) & 1u ? 0xFFFFFFFFu : 0u;
      unsigned wq[4] = {xreg[i].x & msk & cm[0], xreg[i].y & msk & cm[1], xreg[i].z & msk & cm[2], xreg[i].w & msk & cm[3]};
      if (NRM != 0) {
        // (the fused AdaIN input never carries an input LeakyReLU / ELU: Ctx::conv sets one or the other)
      } else if (a.in_act == KK_ACT_ELU) {  // nn.elu: where(x > 0, x, exp(x) - 1); elu(0) = 0 keeps the padding zero
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float lo = __uint_as_float(wq[k] << 16), hi = __uint_as_float(wq[k] & 0xFFFF0000u);
          lo = lo > 0.f ? lo : __expf(lo) - 1.0f;
          hi = hi > 0.f ? hi : __expf(hi) - 1.0f;
          const bf16x2 pk = {(bf16_t)lo, (bf16_t)hi};
          wq[k] = __builtin_bit_cast(unsigned, pk);
        }
      } else if (a.in_slope != 1.0f) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float lo = __uint_as_float(wq[k] << 16), hi = __uint_as_float(wq[k] & 0xFFFF0000u);
          lo = lo > 0.f ? lo : lo * a.in_slope;
          hi = hi > 0.f ? hi : hi * a.in_slope;
          const bf16x2 pk = {(bf16_t)lo, (bf16_t)hi};
          wq[k] = __builtin_bit_cast(unsigned, pk);
        }
      }
      xreg[i] = make_uint4(wq[0], wq[1], wq[2], wq[3]);
    }
  };
  auto store_x = [&](const uint4 (&xreg)[XREG]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XREG; ++i) {
      const int id = i * 256 + stid;
      const int r = id >> 3, c8 = (id & 7) * 8;
      if (r < xrows) *(uint4*)(Xs + r * XLD + c8) = xreg[i];
    }
  };

  // ---- epilogue of one tile out of the C buffer, TPP of this thread's 12 row tasks at a time (row = task * 16 + stid / 16, 8 output channels
  // (stid & 15) * 8 ..).  The residual / accumulate rows of a group are REQUESTED ONE SLAB PERIOD EARLY (epi_prefetch): they come from HBM,
  // and a just-in-time load is a 2 us stall per group (the first version spent 17-25 k cycles per tile there).
  v2f st_s[4], st_q[4];
  uint4* dump = g_dump5 + (((int)blockIdx.x & 255) << 8) + stid;
  uint4 rres[TPP], rold[ACC ? TPP : 1];
  auto epi_prefetch = [&](const Item& t, int tbase) __attribute__((always_inline)) {
    const bf16_t* ob = (const bf16_t*)a.out + (long long)t.b * a.obs;
    const bf16_t* rb = a.res ? (const bf16_t*)a.res + (long long)t.b * a.rbs : nullptr;
    const int n = t.n0 + (stid & 15) * 8;
    const int nc = n < a.Cout ? n : 0;
    const int lo_hi = a.Lo_rows - 1;
#pragma unroll
    for (int i = 0; i < TPP; ++i) {
      const int q = t.q0 + (tbase + i) * 16 + (stid >> 4);
      const int opc = q < 0 ? 0 : (q > lo_hi ? lo_hi : q);
      // (a wave-uniform condition; the explicit zero keeps the old value dead on the other path -- otherwise hipcc loads into temporaries
      //  and copies them behind a vmcnt(0) at the merge)
      if (rb) rres[i] = *(const uint4*)(rb + (long long)opc * a.ldr + nc);
      else rres[i] = make_uint4(0u, 0u, 0u, 0u);
      if (ACC) rold[i] = *(const uint4*)(ob + (long long)opc * a.ldo + nc);
    }
    asm volatile("" ::: "memory");
  };
  auto epi_compute = [&](const Item& t, int tbase) __attribute__((always_inline)) {
    bf16_t* ob = (bf16_t*)a.out + (long long)t.b * a.obs;
    const int n = t.n0 + (stid & 15) * 8;
    const int lo_hi = a.Lo_rows - 1;
    v2f bias2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bias2[k] = v2f{s_bias[n + 2 * k], s_bias[n + 2 * k + 1]};
    const v2f scale2 = {a.scale, a.scale}, act_slope2 = {a.act_slope, a.act_slope};
    const bool tile_full = t.live && t.q0 + BM <= t.Lout && t.q0 + BM <= a.Q && t.q0 + BM <= a.Lo_rows && t.n0 + BN <= a.Cout;  // uniform
    // C rows one task ahead (the read of task i + 1 is in flight while task i is computed); unconditional, a dead tile's are not used
    const float* crow = Cs + (tbase * 16 + (stid >> 4)) * CLD + (stid & 15) * 8;
    float4 cn0 = *(const float4*)crow, cn1 = *(const float4*)(crow + 4);
#pragma unroll
    for (int i = 0; i < TPP; ++i) {
      const int row = (tbase + i) * 16 + (stid >> 4);
      const int q = t.q0 + row;
      const int opc = q < 0 ? 0 : (q > lo_hi ? lo_hi : q);
      const bool wr_ok = q < a.Q && q < a.Lo_rows && n < a.Cout;
      const bool lv = t.live && q < t.Lout;
      const float4 c0 = cn0, c1 = cn1;
      if (i + 1 < TPP) {
        cn0 = *(const float4*)(crow + (i + 1) * 16 * CLD);
        cn1 = *(const float4*)(crow + (i + 1) * 16 * CLD + 4);
      }
      v2f v[4];
      v[0] = v2f{c0.x, c0.y}; v[1] = v2f{c0.z, c0.w}; v[2] = v2f{c1.x, c1.y}; v[3] = v2f{c1.z, c1.w};
      if (!t.live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v2f{0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += bias2[k];
      if (a.act == KK_ACT_LRELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const v2f m = v[k] * act_slope2;
          v[k].x = v[k].x > 0.f ? v[k].x : m.x;
          v[k].y = v[k].y > 0.f ? v[k].y : m.y;
        }
      } else if (a.act == KK_ACT_GELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k].x = gelu_exact(v[k].x);
          v[k].y = gelu_exact(v[k].y);
        }
      } else if (NRM == 0 && a.act == KK_ACT_GELU_TANH) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k].x = 0.5f * v[k].x * (1.0f + tanhf(0.7978845608028654f * (v[k].x + 0.044715f * (v[k].x * v[k].x * v[k].x))));
          v[k].y = 0.5f * v[k].y * (1.0f + tanhf(0.7978845608028654f * (v[k].y + 0.044715f * (v[k].y * v[k].y * v[k].y))));
        }
      }
      if (a.res) {
        const unsigned w4[4] = {rres[i].x, rres[i].y, rres[i].z, rres[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += v2f{__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xFFFF0000u)};
      }
      if (a.scale != 1.0f) {  // (uniform; 1 everywhere but the mean over the three resblocks)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= scale2;
      }
      if (ACC) {
        const unsigned w4[4] = {rold[i].x, rold[i].y, rold[i].z, rold[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += v2f{__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xFFFF0000u)};
      }
      if (a.post_slope != 0.f && a.post_slope != 1.0f) {  // (uniform) LeakyReLU of the finished value
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float mx = v[k].x * a.post_slope, my = v[k].y * a.post_slope;
          v[k].x = v[k].x > mx ? v[k].x : mx;
          v[k].y = v[k].y > my ? v[k].y : my;
        }
      }
      // rows past the utterance are stored as exact zeros (and count as zeros in the statistics); only a tile that reaches past its
      // utterance pays the selects (tile_full is uniform)
      if (!tile_full && !(lv && wr_ok)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v2f{0.f, 0.f};
      }
      unsigned w4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bf16x2 pk = {(bf16_t)v[k].x, (bf16_t)v[k].y};
        w4[k] = __builtin_bit_cast(unsigned, pk);
      }
      // The store is UNCONDITIONAL (lanes outside the output write their own slot of a dump area): stores retire through the same
      // in-order counter as loads, and a store hipcc cannot count makes the next wait for a load drain the row requests behind it.
      uint4* dst = wr_ok ? (uint4*)(ob + (long long)opc * a.ldo + n) : dump;
      *dst = make_uint4(w4[0], w4[1], w4[2], w4[3]);
      if (a.stat_part) {  // column statistics from the fp32 values before the bf16 rounding (see kk_conv_mfma_epilogue.h)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          st_s[k] += v[k];
          st_q[k] = fma2(v[k], v[k], st_q[k]);
        }
      }
      asm volatile("" ::: "memory");  // one task at a time (plus the C rows read ahead): hoisting every task's reads costs ~50 registers
    }
  };
  auto stats_reset = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st_s[k] = st_q[k] = v2f{0.f, 0.f};
  };
  // per-wave column sums of the tile just finished -> LDS (rows of one column group live in lanes l, l+16, l+32, l+48)
  auto stats_to_lds = [&]() __attribute__((always_inline)) {
    float ss[8], sq[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ss[2 * k] = st_s[k].x; ss[2 * k + 1] = st_s[k].y;
      sq[2 * k] = st_q[k].x; sq[2 * k + 1] = st_q[k].y;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ss[k] += __shfl_xor(ss[k], 16);
      ss[k] += __shfl_xor(ss[k], 32);
      sq[k] += __shfl_xor(sq[k], 16);
      sq[k] += __shfl_xor(sq[k], 32);
    }
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(swave * 2 + 0) * 128 + lane * 8 + k] = ss[k];
        red[(swave * 2 + 1) * 128 + lane * 8 + k] = sq[k];
      }
    }
  };
  // the four waves' sums, in wave order -> one deterministic partial per (utterance, tile, column)
  auto stats_to_global = [&](const Item& t) __attribute__((always_inline)) {
    const int which = stid >> 7, col = stid & 127;  // threads 0..127 -> sums, 128..255 -> sums of squares
    if (t.n0 + col < a.Cout) {
      const float v = red[(0 * 2 + which) * 128 + col] + red[(1 * 2 + which) * 128 + col] + red[(2 * 2 + which) * 128 + col] +
                      red[(3 * 2 + which) * 128 + col];
      const int tile = t.q0 / BM;
      a.stat_part[(((long long)t.b * a.stat_ntiles + tile) * 2 + which) * a.Cout + t.n0 + col] = v;
    }
  };

  // nothing to mask in a staged slab: its rows [q0 + off0, q0 + off0 + xrows) lie inside the utterance and its 64 channels are real (uniform)
  auto slab_inside = [&](const Item& t, int chunk) __attribute__((always_inline)) -> bool {
    const int r0 = t.q0 + off0, r1 = t.q0 + off0 + xrows - 1;
    return r0 >= 0 && (a.in_shift ? (r1 >> a.in_shift) : r1) < t.Lin && chunk * CK + CK <= cin_real;
  };
  // ---- prologue: slab 0 of the first tile into LDS, slab 1 requested
  Item cur = decode(v);
  if (cur.live) {
    load_x(cur, 0, xr0, xk0);
    load_params(cur, 0);
    transform_x(0, xr0, xk0, slab_inside(cur, 0));
    store_x(xr0);
    load_x(cur, 1, xr1, xk1);
  }
  KK_BAR5();  // [P]
  // the previous tile's 12 epilogue tasks ride in the LAST 12 / TPP slab periods of the current tile, TPP tasks each
  constexpr int NACT = NTASK / TPP;
  const int act0 = nchunk - NACT;  // first slab period that carries epilogue work (>= 0 by the launcher's choice of TPP)
  Item prev = cur, nxt = cur;
  bool have_prev = false, has_next_item = false;
  // One slab period: the MFMA waves multiply slab c of `cur`; here slab c + 1 (requested a period ago, register set XT) is transformed and
  // stored behind [A], slab c + 2 is requested (set XL, free since the last [A]), and TPP row tasks of the previous tile's epilogue run in
  // between.  Slabs past the tile's end are the next tile's first ones.
  auto period = [&](int c, uint4 (&XT)[XREG], unsigned& xkT, uint4 (&XL)[XREG], unsigned& xkL) __attribute__((always_inline)) {
    const bool tail = c == nchunk - 1, tail2 = c + 2 >= nchunk;
    const bool stage1 = tail ? (has_next_item && nxt.live) : cur.live;
    const bool stage2 = tail2 ? (has_next_item && nxt.live) : cur.live;
    const Item& s1 = tail ? nxt : cur;
    const Item& s2 = tail2 ? nxt : cur;
    const int c1 = tail ? 0 : c + 1, c2 = tail2 ? c + 2 - nchunk : c + 2;
    const unsigned long long t0 = TR5_NOW();
    // Every request below is UNCONDITIONAL (a slab / row group nobody needs re-reads valid rows of `cur`): a load under a condition costs a
    // register copy behind vmcnt(0) where the paths merge.  Oldest first: parameters of slab c + 1, rows of slab c + 2, then the epilogue
    // (its residual rows are older than all of these), then the transform (needs the parameters: at most the 8 row loads stay in flight).
    load_params(s1, c1);
    load_x(s2, c2, XL, xkL);
    if (!stage2) xkL = 0;
    const unsigned long long t1 = TR5_NOW();
    if (have_prev && c >= act0 && !(a.dbg & 4)) {  // (its residual rows were requested a slab period ago)
      epi_compute(prev, (c - act0) * TPP);
      if (tail && a.stat_part) stats_to_lds();
    }
    const unsigned long long t2 = TR5_NOW();
    if (stage1 && !(a.dbg & 8)) transform_x(c1, XT, xkT, slab_inside(s1, c1));
    // request the residual rows of the NEXT period's epilogue share: of `prev` inside a tile, of `cur` (the next `prev`) at its end
    {
      const int tb = tail ? 0 : (c + 1 >= act0 ? (c + 1 - act0) * TPP : 0);
      epi_prefetch(tail ? cur : prev, tb);
    }
    const unsigned long long t3 = TR5_NOW();
    KK_BAR5();  // [A] the MFMA waves are done with the slab in LDS; the previous tile's C is consumed when this is a tile's last slab
    const unsigned long long t4 = TR5_NOW();
    if (stage1) store_x(XT);
    if (have_prev && tail && a.stat_part) stats_to_global(prev);
    KK_BAR5();  // [B]
    TR5_ADD(4, t4 - t3);
    TR5_ADD(5, TR5_NOW() - t4);
    TR5_ADD(6, t2 - t1);
    TR5_ADD(7, (t1 - t0) + (t3 - t2));
    (void)t0; (void)t1; (void)t2; (void)t3; (void)t4;
  };
  const unsigned long long trs0 = TR5_NOW();
  (void)trs0;
  for (int j = 0; j < ntile; ++j) {
    cur = decode(v + j * G);
    has_next_item = j + 1 < ntile;
    nxt = has_next_item ? decode(v + (j + 1) * G) : cur;
    if (have_prev) stats_reset();
    for (int c = 0; c < nchunk; c += 2) {  // (nchunk is even: slab s of a tile always lives in register set s & 1)
      period(c, xr1, xk1, xr0, xk0);
      period(c + 1, xr0, xk0, xr1, xk1);
    }
    prev = cur;
    have_prev = true;
  }
  TR5_ADD(3, TR5_NOW() - trs0);
  // ---- drain: the last tile's epilogue (group 0 was requested at the end of the loop)
  stats_reset();
#pragma unroll
  for (int g = 0; g < NACT; ++g) {
    epi_compute(prev, g * TPP);
    if (g + 1 < NACT) epi_prefetch(prev, (g + 1) * TPP);
  }
  if (a.stat_part) {
    stats_to_lds();
    KK_BAR5();
    stats_to_global(prev);
  }
}

template <int NRM, int TPP, bool ACC>
int launch5(const KKMfmaArgs& a, int B, hipStream_t st) {
  static KKDevOnce attr_once;
  static int ncu_dev[64];  // CUs of each device this process has launched on (the persistent grid = CUs)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)conv_mfma5_kernel<NRM, TPP, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS5_BYTES);
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu_dev[dev & 63] = n;
    attr_once.done();
  }
  const int ncu = ncu_dev[dev & 63];
  const int total = B * kk_cdiv(a.Q, BM) * (a.CoutP / BN);
  const int grid = total < ncu ? total : ncu;
  hipLaunchKernelGGL((conv_mfma5_kernel<NRM, TPP, ACC>), dim3(grid), dim3(512), LDS5_BYTES, st, a, B);
  KK_CHECK_LAUNCH();
  return 0;
}
template <int NRM>
int launch5n(const KKMfmaArgs& a, int B, hipStream_t st) {
  const int nchunk = a.CinP / CK;  // slab periods per tile: the epilogue of the previous tile is spread over the last min(nchunk, 4) of them
  if (nchunk >= 4) return a.accumulate ? launch5<NRM, 3, true>(a, B, st) : launch5<NRM, 3, false>(a, B, st);
  return launch5<NRM, 6, false>(a, B, st);  // (accumulate with 2 slabs is not eligible)  // (2 or 3 slabs; single-slab convolutions stay on variant 4, kk_mfma5_eligible)
}

}  // namespace

// what variant 5 takes: stride-1 convolutions with a bf16 output and fragment-order weights (the transposed convolutions of Generator.ups
// and fp32 outputs stay on variants 4 / 2)
bool kk_mfma5_eligible(const KKMfmaArgs& a, int out_dtype) {
  return a.wf && out_dtype == KK_BF16 && a.mode == KK_CONV && a.stride == 1 && (a.Kw - 1) * a.dil <= MAX_HALO && a.CinP % (2 * CK) == 0 && a.CinP <= MAX_C5 && a.CoutP <= MAX_C5 &&
         !(a.accumulate && a.CinP < 4 * CK) &&  // (read-and-add with 6 row tasks per period: 30-60 spilled registers; variant 4 takes it)
         a.CoutP % BN == 0;
}
int kk_mfma5_tile_rows() { return BM; }

int kk_launch_conv_mfma5(const KKMfmaArgs& a, int B, int out_dtype, hipStream_t st) {
  if (a.Q <= 0 || B <= 0) return 0;
  if (!kk_mfma5_eligible(a, out_dtype) || B > MAX_B5) return kk_fail("conv_mfma5: not eligible");
  const int nrm = a.nrm_a == nullptr ? 0 : (a.nrm_act == KK_ACT_SNAKE ? 1 : 2);
  KKMfmaArgs g = a;
  if (nrm == 2 && a.nrm_act != KK_ACT_LRELU) g.nrm_slope = 1.0f;  // plain AdaIN: identity activation
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("KK_MFMA5_DBG");  // timing experiments only (wrong results): 2 no X loads, 4 no epilogue, 8 no transform
    dbg = e ? atoi(e) : 0;
  }
  g.dbg = dbg;
  if (nrm == 1) return launch5n<1>(g, B, st);
  if (nrm == 2) return launch5n<2>(g, B, st);
  return launch5n<0>(g, B, st);
}
