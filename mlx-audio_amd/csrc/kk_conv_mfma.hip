// bf16 MFMA implicit-GEMM convolution for frames-major tensors (gfx950, v_mfma_f32_32x32x16_bf16).
//
//   out[b][q][n] = sum_t sum_ci  W[t][n][ci] * f(X[b][q + off_t][ci])       (stride 1, any dilation)
//   transposed conv = `stride` independent phase convolutions with 2 taps each (polyphase form, see kk_conv.hip)
//   f = identity, LeakyReLU, or the fused AdaIN apply + Snake / LeakyReLU of the reference's resblocks
//
// GEMM view per workgroup: M = BM output rows (128 or 256), N = 128 output channels, K = taps x Cin walked in slabs of
// 64 channels.  A = X rows (positions), B = W rows (output channels); both are k-contiguous in LDS so a lane's MFMA
// fragment (8 consecutive k) is ONE ds_read_b128.  LDS rows are padded 128 B -> 144 B: 16 consecutive rows start on 16
// distinct 4-bank groups, so the b128 fragment reads are conflict-free (banks = (addr/4) % 64).
//
//   * 4 waves as 2 x 2; a wave owns WM x 64 outputs (WM = 64 or 128) = (WM/32) x 2 accumulators of 32 x 32.  WM = 128
//     re-uses every B (weight) fragment for 4 row tiles: 6 ds_read_b128 per 8 MFMAs instead of 4 per 4 -- with two
//     workgroups per CU the 64-row variant saturates the LDS read port (1 b128 read per MFMA and wave = 256 B/clk/CU),
//     and it halves the L2 traffic for W per MFMA.
//   * the X slab [BM + halo rows][64 ch] is loaded once per channel slab (register prefetch issued at the first tap of
//     the previous slab) and re-used by every tap as a shifted window
//   * W tiles [128 n][64 ci] are double-buffered in LDS and prefetched through registers one tap ahead: one barrier per tap
//   * all global loads are unconditional (clamped address + mask at use): a load under a data-dependent branch makes
//     hipcc wait vmcnt(0) right behind it, which serialises the loads (one HBM round trip each)
//   * epilogue through a 128 x 128 fp32 LDS tile per 128 rows: bias / activation / residual / scale / accumulate / length
//     mask on coalesced 16-byte rows, ONE rounding to bf16, optional per-tile column sums for the next instance norm.
#include <stdlib.h>

#include "kk_common.h"
#include "kk_kernels.h"
#include "kk_conv_mfma_shared.h"

#ifdef KK_MFMA_TRACE
// phase timing of wave 0 of every workgroup (tools/bench_conv.py --trace; never compiled into the shipped library)
__device__ unsigned long long kk_mfma_trace_acc[1024][8];  // spread over 1024 rows: same-address atomics would serialise
#define TR_NOW() (__builtin_readcyclecounter())
#define TR_ADD(slot, v) \
  do { if (threadIdx.x == 0) atomicAdd(&kk_mfma_trace_acc[(blockIdx.x + 37 * blockIdx.z) & 1023][slot], (unsigned long long)(v)); } while (0)
extern "C" int kk_debug_mfma_trace(unsigned long long* out8, int reset) {
  static unsigned long long h[1024][8];
  if (out8) {
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kk_mfma_trace_acc), sizeof(h)) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    for (int r = 0; r < 1024; ++r)
      for (int k = 0; k < 8; ++k) out8[k] += h[r][k];
  }
  if (reset) {
    for (int r = 0; r < 1024; ++r)
      for (int k = 0; k < 8; ++k) h[r][k] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(kk_mfma_trace_acc), h, sizeof(h)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define TR_NOW() 0ull
#define TR_ADD(slot, v) do { } while (0)
#endif
namespace {
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BN = 128, CK = 64;
constexpr int XLD = CK + 8;   // elements per LDS row (144 B)
constexpr int MAX_HALO = 50;  // (Kw-1)*dil of the largest resblock conv (k 11, dilation 5)
constexpr int CLD = BN;       // fp32 epilogue tile pitch: 128 x 128 x 4 B = exactly 64 KiB
constexpr int WS_BYTES = BN * XLD * 2;    // 18432 per buffer
constexpr int PS_BYTES = 2 * 3 * CK * 4;  // double-buffered AdaIN parameter table of one slab (A, B, alpha)

template <int BM>
struct Geo {
  static constexpr int XROWS = BM + MAX_HALO;
  static constexpr int XS_BYTES = XROWS * XLD * 2;
  static constexpr int MAIN_BYTES = XS_BYTES + 2 * WS_BYTES + PS_BYTES;
  static constexpr int RPP = BM / 2;  // rows per epilogue pass: one wave row group (64 or 96 rows)
  static constexpr int EPI_BYTES = RPP * CLD * 4;
  static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
  static constexpr int XREG = (XROWS * 8 + 255) / 256;  // 16-byte chunks of the slab per thread
};


union U16 {
  uint4 u;
  bf16_t h[8];
};
union U32x8 {
  uint4 u[2];
  float f[8];
};

// NRM: 0 = raw input, 1 = AdaIN + Snake while staging, 2 = AdaIN + LeakyReLU(nrm_slope; 1 = identity) while staging
template <typename TO, int WM, int NRM>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_mfma_kernel(KKMfmaArgs a) {
  constexpr int BM = 2 * WM, MI = WM / 32;
  using G = Geo<BM>;
  constexpr int XREG = G::XREG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xs = (bf16_t*)smem;
  bf16_t* Ws0 = (bf16_t*)(smem + G::XS_BYTES);
  bf16_t* Ws1 = (bf16_t*)(smem + G::XS_BYTES + WS_BYTES);
  float* Ps = (float*)(smem + G::XS_BYTES + 2 * WS_BYTES);  // [2][3][64]
  float* Cs = (float*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  const int b = blockIdx.z / nphase, phase = blockIdx.z - b * nphase;
  const int q0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int Lin = kk_len(a.lin, b), Lout = kk_len(a.lout, b);

  // taps: input row of output q for tap t is q + off0 + t*dstep ; weight slice widx0 + t*wstep
  int ntaps, off0, dstep, widx0, wstep;
  if (a.mode == KK_CONV) {
    ntaps = a.Kw; off0 = -a.pad; dstep = a.dil; widx0 = 0; wstep = 1;
  } else {
    const int k0 = (phase + a.pad) % a.stride;
    ntaps = (a.Kw - k0 + a.stride - 1) / a.stride;
    off0 = (phase + a.pad - k0) / a.stride; dstep = -1; widx0 = k0; wstep = a.stride;
  }
  const int min_off = dstep >= 0 ? off0 : off0 + (ntaps - 1) * dstep;
  const int halo = (ntaps - 1) * (dstep >= 0 ? dstep : -dstep);
  const int xrows = BM + halo;

  const int op_first = a.mode == KK_CONV ? q0 : phase + a.stride * q0;
  const bool tile_live = op_first < Lout;  // uniform over the workgroup

  const unsigned long long tr0 = TR_NOW();
  unsigned long long tr_sx = 0, tr_sw = 0;
  (void)tr0; (void)tr_sx; (void)tr_sw;
  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (tile_live) {
    const bf16_t* xb = a.x + (long long)b * a.xbs;
    const int nchunk = a.CinP / CK;
    const int nit = nchunk * ntaps;
    const int lin_hi = Lin > 0 ? Lin - 1 : 0;
    const int cin_real = a.Cin > 0 ? a.Cin : a.CinP;

#include "kk_conv_mfma_stage.h"  // xreg / preg / xok, load_x, store_p, store_x (shared with variant 4)
    // four named registers instead of an array: hipcc kept a `uint4 wreg[4]` captured by the lambdas in scratch memory
    // (global load -> wait -> scratch store), which turned the prefetch into a synchronous copy
    uint4 w0, w1, w2, w3;
    auto load_w = [&](int it) __attribute__((always_inline)) {
      const int chunk = it / ntaps, tap = it - chunk * ntaps;
      const bf16_t* wt = a.w + ((long long)(widx0 + tap * wstep) * a.CoutP + n0) * a.CinP + chunk * CK + (long long)(tid >> 3) * a.CinP + (tid & 7) * 8;
      const long long step = (long long)32 * a.CinP;  // 256 threads cover 32 rows of 8 chunks
      w0 = *(const uint4*)(wt);
      w1 = *(const uint4*)(wt + step);
      w2 = *(const uint4*)(wt + 2 * step);
      w3 = *(const uint4*)(wt + 3 * step);
      asm volatile("" ::: "memory");
    };
    auto store_w = [&](bf16_t* Ws) __attribute__((always_inline)) {
      bf16_t* d = Ws + (tid >> 3) * XLD + (tid & 7) * 8;
      *(uint4*)(d) = w0;
      *(uint4*)(d + 32 * XLD) = w1;
      *(uint4*)(d + 64 * XLD) = w2;
      *(uint4*)(d + 96 * XLD) = w3;
    };

    // ---- prologue
    load_x(0);
    load_w(0);
    store_p(0);
    store_w(Ws0);
    __syncthreads();
    store_x(0);
    __syncthreads();
    if (nit > 1) load_w(1);
    const unsigned long long tr1 = TR_NOW();
    (void)tr1;
    TR_ADD(0, tr1 - tr0);  // prologue: first X slab + first W tile on chip

    const int arow = wr * WM + (lane & 31);  // + mi*32 + tap shift
    const int brow = wc * 64 + (lane & 31);  // + ni*32
    const int kofs = 8 * (lane >> 5);

    for (int it = 0; it < nit; ++it) {
      const int chunk = it / ntaps, tap = it - chunk * ntaps;
      const bf16_t* Ws = (it & 1) ? Ws1 : Ws0;
      const bool last_tap = tap == ntaps - 1;
      // prefetch the next channel slab of X at the FIRST tap of this slab: it has ntaps iterations to land
      if (tap == 0 && chunk + 1 < nchunk) load_x(chunk + 1);

      const int shift = (off0 + tap * dstep) - min_off;  // row shift of this tap inside the X slab
      const bf16_t* xa = Xs + (arow + shift) * XLD + kofs;
      const bf16_t* wb = Ws + brow * XLD + kofs;
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        const bf16x8 b0 = *(const bf16x8*)(wb + ks * 16);
        const bf16x8 b1 = *(const bf16x8*)(wb + 32 * XLD + ks * 16);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const bf16x8 av = *(const bf16x8*)(xa + mi * 32 * XLD + ks * 16);
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b0, acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1, acc[mi][1], 0, 0, 0);
        }
      }
#ifndef KK_MFMA_NO_SGB
      // prescribe the issue order: the fragments of k-step ks+1 are read from LDS WHILE the MFMAs of k-step ks run.  Left to
      // itself hipcc issues all 4 k-steps' ds_reads first and the 24 MFMAs after them, and the waves of a CU then fall into
      // lock step (everyone reads LDS, then everyone computes): measured 2860 cycles per iteration = LDS time + MFMA time.
      {
        constexpr int NF = 2 + MI, NM = 2 * MI;
        __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
#pragma unroll
        for (int ks = 0; ks < CK / 16 - 1; ++ks) {
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      }
#endif
      if (it + 1 < nit) {
        const unsigned long long ta = TR_NOW();
        store_w((it & 1) ? Ws0 : Ws1);  // buffer last read in iteration it-1; every wave has passed that barrier
        if (tap == 0 && chunk + 1 < nchunk) store_p(chunk + 1);  // parameter loads were issued with load_x above
        const unsigned long long tb = TR_NOW();
        tr_sw += tb - ta;
        if (last_tap) {
          __syncthreads();  // all waves are done with the X slab
          store_x(chunk + 1);
          tr_sx += TR_NOW() - tb;
        }
        __syncthreads();
        if (it + 2 < nit) load_w(it + 2);
      }
    }
    __syncthreads();  // main-loop LDS is dead; the epilogue tile aliases it
    TR_ADD(1, TR_NOW() - tr1);  // main loop
    TR_ADD(2, tr_sw);           //   of which: waiting for + storing the prefetched W tile
    TR_ADD(3, tr_sx);           //   of which: barrier + transform + store of the next X slab
  }

  const int bx = blockIdx.x;  // (row-tile index; variant 4 re-deals it per XCD)
#include "kk_conv_mfma_epilogue.h"  // (shared with variant 4)
}

template <typename TO, int WM, int NRM>
int launch_one(const KKMfmaArgs& a, int B, hipStream_t st) {
  using G = Geo<2 * WM>;
  static KKDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)conv_mfma_kernel<TO, WM, NRM>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_once.done();
  }
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  dim3 grid(kk_cdiv(a.Q, 2 * WM), a.CoutP / BN, B * nphase);
  hipLaunchKernelGGL((conv_mfma_kernel<TO, WM, NRM>), grid, dim3(256), G::LDS_BYTES, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}

}  // namespace

bool kk_mfma_eligible(int Cin, int Cout, int Kw, int mode, int stride, int dil) {
  if (Cout % 8 != 0 || Cout < 16 || Cin < 16) return false;  // callers may round a small Cout up to 8 (zero weights)
  if (mode == KK_CONV && stride != 1) return false;
  const int ntaps = mode == KK_CONV ? Kw : kk_cdiv(Kw, stride);
  const int halo = mode == KK_CONV ? (Kw - 1) * dil : (ntaps - 1);
  return halo <= MAX_HALO;
}

// rows of the output tile a launch with Q rows per phase uses (128 or 192); also the statistics tile size
int kk_mfma_tile_rows(int Q) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("KK_MFMA_BM");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 128 || forced == 192) return forced;
  (void)Q;
  // measured (tools/bench_conv.py): 192 rows (3 MFMA row blocks per wave, 250 VGPRs, 73 KB LDS -> still two workgroups per CU)
  // streams a third less W per MFMA than 128 and is 8-10 % faster on every shape; a 256-row variant spilled and was slower
  return 192;
}

int kk_mfma_stat_tile_rows(const KKMfmaArgs& a, int out_dtype) {
  (void)out_dtype;
  return kk_mfma_tile_rows(a.Q);
}

int kk_launch_conv_mfma(const KKMfmaArgs& a, int B, int out_dtype, hipStream_t st) {
  if (a.Q <= 0 || B <= 0) return 0;
  if (a.CinP % CK != 0 || a.CoutP % BN != 0) return kk_fail("conv_mfma: CinP must be a multiple of 64 and CoutP of 128");
  if (a.ldx % 8 != 0 || a.ldo % 8 != 0 || (a.res && a.ldr % 8 != 0)) return kk_fail("conv_mfma: row pitches must be multiples of 8 elements");
  if (((uintptr_t)a.x & 15) || ((uintptr_t)a.out & 15) || ((uintptr_t)a.res & 15) || ((uintptr_t)a.w & 15))
    return kk_fail("conv_mfma: pointers must be 16-byte aligned");
  if (a.ldx < a.CinP) return kk_fail("conv_mfma: input pitch smaller than the padded channel count");
  if (a.nrm_a && (a.nrm_stride % 4 != 0 || a.nrm_stride < a.CinP)) return kk_fail("conv_mfma: bad AdaIN parameter pitch");
  const int rows = kk_mfma_tile_rows(a.Q);
  const int nrm = a.nrm_a == nullptr ? 0 : (a.nrm_act == KK_ACT_SNAKE ? 1 : 2);
  if (nrm == 1 && a.nrm_C % 4 != 0) return kk_fail("conv_mfma: the fused Snake input needs a channel count that is a multiple of 4");
  KKMfmaArgs g = a;
  if (nrm == 2 && a.nrm_act != KK_ACT_LRELU) g.nrm_slope = 1.0f;  // plain AdaIN: identity activation
  if (out_dtype != KK_BF16) {
    if (nrm) return kk_fail("conv_mfma: fused AdaIN input needs a bf16 output");
    return rows == 192 ? launch_one<float, 96, 0>(g, B, st) : launch_one<float, 64, 0>(g, B, st);
  }
  if (rows == 192) {
    if (nrm == 1) return launch_one<bf16_t, 96, 1>(g, B, st);
    if (nrm == 2) return launch_one<bf16_t, 96, 2>(g, B, st);
    return launch_one<bf16_t, 96, 0>(g, B, st);
  }
  if (nrm == 1) return launch_one<bf16_t, 64, 1>(g, B, st);
  if (nrm == 2) return launch_one<bf16_t, 64, 2>(g, B, st);
  return launch_one<bf16_t, 64, 0>(g, B, st);
}
