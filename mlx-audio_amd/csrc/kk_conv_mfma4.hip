// Variant 4 of the bf16 MFMA implicit-GEMM convolution (see kk_conv_mfma.hip for the base design): the weight (B) fragments
// do NOT go through LDS.  W is packed in FRAGMENT ORDER at load time (kk_mfma4_pack_index), so a wave fetches the 1 KiB
// fragment of its 32 output channels x 16 k for one k-step as ONE fully coalesced global_load_dwordx4, straight into the MFMA
// operand registers, one (tap, slab) iteration ahead.  That removes the W double buffer (LDS 73 -> 48 KB), the ds_write of W,
// 8 of the 20 ds_read_b128 per wave and iteration, and the barrier per tap: waves synchronise only when the X slab changes.
//
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
// Round 3: the matrix instruction is v_mfma_f32_16x16x32_bf16 (KK_MFMA16, the default; -DKK_MFMA32 builds the round-2 32x32x16 form for A/B).
// Same FLOPs per cycle and the same operand bytes per FLOP from LDS / global memory, but the chip holds a higher clock on it under load
// (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15 x in MFMA-paced loops); measured here with a timing-only build first
// (build.py --exp16): -4 ... -10 % on the MFMA-heavy layers.  What changes with the shape:
//   * A fragment of lane l: row l % 16, k-group l / 16 (4 groups of 8 k = 32 k per step, two steps per 64-channel slab).  With the 144-byte
//     row pitch the four lane groups of a ds_read_b128 collide 2-way; a 160-BYTE PITCH puts 16 consecutive rows on the even 16-byte granules
//     and the odd k-groups on the odd ones: conflict-free again (slab 38.7 KB instead of 34.8).
//   * B fragment of lane l: column l % 16, k-group l / 16; a wave's 64 columns are 4 fragments per k-step: pack order
//     [tap][n block][chunk][wc][ks][ni][lane] x 16 bytes, still ONE coalesced 1 KiB load per fragment.
//   * accumulators: 16 x 16 blocks, lane l holds rows 4 * (l / 16) .. + 3 of column l % 16: (WM / 16) x 4 blocks of 4 registers = the same 96.
#include <stdlib.h>

#include "kk_common.h"
#include "kk_kernels.h"
#include "kk_conv_mfma_shared.h"

#define TR_NOW() 0ull
#define TR_ADD(slot, v) do { } while (0)
namespace {
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128, CK = 64;
constexpr int XLD = KK_XLD;   // elements per LDS row (160 B with the 16x16x32 fragments, 144 B with 32x32x16: kk_conv_mfma_shared.h)
constexpr int MAX_HALO = 50;  // (Kw-1)*dil of the largest resblock conv (k 11, dilation 5)
constexpr int CLD = BN;       // fp32 epilogue tile pitch: 128 x 128 x 4 B = exactly 64 KiB

constexpr int PS_BYTES = 2 * 3 * CK * 4;  // double-buffered AdaIN parameter table of one slab (A, B, alpha)

template <int BM>
struct Geo {
  static constexpr int XROWS = BM + MAX_HALO;
  static constexpr int XS_BYTES = XROWS * XLD * 2;
  static constexpr int MAIN_BYTES = XS_BYTES + PS_BYTES;
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
// WM = 96: 192-row tile, 2 workgroups per CU (230-256 VGPRs).  WM = 64: 128-row tile, 3 workgroups per CU (<= 168 VGPRs, 32 KB LDS): more
// W traffic per MFMA, but a third workgroup to cover the serial prologue / slab staging / epilogue phases of the other two.
template <typename TO, int WM, int NRM>
__global__ __launch_bounds__(256, (WM == 96 ? 2 : 3)) __attribute__((amdgpu_waves_per_eu((WM == 96 ? 2 : 3), (WM == 96 ? 2 : 3)))) void conv_mfma4_kernel(KKMfmaArgs a) {
  constexpr int BM = 2 * WM, MI = WM / 32;
  (void)MI;
  using G = Geo<BM>;
  constexpr int XREG = G::XREG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xs = (bf16_t*)smem;
  float* Ps = (float*)(smem + G::XS_BYTES);  // [2][3][64]
  float* Cs = (float*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  // XCD-aware tile order: the dispatcher deals workgroup ids (x fastest, then y, z) round-robin to the 8 XCDs, each with its own L2.
  // Neighbouring row tiles share their halo rows (up to 50 of 192 + 50 for k = 11, dilation 5) and the column blocks of one row tile share
  // the whole X slab, so the flat id is re-dealt: an XCD walks a contiguous run of (utterance, row tile) pairs, column blocks innermost.
  int bx, by, bz;
  {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    int lid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int per = total / 8, rem = total - per * 8;
    const int xcd = lid & 7, idx = lid >> 3;
    if (!(a.dbg & 4)) lid = xcd * per + (xcd < rem ? xcd : rem) + idx;  // dbg bit 2 (KK_MFMA_NOXCD=1): the dispatcher's own order, for A/B timing
    else { by = blockIdx.y; bx = blockIdx.x; bz = blockIdx.z; }
    if (!(a.dbg & 4)) {
      by = lid % gy;  // column blocks innermost
      const int t = lid / gy;
      bx = t % gx;
      bz = t / gx;
    }
  }
  const int b = bz / nphase, phase = bz - b * nphase;
  const int q0 = bx * BM, n0 = by * BN;
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
#ifdef KK_MFMA16
  constexpr int MI16 = WM / 16;
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

  if (tile_live) {
    const bf16_t* xb = a.x + (long long)b * a.xbs;
    const int nchunk = a.CinP / CK;
    const int nit = nchunk * ntaps;
    const int lin_hi = Lin > 0 ? Lin - 1 : 0;
    const int cin_real = a.Cin > 0 ? a.Cin : a.CinP;

#include "kk_conv_mfma_stage.h"  // xreg / preg / xok, load_x, store_p, store_x (shared with variant 2)
    // B fragments of one (tap, slab) iteration: [ni][ks], loaded from the fragment-order pack one iteration ahead.  Named
    // scalars, not an array (hipcc put a lambda-captured register array in scratch once already).
    uint4 q00, q01, q02, q03, q10, q11, q12, q13;
    const int nb = by;
    auto frag_ptr = [&](int it) __attribute__((always_inline)) -> const uint4* {
      const int chunk = it / ntaps, tap = it - chunk * ntaps;
      // pack order: [tap][n block][chunk][wc][ks][ni][lane] x 16 bytes (KK_MFMA16; [wc][ni][ks] with 32x32x16)
      const long long blk = ((long long)(widx0 + tap * wstep) * (a.CoutP / BN) + nb) * nchunk + chunk;
      return (const uint4*)a.wf + blk * 1024 + (wc * 2) * 4 * 64 + lane;
    };
    // ---- prologue
    load_x(0);
    {
      const uint4* fp = frag_ptr(0);
      q00 = fp[0 * 64]; q01 = fp[1 * 64]; q02 = fp[2 * 64]; q03 = fp[3 * 64];
      q10 = fp[4 * 64]; q11 = fp[5 * 64]; q12 = fp[6 * 64]; q13 = fp[7 * 64];
      asm volatile("" ::: "memory");
    }
    store_p(0);
    __syncthreads();
    store_x(0);
    __syncthreads();

#ifdef KK_MFMA16
    const int arow = wr * WM + (lane & 15);  // + mi*16 + tap shift
    const int kofs = 8 * (lane >> 4);
#else
    const int arow = wr * WM + (lane & 31);  // + mi*32 + tap shift
    const int kofs = 8 * (lane >> 5);
#endif

    for (int it = 0; it < nit; ++it) {
      const int chunk = it / ntaps, tap = it - chunk * ntaps;
      const bool last_tap = tap == ntaps - 1;
      // prefetch the next channel slab of X at the FIRST tap of this slab: it has ntaps iterations to land.  (Measured in round 2: hipcc loads the
      // slab into rotated registers under this condition and moves them into place behind vmcnt waits, i.e. it waits for the rows here.  A slab loop
      // with the request outside any condition removes those waits and is 2.5 % SLOWER on the k = 11 layers, +-0 elsewhere: vmcnt retires in order, so
      // the weight fragments requested after the rows wait for them one tap later anyway, and the second workgroup of the CU covers either stall.)
      if (tap == 0 && chunk + 1 < nchunk) load_x(chunk + 1);
      // next iteration's fragments (the last iteration re-requests its own: no branch around the loads)
      const uint4* fn = frag_ptr(it + 1 < nit ? it + 1 : it);

      const int shift = (off0 + tap * dstep) - min_off;  // row shift of this tap inside the X slab
      const bf16_t* xa = Xs + (arow + shift) * XLD + kofs;
#ifdef KK_MFMA16
      // one k-step = 32 channels: 4 B fragments (this wave's 4 x 16 columns) stay in registers, the WM / 16 A fragments stream through
#define KK_KSTEP(KS, B0, B1, B2, B3)                                                                                \
      {                                                                                                              \
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, B0), b1 = __builtin_bit_cast(bf16x8, B1);                     \
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, B2), b3 = __builtin_bit_cast(bf16x8, B3);                     \
        _Pragma("unroll") for (int mi = 0; mi < MI16; ++mi) {                                                        \
          const bf16x8 av = *(const bf16x8*)(xa + mi * 16 * XLD + (KS) * 32);                                       \
          acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b0, acc[mi][0], 0, 0, 0);                         \
          acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b1, acc[mi][1], 0, 0, 0);                         \
          acc[mi][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b2, acc[mi][2], 0, 0, 0);                         \
          acc[mi][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b3, acc[mi][3], 0, 0, 0);                         \
        }                                                                                                            \
        B0 = fn[((KS) * 4 + 0) * 64]; /* the registers are free as soon as these MFMAs are issued */               \
        B1 = fn[((KS) * 4 + 1) * 64];                                                                               \
        B2 = fn[((KS) * 4 + 2) * 64];                                                                               \
        B3 = fn[((KS) * 4 + 3) * 64];                                                                               \
      }
      KK_KSTEP(0, q00, q01, q02, q03)
      KK_KSTEP(1, q10, q11, q12, q13)
#undef KK_KSTEP
      // issue order: the A fragment of row block mi + 1 is read from LDS while the 4 MFMAs of row block mi run; the four global loads that
      // refill a k-step's B registers go out right behind that k-step's MFMAs
      {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int j = 0; j < MI16; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (ks * MI16 + j + 2 < 2 * MI16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        }
      }
#else
#define KK_KSTEP(KS, B0, B1)                                                                                        \
      {                                                                                                              \
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, B0), b1 = __builtin_bit_cast(bf16x8, B1);                     \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                          \
          const bf16x8 av = *(const bf16x8*)(xa + mi * 32 * XLD + (KS) * 16);                                       \
          acc[mi][0] = kk_mfma32<(KS)>(av, b0, acc[mi][0]);                         \
          acc[mi][1] = kk_mfma32<(KS)>(av, b1, acc[mi][1]);                         \
        }                                                                                                            \
        B0 = fn[(KS) * 64];       /* the registers are free as soon as these MFMAs are issued */                   \
        B1 = fn[(4 + (KS)) * 64];                                                                                   \
      }
      KK_KSTEP(0, q00, q10)
      KK_KSTEP(1, q01, q11)
      KK_KSTEP(2, q02, q12)
      KK_KSTEP(3, q03, q13)
#undef KK_KSTEP
      // issue order: the A fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run, and the two global
      // loads that refill a k-step's B registers go out right behind that k-step's MFMAs (hipcc otherwise parks all eight at the
      // end of the iteration, one barrier away from their first use)
      {
        __builtin_amdgcn_sched_group_barrier(0x100, MI, 0);
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
#pragma unroll
          for (int j = 0; j < MI; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, KK_MFMA_PER, 0);
            if (ks < CK / 16 - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, MI * KK_MFMA_PER, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
      }
#endif
      asm volatile("" ::: "memory");
      if (it + 1 < nit) {
        if (tap == 0 && chunk + 1 < nchunk) store_p(chunk + 1);  // parameter loads were issued with load_x above
        if (last_tap) {
          __syncthreads();  // all waves are done with the X slab
          store_x(chunk + 1);
          __syncthreads();
        }
      }
    }
    __syncthreads();  // main-loop LDS is dead; the epilogue tile aliases it
  }

#ifdef KK_MFMA16
#define KK_EPI_ACC16 1
#endif
#include "kk_conv_mfma_epilogue.h"  // (shared with variant 2)
#undef KK_EPI_ACC16
}

template <typename TO, int WM, int NRM>
int launch_one(const KKMfmaArgs& a, int B, hipStream_t st) {
  using G = Geo<2 * WM>;
  static KKDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)conv_mfma4_kernel<TO, WM, NRM>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_once.done();
  }
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  dim3 grid(kk_cdiv(a.Q, 2 * WM), a.CoutP / BN, B * nphase);
  static int noxcd = -1;
  if (noxcd < 0) noxcd = getenv("KK_MFMA_NOXCD") ? 1 : 0;
  KKMfmaArgs a2 = a;
  if (noxcd) a2.dbg |= 4;
  hipLaunchKernelGGL((conv_mfma4_kernel<TO, WM, NRM>), grid, dim3(256), G::LDS_BYTES, st, a2);
  KK_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// element index of W[tap][cout][k] in the fragment-order pack ([tap][n block][chunk][wc][ni][ks][lane][8])
long long kk_mfma4_pack_index(int tap, int cout, int k, int CoutP, int CinP) {
  const int nbk = cout / BN, cr = cout % BN, chunk = k / CK, kr = k % CK;
  const long long blk = ((long long)tap * (CoutP / BN) + nbk) * (CinP / CK) + chunk;
#ifdef KK_MFMA16
  // [wc][ks (2 x 32 k)][ni (4 x 16 columns)][lane = k-group * 16 + column][8]
  const int wc = cr / 64, ni = (cr % 64) / 16, ks = kr / 32, kq = (kr % 32) / 8, j = kr % 8;
  const int lane = kq * 16 + (cr % 16);
  return blk * (BN * CK) + ((long long)(((wc * 2 + ks) * 4 + ni) * 64 + lane)) * 8 + j;
#else
  const int wc = cr / 64, ni = (cr % 64) / 32, ks = kr / 16, hh = (kr % 16) / 8, j = kr % 8;
  const int lane = hh * 32 + (cr % 32);
  return blk * (BN * CK) + ((long long)(((wc * 2 + ni) * 4 + ks) * 64 + lane)) * 8 + j;
#endif
}

namespace {
__global__ __launch_bounds__(256) void pack_w_frag_kernel(const bf16_t* w, bf16_t* wf, int Kw, int CoutP, int CinP) {
  const long long n = (long long)Kw * CoutP * CinP;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int k = (int)(e % CinP);
    const long long r = e / CinP;
    const int cout = (int)(r % CoutP), tap = (int)(r / CoutP);
    const int nbk = cout / BN, cr = cout % BN, chunk = k / CK, kr = k % CK;
    const long long blk = ((long long)tap * (CoutP / BN) + nbk) * (CinP / CK) + chunk;
#ifdef KK_MFMA16
    const int wc = cr / 64, ni = (cr % 64) / 16, ks = kr / 32, kq = (kr % 32) / 8, j = kr % 8;
    const int lane = kq * 16 + (cr % 16);
    wf[blk * (BN * CK) + ((long long)(((wc * 2 + ks) * 4 + ni) * 64 + lane)) * 8 + j] = w[e];
#else
    const int wc = cr / 64, ni = (cr % 64) / 32, ks = kr / 16, hh = (kr % 16) / 8, j = kr % 8;
    const int lane = hh * 32 + (cr % 32);
    wf[blk * (BN * CK) + ((long long)(((wc * 2 + ni) * 4 + ks) * 64 + lane)) * 8 + j] = w[e];
#endif
  }
}
}  // namespace

// device-side re-layout [Kw][CoutP][CinP] -> fragment order (the single-kernel test entry points; the model packs on the host)
int kk_launch_pack_w_frag(const void* w, void* wf, int Kw, int CoutP, int CinP, hipStream_t st) {
  if (CinP % CK != 0 || CoutP % BN != 0) return kk_fail("pack_w_frag: CinP must be a multiple of 64 and CoutP of 128");
  hipLaunchKernelGGL(pack_w_frag_kernel, dim3(1024), dim3(256), 0, st, (const bf16_t*)w, (bf16_t*)wf, Kw, CoutP, CinP);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_conv_mfma4(const KKMfmaArgs& a, int B, int out_dtype, hipStream_t st) {
  if (a.Q <= 0 || B <= 0) return 0;
  if (!a.wf) return kk_fail("conv_mfma4: fragment-order weights missing");
  const int nrm = a.nrm_a == nullptr ? 0 : (a.nrm_act == KK_ACT_SNAKE ? 1 : 2);
  if (nrm == 1 && a.nrm_C % 4 != 0) return kk_fail("conv_mfma: the fused Snake input needs a channel count that is a multiple of 4");
  KKMfmaArgs g = a;
  if (nrm == 2 && a.nrm_act != KK_ACT_LRELU) g.nrm_slope = 1.0f;  // plain AdaIN: identity activation
  if (out_dtype != KK_BF16) return kk_fail("conv_mfma4: bf16 output only");
  if (kk_mfma_tile_rows(a.Q) == 128) {
    if (nrm == 1) return launch_one<bf16_t, 64, 1>(g, B, st);
    if (nrm == 2) return launch_one<bf16_t, 64, 2>(g, B, st);
    return launch_one<bf16_t, 64, 0>(g, B, st);
  }
  if (nrm == 1) return launch_one<bf16_t, 96, 1>(g, B, st);
  if (nrm == 2) return launch_one<bf16_t, 96, 2>(g, B, st);
  return launch_one<bf16_t, 96, 0>(g, B, st);
}
