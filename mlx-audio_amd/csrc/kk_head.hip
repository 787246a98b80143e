// Fused vocoder head (bf16 mode): LeakyReLU(0.01) -> conv_post (128 -> 22 channels, k = 7, pad 3; istftnet.py:798-803) -> spec = exp,
// phase = sin (:804-805) -> MLXSTFT.inverse (:497-523) / istft (mlx_audio/utils.py:104-158): inverse real DFT of 11 bins, periodic Hann,
// overlap-add at hop 5, division by the window sum, trim.
//
// The stand-alone pair -- conv_post on the 128-column MFMA tile (106 of its 128 output columns are padding) + the iSTFT head kernel --
// moves the 22-channel tensor through HBM once in each direction (44 of the head's 64 algorithmic bytes per frame column) and spends a
// full 128 -> 128 k = 7 convolution's time on a 128 -> 22 one.  Here one workgroup owns 256 consecutive frames:
//   * the [262][128] bf16 input slab is staged ONCE into LDS (all loads in flight together, LeakyReLU applied on the way in),
//   * v_mfma_f32_32x32x16_bf16 with K = 7 taps x 128 channels = 56 k-steps SPLIT OVER THE 4 WAVES: a wave holds the weight fragments of its
//     14 k-steps in registers (56 VGPRs, fetched once per tile in MFMA fragment order: 57 KB per workgroup instead of 57 KB per wave, and no
//     load inside the loop -- the first version, 64 rows per wave with an 8-deep ring of in-loop weight loads, waited on L2 latency every
//     k-step: 307 us) and multiplies all 8 row blocks of the tile with them (112 MFMAs, 128 accumulator registers),
//   * the four partial sums are combined through LDS in a fixed order ((w0 + w2) + (w1 + w3)) into a frame-major tile, so that thread t holds
//     the 22 values of frame t (+ bias) and computes its 20 windowed samples (kk_istft_math.h: the arithmetic of
//     istft_head_wave_fast_kernel, on the UNROUNDED fp32 conv output),
//   * overlap-add in ascending frame order (the reference's scatter-add order): in-wave shuffles, the three frames a wave needs from its
//     predecessor through LDS; 3 halo frames per tile are recomputed (253 hop blocks per 256 frames),
// so the kernel reads 256 B and writes 20 B per frame column and nothing else: HBM-bound (0.69 GB per B = 32 batch).
#include <stdlib.h>

#include "kk_common.h"
#include "kk_kernels.h"
#include "kk_istft_math.h"

namespace {
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD_C = 128;            // input channels (upsample_initial_channel >> n_upsamples)
constexpr int HD_K = 7;              // conv_post taps
constexpr int HD_NOUT = 22;          // n_fft + 2
constexpr int HD_XLD = HD_C + 8;     // LDS row pitch in elements (272 B: conflict-free ds_read_b128 fragments)
constexpr int HD_NKS = HD_C / 16;    // k-steps per tap
constexpr int HD_NIT = HD_K * HD_NKS;
constexpr int HD_YLD = 23;           // fp32 pitch of the frame-major exchange tile (odd: conflict-free rows)
constexpr int HD_EX_BYTES = 4 * 3 * 15 * 4;  // y[5..19] of the last three frames of every wave
template <int ROWS>  // frames per workgroup (256: one frame per thread; 128: half the threads idle in the frame phase, but 4 workgroups per CU)
struct HeadGeo {
  static constexpr int HB = ROWS - 3;  // hop blocks produced per workgroup
  static constexpr int XROWS = ROWS + HD_K - 1;
  static constexpr int XREG = (XROWS * (HD_C / 8) + 255) / 256;  // 16-byte chunks of the slab per thread
  static constexpr int XS_BYTES = XROWS * HD_XLD * 2;
  static constexpr int YS_BYTES = 2 * ROWS * HD_YLD * 4;  // two partial-sum tiles
  static constexpr int MAINB = XS_BYTES > YS_BYTES ? XS_BYTES : YS_BYTES;
  static constexpr int LDS = MAINB + HD_EX_BYTES;
};
constexpr int HD_WIT = HD_NIT / 4;   // k-steps (weight fragments) per wave

template <int HD_ROWS>
__global__ __launch_bounds__(256, (HD_ROWS == 256 ? 2 : 3)) void conv_post_istft_kernel(KKHeadArgs a) {
  using G = HeadGeo<HD_ROWS>;
  constexpr int HD_HB = G::HB, HD_XROWS = G::XROWS, HD_XREG = G::XREG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xs = (bf16_t*)smem;
  float* Ys = (float*)smem;  // aliases the slab after the main loop
  float* Ex = (float*)(smem + G::MAINB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
  const int Tf = a.len_frames ? a.len_frames[b] : a.Tfmax;
  const int f0 = blockIdx.x * HD_HB;  // first hop block of this tile; frame of row t: f0 - 3 + t
  const int nout = Tf > 0 ? 5 * (Tf - 1) : 0;
  const int ntot = 5 * (a.Tfmax - 1);
  float* wb = a.wav + (long long)b * a.wbs;
  if (f0 - 3 >= Tf) {  // (uniform over the workgroup)
    // nothing of this utterance reaches the tile: its samples are zeros
    const int n0 = 5 * f0 - 10;
    for (int e = tid; e < 5 * HD_HB; e += 256) {
      const int n = n0 + e;
      if (n >= 0 && n < ntot) wb[n] = 0.f;
    }
    return;
  }
  // this wave's 14 weight fragments: requested first, they land while the slab is being staged.  (A persistent form -- 512 workgroups
  // walking the tiles, the fragments loaded once -- was 8 % faster before the main loop was pipelined but pushes the kernel to 256 VGPRs
  // + spills, where hipcc serialises the LDS reads again.)
  uint4 wreg[HD_WIT];
  {
    const uint4* wf = (const uint4*)a.wf + (long long)wave * HD_WIT * 64 + lane;  // fragment (tap, ks) at (tap * NKS + ks) * 64 uint4
#pragma unroll
    for (int j = 0; j < HD_WIT; ++j) wreg[j] = wf[j * 64];
  }
  // ---- stage the input slab: rows f0 - 6 .. f0 - 6 + 261 of x, LeakyReLU(in_slope), zeros outside [0, Tf)
  {
    const bf16_t* xb = a.x + (long long)b * a.xbs;
    const int hi = a.Tfmax - 1;
    uint4 xr[HD_XREG];
#pragma unroll
    for (int i = 0; i < HD_XREG; ++i) {
      const int id = i * 256 + tid;
      const int r = id >> 4, c8 = (id & 15) * 8;
      int row = f0 - 6 + r;
      row = row < 0 ? 0 : (row > hi ? hi : row);
      if (a.dbg & 4) row &= 63;  // timing experiment: the slab comes from the first 64 rows (cache resident)
      xr[i] = *(const uint4*)(xb + (long long)row * a.ldx + c8);
    }
    asm volatile("" ::: "memory");
    const float slope = a.in_slope;
#pragma unroll
    for (int i = 0; i < HD_XREG; ++i) {
      const int id = i * 256 + tid;
      const int r = id >> 4, c8 = (id & 15) * 8;
      const int row = f0 - 6 + r;
      if (r < HD_XROWS) {
        const unsigned msk = (row >= 0 && row < Tf) ? 0xFFFFFFFFu : 0u;
        unsigned w4[4] = {xr[i].x & msk, xr[i].y & msk, xr[i].z & msk, xr[i].w & msk};
        if (slope != 1.0f)  // (uniform; 1 = the producer's epilogue applied the LeakyReLU already: the slab is a masked copy)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // LeakyReLU with 0 < slope < 1 is max(x, slope * x): one packed multiply + one packed max per PAIR (the staging transform is
          // a third of this kernel's vector instructions)
          const kk_istft::v2f xv = {__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xFFFF0000u)};
          const kk_istft::v2f yv = __builtin_elementwise_max(xv, xv * kk_istft::v2f{slope, slope});
          const bf16x2 pk = {(bf16_t)yv.x, (bf16_t)yv.y};
          w4[k] = __builtin_bit_cast(unsigned, pk);
        }
        *(uint4*)(Xs + r * HD_XLD + c8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
      }
    }
  }
  // ---- main loop: out[row][n] = sum_tap sum_c W[tap][n][c] * X[row + tap][c].  The 56 k-steps (7 taps x 8) are SPLIT OVER THE 4 WAVES: wave w
  // keeps the weight fragments of its 14 k-steps in registers (loaded once, before the slab is staged: nothing is fetched inside the loop)
  // and walks all 8 row blocks of the tile with them; the four partial sums meet in LDS afterwards, in a fixed order.
  f32x16 acc[HD_ROWS / 32];
#pragma unroll
  for (int i = 0; i < HD_ROWS / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  __syncthreads();
  const bf16_t* xa0 = Xs + (lane & 31) * HD_XLD + 8 * (lane >> 5);
  {
    // 14 k-steps x NB row blocks = one flat sequence of MFMAs; the A fragment of step s + DEPTH is requested right behind the MFMA of step s
    // (left to itself hipcc, short of registers next to 128 accumulators, reads each fragment just before its MFMA and waits out the LDS
    // latency every time: 112 x ~150 cycles per tile).  The sched_group_barriers pin "one MFMA, one LDS read" in the emitted order.
    constexpr int NB = HD_ROWS / 32, NS = HD_WIT * NB, DEPTH = 6;
    auto a_addr = [&](int sidx) -> const bf16x8* {
      const int j = sidx / NB, mi = sidx - j * NB;
      const int it = wave * HD_WIT + j, tap = it / HD_NKS, ks = it - tap * HD_NKS;
      return (const bf16x8*)(xa0 + (tap + mi * 32) * HD_XLD + ks * 16);
    };
    bf16x8 ar[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ar[d] = *a_addr(d);
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int j = sidx / NB, mi = sidx - j * NB;
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[sidx % DEPTH], __builtin_bit_cast(bf16x8, wreg[j]), acc[mi], 0, 0, 0);
      if (sidx + DEPTH < NS) ar[sidx % DEPTH] = *a_addr(sidx + DEPTH);
    }
    // emitted order: DEPTH reads, then (MFMA, read) pairs, then the last DEPTH MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
#pragma unroll
    for (int sidx = 0; sidx < NS - DEPTH; ++sidx) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, DEPTH, 0);
  }
  __syncthreads();  // every wave is done with the slab: the exchange tiles alias it
  // ---- partial sums -> two frame-major fp32 tiles: A = (wave 0 + wave 2), B = (wave 1 + wave 3).  Both waves of a pair work in both steps:
  // step 1, each stores its partial of ONE half of the row blocks (wave w < 2 the lower half, its partner the upper half); step 2, each adds
  // its partial of the OTHER half onto what its partner stored -- per row block 16 independent reads, 16 adds, 16 writes (a plain
  // `*p += v` loop is 128 dependent LDS round trips: the compiler must assume the addresses alias).  A lane always touches the same addresses.
  {
    constexpr int NB = HD_ROWS / 32, HALF = NB / 2;
    const int col = lane & 31;
    float* Yw = Ys + (wave & 1) * (HD_ROWS * HD_YLD) + (4 * (lane >> 5)) * HD_YLD + col;
    const int mine0 = wave < 2 ? 0 : HALF;  // first row block this wave stores in step 1
    if (col < HD_NOUT) {
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // (both halves are addressed with compile-time register indices: acc[i] for the lower half, acc[HALF + i] for the upper)
          const float v = wave < 2 ? acc[i][r] : acc[HALF + i][r];
          Yw[((mine0 + i) * 32 + (r & 3) + 8 * (r >> 2)) * HD_YLD] = v;
        }
      }
    }
    __syncthreads();
    if (col < HD_NOUT) {
      const int other0 = wave < 2 ? HALF : 0;
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = Yw[((other0 + i) * 32 + (r & 3) + 8 * (r >> 2)) * HD_YLD];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // fixed order of the pair's sum: (lower wave's partial) + (upper wave's partial)
          const float v = wave < 2 ? acc[HALF + i][r] : acc[i][r];
          t[r] = wave < 2 ? v + t[r] : t[r] + v;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Yw[((other0 + i) * 32 + (r & 3) + 8 * (r >> 2)) * HD_YLD] = t[r];
      }
    }
  }
  __syncthreads();
  // ---- thread t = frame f0 - 3 + t
  const int f = f0 - 3 + tid;
  const bool act = tid < HD_ROWS;  // (ROWS = 128: waves 2, 3 only keep the barriers company from here on)
  const bool fv = act && f >= 0 && f < Tf;
  const int tq = act ? tid : 0;
  float in[22];
#pragma unroll
  for (int k = 0; k < 22; ++k) in[k] = (Ys[tq * HD_YLD + k] + Ys[HD_ROWS * HD_YLD + tq * HD_YLD + k]) + a.bias[k];
  if (a.cp_out && tid >= 3 && fv) {  // debug: materialise conv_post (bf16, pitch cp_ld) for kk_debug_fetch
    bf16_t* cr = a.cp_out + (long long)b * a.cp_bs + (long long)f * a.cp_ld;
#pragma unroll
    for (int k = 0; k < 22; ++k) cr[k] = (bf16_t)in[k];
  }
  float y[20];
  if (a.dbg & 2) {  // timing experiment: no frame arithmetic
#pragma unroll
    for (int o = 0; o < 20; ++o) y[o] = in[o];
  } else {
    kk_istft::frame_fast(in, fv ? 1.0f : 0.0f, y);
  }
  if (lane >= 61 && act) {
#pragma unroll
    for (int k = 0; k < 15; ++k) Ex[(wave * 3 + (lane - 61)) * 15 + k] = y[5 + k];
  }
  __syncthreads();
  const int g = f;
  const bool interior = g >= 3 && g < Tf;  // frames g-3 .. g all exist
  const bool mine = tid >= 3 && act;
  float v[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    float s3 = __shfl_up(y[15 + r], 3), s2 = __shfl_up(y[10 + r], 2), s1 = __shfl_up(y[5 + r], 1);
    if (wave > 0 && lane < 3) {  // predecessors live in the previous wave: its lanes 61 + (lane - j) -> exchange slot lane - j + 3
      const float* e = Ex + (wave - 1) * 3 * 15;
      s3 = e[lane * 15 + 10 + r];                         // frame t-3 = previous wave's lane 61 + lane, y[15 + r]
      if (lane < 2) s2 = e[(lane + 1) * 15 + 5 + r];      // frame t-2 = lane 62 + lane,            y[10 + r]
      if (lane < 1) s1 = e[(lane + 2) * 15 + r];          // frame t-1 = lane 63,                   y[5 + r]
    }
    v[r] = ((s3 + s2) + s1) + y[r];  // ascending frame order (utils.py:138-147); frames outside [0, Tf) hold exact zeros
  }
  if (__ballot(mine && !interior) != 0ull) {  // an utterance edge inside this wave: partial window sums (utils.py:143-150)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      float ws = 0.f;
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        const int fj = g - j;
        if (fj >= 0 && fj < Tf) ws += a.hann_per[5 * j + r];
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

}  // namespace

// element index of conv_post's W[tap][cout][cin] in the head kernel's fragment order: [tap][k-step][lane = 32 h + cout][8], cin = 16 ks + 8 h + j
long long kk_head_pack_index(int tap, int cout, int cin) {
  const int ks = cin / 16, h = (cin % 16) / 8, j = cin % 8;
  return ((long long)(tap * HD_NKS + ks) * 64 + h * 32 + cout) * 8 + j;
}
size_t kk_head_pack_elems() { return (size_t)HD_NIT * 64 * 8; }
bool kk_head_eligible(int Cin, int Cout, int Kw, int n_fft, int hop) { return Cin == HD_C && Cout == HD_NOUT && Kw == HD_K && n_fft == 20 && hop == 5; }

namespace {
__global__ __launch_bounds__(256) void pack_head_w_kernel(const bf16_t* w, bf16_t* wf) {
  const int n = HD_NIT * 64 * 8;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int j = e & 7, ln = (e >> 3) & 63, blk = e >> 9;  // [tap * NKS + ks][lane][8]
    const int tap = blk / HD_NKS, ks = blk % HD_NKS, cout = ln & 31, cin = ks * 16 + 8 * (ln >> 5) + j;
    wf[e] = cout < HD_NOUT ? w[((long long)tap * HD_NOUT + cout) * HD_C + cin] : (bf16_t)0.0f;
  }
}
}  // namespace

int kk_launch_pack_head_w(const bf16_t* w, bf16_t* wf, hipStream_t st) {
  hipLaunchKernelGGL(pack_head_w_kernel, dim3(32), dim3(256), 0, st, w, wf);
  KK_CHECK_LAUNCH();
  return 0;
}

int kk_launch_conv_post_istft(const KKHeadArgs& a, int B, hipStream_t st) {
  if (B <= 0 || a.Tfmax <= 0) return 0;
  if (a.ldx < HD_C || (a.ldx & 7) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.wf & 15)) return kk_fail("conv_post_istft: input pitch / alignment");
  if (5LL * a.Tfmax >= 0x7fffffffLL) return kk_fail("conv_post_istft: utterance too long");
  static KKDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute((const void*)conv_post_istft_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, HeadGeo<256>::LDS);
    (void)hipFuncSetAttribute((const void*)conv_post_istft_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, HeadGeo<128>::LDS);
    attr_once.done();
  }
  static int rows = -1;
  if (rows < 0) {
    const char* e = getenv("KK_HEAD_ROWS");  // A/B switch
    rows = e ? atoi(e) : 256;
  }
  if (rows == 128) {
    dim3 grid(kk_cdiv(a.Tfmax + 3, HeadGeo<128>::HB), B);
    hipLaunchKernelGGL(conv_post_istft_kernel<128>, grid, dim3(256), HeadGeo<128>::LDS, st, a);
  } else {
    dim3 grid(kk_cdiv(a.Tfmax + 3, HeadGeo<256>::HB), B);
    hipLaunchKernelGGL(conv_post_istft_kernel<256>, grid, dim3(256), HeadGeo<256>::LDS, st, a);
  }
  KK_CHECK_LAUNCH();
  return 0;
}
