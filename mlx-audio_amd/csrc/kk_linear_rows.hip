// Linear layers over the token axis (k = 1, stride 1, no fused transform: Albert's q|k|v / dense / ffn / ffn_output, `map_in`, `bert_encoder`, the LSTM input
// projections) as a STREAMING matrix-core kernel without LDS and without barriers (round 3, for the B = 1 latency).
// The tiled convolution kernel runs these at a grid of 6-18 workgroups at B = 1 and walks the 12-32 K slabs of a tile one exposed load latency at a time (one
// slab of X prefetch, two barriers per slab): 37 us per launch at B = 1 and 46 us at B = 32 -- it does not scale with the work.  Here a WAVE owns 16 rows x 64
// (or 32) columns: its A fragment (16 rows x 32 k) is ONE 16-byte global load per lane straight from the row-major activation (lane L: row L % 16, k octet L / 16), its four
// B fragments come from a fragment-order pack of the weight ([N / 16][K / 32][64 lanes][8 bf16]: 1 KiB per wave load), and a register ring keeps PD K chunks in
// flight.  The rows of a dense [B][T][C] tensor are ONE flat row axis (a tile may span utterances; a row past its utterance's length is stored as zeros).
// Bias, exact-erf GELU (Albert's ffn), bf16 stores as packed pairs.  A row's result depends on nothing but its own input row: batch-invariant.
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

typedef float lr_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lr_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned lr_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lr_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// NS 16-column sub-blocks and NR 16-row tiles per wave (the wave owns 16 NR rows x 16 NS columns), PD K chunks in flight per wave ((NR + NS) 16-byte registers
// each).  <2, 1, 12>: two dependent load rounds for a K = 768 product -- the rows of one or two utterances; <4, 1, 6>: up to ~1000 rows (beyond that the
// tiled convolution kernel is used).  The tiling never changes an output element's K order.
template <int NS, int NR, int PD>
__global__ __launch_bounds__(256) void linear_rows_mfma_kernel(KKLinMfmaArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int item = blockIdx.z;  // 0 for the flat form
  const int M = a.flat ? a.items * a.rows : a.rows;
  const int r0 = (blockIdx.y * 4 + wave) * (16 * NR);
  if (r0 >= M) return;  // (no barrier in this kernel: a wave may leave)
  const int nch = a.K >> 5, nsb = (a.N + 15) >> 4;
  const bf16_t* xa[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = r0 + 16 * r + (lane & 15), rc = row < M ? row : M - 1;
    xa[r] = a.x + (long long)item * a.xbs + (long long)rc * a.ldx + 8 * (lane >> 4);
  }
  const lr_u32x4* bp[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int sb = blockIdx.x * NS + s < nsb ? blockIdx.x * NS + s : nsb - 1;  // (a sub-block past the end repeats the last one; not stored)
    bp[s] = (const lr_u32x4*)a.wl + ((long long)sb * nch) * 64 + lane;
  }
  lr_u32x4 ra[PD][NR], rb[PD][NS];
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    const int c = p < nch ? p : nch - 1;
#pragma unroll
    for (int r = 0; r < NR; ++r) ra[p][r] = *(const lr_u32x4*)(xa[r] + 32 * c);
#pragma unroll
    for (int s = 0; s < NS; ++s) rb[p][s] = bp[s][(long long)c * 64];
  }
  lr_f32x4 acc[NR][NS];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[r][s] = lr_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nch; c0 += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int c = c0 + p;
      lr_bf16x8 av[NR], bv[NS];
#pragma unroll
      for (int r = 0; r < NR; ++r) av[r] = __builtin_bit_cast(lr_bf16x8, ra[p][r]);
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[s] = __builtin_bit_cast(lr_bf16x8, rb[p][s]);
      {  // refill the slot (clamped past the end: unconditional loads, exact wait counts)
        const int cn = c + PD < nch ? c + PD : nch - 1;
#pragma unroll
        for (int r = 0; r < NR; ++r) ra[p][r] = *(const lr_u32x4*)(xa[r] + 32 * cn);
#pragma unroll
        for (int s = 0; s < NS; ++s) rb[p][s] = bp[s][(long long)cn * 64];
      }
      if (c < nch) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int s = 0; s < NS; ++s) acc[r][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[r], bv[s], acc[r][s], 0, 0, 0);
      }
    }
  }
  // epilogue: lane L holds rows r0 + 16 r + 4 (L / 16) + i, column 16 s + L % 16
  bf16_t* ob = a.out + (long long)item * a.obs;
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 16 * r + 4 * (lane >> 4) + i;
      const int b = a.flat ? row / a.rows : item, t = a.flat ? row - b * a.rows : row;
      const bool live = row < M && t < kk_len(a.len, b < a.items ? b : a.items - 1);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int col = (blockIdx.x * NS + s) * 16 + (lane & 15);
        float v = acc[r][s][i] + (a.bias ? a.bias[col < a.Nb ? col : 0] : 0.f);
        if (a.act == KK_ACT_GELU) v = lr_gelu(v);
        if (!live) v = 0.f;
        const float vn = __shfl_xor(v, 1);  // the neighbouring column: even lanes store a packed pair
        if (!(lane & 1) && row < M && col < a.N) {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          const bf2 pk = {(__bf16)v, (__bf16)vn};
          *(unsigned*)(ob + (long long)row * a.ldo + col) = __builtin_bit_cast(unsigned, pk);
        }
      }
    }
}

}  // namespace

int kk_launch_linear_rows_mfma(const KKLinMfmaArgs& a, hipStream_t st) {
  if (a.K % 32 || a.N % 2 || a.rows < 1 || a.items < 1) return kk_fail("linear_rows_mfma: bad shape");
  const int M = a.flat ? a.items * a.rows : a.rows;
  static int wide = -1;
  if (wide < 0) wide = getenv("KK_LINROWS_WIDE") ? 1 : 0;  // (A/B: 64 columns per wave at every size)
  if (M <= 256 && !wide) {
    const dim3 grid((a.N + 31) / 32, (M + 63) / 64, a.flat ? 1 : a.items);
    hipLaunchKernelGGL((linear_rows_mfma_kernel<2, 1, 12>), grid, dim3(256), 0, st, a);
  } else {  // (a 64-row x 64-column wave, <4, 4, 4>, was measured for full batches too: 40.7 vs 39.9 ms per B = 32 step against the tiled kernel -- the caller keeps that one above ~1000 rows)
    const dim3 grid((a.N + 63) / 64, (M + 63) / 64, a.flat ? 1 : a.items);
    hipLaunchKernelGGL((linear_rows_mfma_kernel<4, 1, 6>), grid, dim3(256), 0, st, a);
  }
  KK_CHECK_LAUNCH();
  return 0;
}
