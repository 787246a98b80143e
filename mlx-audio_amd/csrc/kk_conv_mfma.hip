// bf16 MFMA implicit-GEMM convolution for frames-major tensors (gfx950, v_mfma_f32_32x32x16_bf16).
//
//   out[b][q][n] = sum_t sum_ci  W[t][n][ci] * X[b][q + off_t][ci]       (stride 1, any dilation)
//   transposed conv = `stride` independent phase convolutions with 2 taps each (polyphase form, see kk_conv.hip)
//
// GEMM view per workgroup: M = 128 output rows, N = 128 output channels, K = taps x Cin walked in slabs of 64
// channels.  A = X rows (positions), B = W rows (output channels); both are k-contiguous in LDS so a lane's MFMA
// fragment (8 consecutive k) is ONE ds_read_b128.  LDS rows are padded 128 B -> 144 B: 16 consecutive rows then start
// on 16 distinct 4-bank groups, which makes the b128 fragment reads conflict-free (banks = (addr/4) % 64).
//
//   * the X slab [128 + halo rows][64 ch] is loaded once per channel slab and re-used by every tap as a shifted window
//   * W tiles [128 n][64 ci] are double-buffered in LDS and prefetched through registers one tap ahead, so the
//     global (L2-resident) weight loads run under the 16 MFMAs per wave of the current tap: one barrier per tap
//   * 4 waves as 2 x 2, each 64 x 64 outputs = 2 x 2 accumulators of 32 x 32 (64 acc VGPRs)
//   * epilogue through LDS (fp32 tile) so bias / activation / residual / scale / accumulate / length mask are applied
//     on coalesced 16-byte rows and the result is rounded to bf16 exactly once.
#include "kk_common.h"
#include <stdlib.h>

#include "kk_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, CK = 64;
constexpr int XLD = CK + 8;        // elements per LDS row (144 B)
constexpr int MAX_HALO = 64;       // (Kw-1)*dil <= 50 on this path
constexpr int XROWS = BM + MAX_HALO;
constexpr int CLD = BN;            // fp32 epilogue tile pitch (128 x 128 x 4 B = exactly 64 KiB)

constexpr int XS_BYTES = XROWS * XLD * 2;          // 27648
constexpr int WS_BYTES = BN * XLD * 2;             // 18432 per buffer
constexpr int MAIN_BYTES = XS_BYTES + 2 * WS_BYTES;  // 64512
constexpr int EPI_BYTES = BM * CLD * 4;            // 67584
constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

union U16 {
  uint4 u;
  bf16_t h[8];
};
union U32x8 {
  uint4 u[2];
  float f[8];
};

template <typename TO>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(KKMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xs = (bf16_t*)smem;
  bf16_t* Ws0 = (bf16_t*)(smem + XS_BYTES);
  bf16_t* Ws1 = (bf16_t*)(smem + XS_BYTES + WS_BYTES);
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

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (tile_live) {
    const bf16_t* xb = a.x + (long long)b * a.xbs;
    const int nchunk = a.CinP / CK;
    const int nit = nchunk * ntaps;

    // ---- loaders (global -> registers) --------------------------------------------------------------
    uint4 xreg[6];  // up to 192 rows x 8 chunks of 16 B = 1536 chunks / 256 threads
    uint4 wreg[4];  // 128 rows x 8 chunks = 1024 chunks / 256 threads
    // Unconditional loads from clamped addresses; validity is applied when the registers are written to LDS.  A load under
    // a data-dependent branch (or a select the optimiser turns into one) makes hipcc wait vmcnt(0) right behind it, which
    // serialises the six loads (one HBM round trip each); the empty asm with a memory clobber keeps the loads from being
    // sunk towards their use.
    const int lin_hi = Lin > 0 ? Lin - 1 : 0;
    unsigned xok = 0;
    float4 nA[2], nB[2], nAl[2];  // fused AdaIN parameters of this thread's 8 channels of the slab
    const bool has_nrm = a.nrm_a != nullptr;
    auto load_x = [&](int chunk) {
      xok = 0;
      if (has_nrm) {
        const int c = chunk * CK + (tid & 7) * 8;  // (id & 7) == (tid & 7): 256 is a multiple of 8
        const float* pa = a.nrm_a + (long long)b * a.nrm_stride + c;
        const float* pb = a.nrm_b + (long long)b * a.nrm_stride + c;
        nA[0] = *(const float4*)pa; nA[1] = *(const float4*)(pa + 4);
        nB[0] = *(const float4*)pb; nB[1] = *(const float4*)(pb + 4);
        if (a.nrm_act == KK_ACT_SNAKE) {
          float al[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) al[k] = (c + k) < a.nrm_C ? a.nrm_alpha[c + k] : 1.0f;
          nAl[0] = make_float4(al[0], al[1], al[2], al[3]);
          nAl[1] = make_float4(al[4], al[5], al[6], al[7]);
        }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int id = i * 256 + tid;
        const int r = id >> 3, c8 = (id & 7) * 8;
        int row = q0 + min_off + r;
        const bool ok0 = row >= 0 && r < xrows;
        if (a.in_shift) row >>= a.in_shift;
        if (ok0 && row < Lin) xok |= 1u << i;
        const int rc = row < 0 ? 0 : (row > lin_hi ? lin_hi : row);
        xreg[i] = *(const uint4*)(xb + (long long)rc * a.ldx + chunk * CK + c8);
      }
      asm volatile("" ::: "memory");
    };
    auto store_x = [&]() {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int id = i * 256 + tid;
        const int r = id >> 3, c8 = (id & 7) * 8;
        if (r < xrows) {
          U16 t;
          const unsigned msk = (xok >> i) & 1u ? 0xFFFFFFFFu : 0u;
          t.u = xreg[i];
          if (has_nrm) {  // y = act(x * A + B): AdaIN1d + Snake / LeakyReLU (istftnet.py:333-337,382) applied while staging
            const float pa[8] = {nA[0].x, nA[0].y, nA[0].z, nA[0].w, nA[1].x, nA[1].y, nA[1].z, nA[1].w};
            const float pb[8] = {nB[0].x, nB[0].y, nB[0].z, nB[0].w, nB[1].x, nB[1].y, nB[1].z, nB[1].w};
            const float pl[8] = {nAl[0].x, nAl[0].y, nAl[0].z, nAl[0].w, nAl[1].x, nAl[1].y, nAl[1].z, nAl[1].w};
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
          t.u = make_uint4(t.u.x & msk, t.u.y & msk, t.u.z & msk, t.u.w & msk);  // padding rows stay exactly zero
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
    auto load_w = [&](int it) {
      if ((a.dbg & 1) && it > 1) return;  // timing experiment: no W traffic after the prologue
      const int chunk = it / ntaps, tap = it - chunk * ntaps;
      const bf16_t* wt = a.w + ((long long)(widx0 + tap * wstep) * a.CoutP + n0) * a.CinP + chunk * CK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = i * 256 + tid;
        const int n = id >> 3, c8 = (id & 7) * 8;
        wreg[i] = *(const uint4*)(wt + (long long)n * a.CinP + c8);
      }
      asm volatile("" ::: "memory");
    };
    auto store_w = [&](bf16_t* Ws) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = i * 256 + tid;
        const int n = id >> 3, c8 = (id & 7) * 8;
        *(uint4*)(Ws + n * XLD + c8) = wreg[i];
      }
    };

    // ---- prologue
    load_x(0);
    load_w(0);
    store_x();
    store_w(Ws0);
    __syncthreads();
    if (nit > 1) load_w(1);

    const int arow = wr * 64 + (lane & 31);       // + mi*32 + tap shift
    const int brow = wc * 64 + (lane & 31);       // + ni*32
    const int kofs = 8 * (lane >> 5);

    for (int it = 0; it < nit; ++it) {
      const int chunk = it / ntaps, tap = it - chunk * ntaps;
      const bf16_t* Ws = (it & 1) ? Ws1 : Ws0;
      // prefetch the next channel slab of X at the FIRST tap of this slab: it has ntaps iterations to land
      const bool last_tap = tap == ntaps - 1;
      if (tap == 0 && chunk + 1 < nchunk && !(a.dbg & 2)) load_x(chunk + 1);

      const int shift = (off0 + tap * dstep) - min_off;  // row shift of this tap inside the X slab
      const bf16_t* xa = Xs + (arow + shift) * XLD + kofs;
      const bf16_t* wb = Ws + brow * XLD + kofs;
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        const bf16x8 a0 = *(const bf16x8*)(xa + ks * 16);
        const bf16x8 a1 = *(const bf16x8*)(xa + 32 * XLD + ks * 16);
        const bf16x8 b0 = *(const bf16x8*)(wb + ks * 16);
        const bf16x8 b1 = *(const bf16x8*)(wb + 32 * XLD + ks * 16);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      }
      if (it + 1 < nit) {
        store_w((it & 1) ? Ws0 : Ws1);  // buffer last read in iteration it-1; every wave has passed that barrier
        if (last_tap) {
          __syncthreads();  // all waves are done with the X slab
          store_x();
        }
        __syncthreads();
        if (it + 2 < nit) load_w(it + 2);
      }
    }
    __syncthreads();  // main-loop LDS is dead; the epilogue tile aliases it
  }

  // ---- epilogue: accumulators -> fp32 LDS tile -> coalesced rows ----------------------------------------
  if (tile_live) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = wc * 64 + ni * 32 + (lane & 31);
        const float bs = a.bias ? a.bias[n0 + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          float v = acc[mi][ni][r] + bs;
          if (a.act == KK_ACT_LRELU) v = v > 0.f ? v : v * a.act_slope;
          else if (a.act == KK_ACT_GELU) v = gelu_exact(v);
          Cs[row * CLD + col] = v;
        }
      }
    __syncthreads();
  }
  TO* ob = (TO*)a.out + (long long)b * a.obs;
  const TO* rb = a.res ? (const TO*)a.res + (long long)b * a.rbs : nullptr;
  const int n = n0 + (tid & 15) * 8;  // this thread's 8 output channels (same for all its rows)
  const int lo_hi = a.Lo_rows - 1;
  constexpr int VEC = sizeof(TO) == 2 ? 1 : 2;  // 16-byte vectors per 8 outputs
  float st_s[8], st_q[8];  // column sums / sums of squares of the values this thread stores
#pragma unroll
  for (int k = 0; k < 8; ++k) st_s[k] = st_q[k] = 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    int opv[4];
    bool wr_ok[4], live[4];
    uint4 rres[4][VEC], rold[4][VEC];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ((half * 4 + i) * 256 + tid) >> 4;
      const int q = q0 + row;
      const int op = a.mode == KK_CONV ? q : phase + a.stride * q;
      opv[i] = op < 0 ? 0 : (op > lo_hi ? lo_hi : op);
      wr_ok[i] = q < a.Q && op < a.Lo_rows && n < a.Cout;
      live[i] = tile_live && op < Lout;
    }
    const int nc = n < a.Cout ? n : 0;  // clamped channel for the unconditional loads
    if (rb) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) rres[i][v] = *((const uint4*)(rb + (long long)opv[i] * a.ldr + nc) + v);
    }
    if (a.accumulate) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) rold[i][v] = *((const uint4*)(ob + (long long)opv[i] * a.ldo + nc) + v);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ((half * 4 + i) * 256 + tid) >> 4;
      float v[8];
      if (tile_live) {
        const float4 c0 = *(const float4*)(Cs + row * CLD + (tid & 15) * 8);
        const float4 c1 = *(const float4*)(Cs + row * CLD + (tid & 15) * 8 + 4);
        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
      }
      if (rb) {
        if (sizeof(TO) == 2) {
          U16 t;
          t.u = rres[i][0];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += (float)t.h[k];
        } else {
          U32x8 t;
          t.u[0] = rres[i][0];
          t.u[1] = rres[i][VEC - 1];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += t.f[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= a.scale;
      if (a.accumulate) {
        if (sizeof(TO) == 2) {
          U16 t;
          t.u = rold[i][0];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += (float)t.h[k];
        } else {
          U32x8 t;
          t.u[0] = rold[i][0];
          t.u[1] = rold[i][VEC - 1];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += t.f[k];
        }
      }
      if (!live[i]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
      }
      if (wr_ok[i]) {
        TO* dst = ob + (long long)opv[i] * a.ldo + n;
        if (sizeof(TO) == 2) {
          U16 t;
#pragma unroll
          for (int k = 0; k < 8; ++k) t.h[k] = (bf16_t)v[k];
          *(uint4*)dst = t.u;
          if (a.stat_part) {  // statistics of what the consumer will read (the bf16-rounded values)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float r = (float)t.h[k];
              st_s[k] += r;
              st_q[k] = __builtin_fmaf(r, r, st_q[k]);
            }
          }
        } else {
          *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          *(float4*)((float*)dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
    }
  }
  if (a.stat_part) {
    // rows of one column group live in threads tid = rg*16 + cg: reduce rg over the wave by shuffles (xor 16, 32),
    // then over the 4 waves through LDS; one deterministic partial per (utterance, tile, column)
    __syncthreads();  // every wave is done reading the Cs tile
    float* red = (float*)smem;  // [4 waves][2][128]
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
    const int which = tid >> 7, col = tid & 127;  // threads 0..127 -> sums, 128..255 -> sums of squares
    if (n0 + col < a.Cout) {
      const float v = red[(0 * 2 + which) * 128 + col] + red[(1 * 2 + which) * 128 + col] + red[(2 * 2 + which) * 128 + col] +
                      red[(3 * 2 + which) * 128 + col];
      const int tile = blockIdx.x * nphase + phase;
      a.stat_part[(((long long)b * a.stat_ntiles + tile) * 2 + which) * a.Cout + n0 + col] = v;
    }
  }
}

}  // namespace

bool kk_mfma_eligible(int Cin, int Cout, int Kw, int mode, int stride, int dil) {
  if (Cout % 8 != 0 || Cout < 64 || Cin < 32) return false;
  if (mode == KK_CONV && stride != 1) return false;
  const int ntaps = mode == KK_CONV ? Kw : kk_cdiv(Kw, stride);
  const int halo = mode == KK_CONV ? (Kw - 1) * dil : (ntaps - 1);
  return halo <= MAX_HALO;
}

int kk_launch_conv_mfma(const KKMfmaArgs& a, int B, int out_dtype, hipStream_t st) {
  if (a.Q <= 0 || B <= 0) return 0;
  if (a.CinP % CK != 0 || a.CoutP % BN != 0) return kk_fail("conv_mfma: CinP must be a multiple of 64 and CoutP of 128");
  if (a.ldx % 8 != 0 || a.ldo % 8 != 0 || (a.res && a.ldr % 8 != 0)) return kk_fail("conv_mfma: row pitches must be multiples of 8 elements");
  if (((uintptr_t)a.x & 15) || ((uintptr_t)a.out & 15) || ((uintptr_t)a.res & 15) || ((uintptr_t)a.w & 15))
    return kk_fail("conv_mfma: pointers must be 16-byte aligned");
  if (a.ldx < a.CinP) return kk_fail("conv_mfma: input pitch smaller than the padded channel count");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_mfma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)conv_mfma_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  const int nphase = a.mode == KK_CONVT ? a.stride : 1;
  dim3 grid(kk_cdiv(a.Q, BM), a.CoutP / BN, B * nphase);
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("KK_MFMA_DBG"); dbg = e ? atoi(e) : 0; }
  KKMfmaArgs a2 = a;
  a2.dbg = dbg;
  if (out_dtype == KK_BF16)
    hipLaunchKernelGGL(conv_mfma_kernel<bf16_t>, grid, dim3(256), LDS_BYTES, st, a2);
  else
    hipLaunchKernelGGL(conv_mfma_kernel<float>, grid, dim3(256), LDS_BYTES, st, a2);
  KK_CHECK_LAUNCH();
  return 0;
}
