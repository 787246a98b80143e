// Bidirectional LSTM recurrence (modules.py:152-239).  The input projection
// x @ Wx^T + (b_ih + b_hh) is a batched GEMM done by the conv/linear kernel; this kernel runs
// only the sequential part, one workgroup per (utterance, direction): 4H threads, thread g owns
// gate row g (order i, f, g, o as in mx.split(ifgo, 4), modules.py:179), h lives in LDS and the
// cell state in a register of threads 0..H-1.  Wh^T is streamed from L2 every step (1 MiB fp32
// for H = 256) -- no inter-workgroup hand-off, so no grid-level synchronisation is needed.
#include "kk_common.h"
#include "kk_kernels.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T>
__global__ __launch_bounds__(1024) void lstm_kernel(KKLstmArgs a) {
  __shared__ float h_s[256];
  __shared__ float g_s[1024];
  const int b = blockIdx.x, dir = blockIdx.y;
  const int g = threadIdx.x, H = a.H, G = 4 * H;
  const int L = kk_len(a.len, b);
  const float* wh = a.whT + (long long)dir * H * G + g;
  const float* xp = a.xproj + (long long)b * a.Lmax * 2 * G + (long long)dir * G + g;
  T* ob = (T*)a.out + (long long)b * a.obs + dir * H;
  float c = 0.f;
  if (g < H) h_s[g] = 0.f;
  __syncthreads();
  const int gate = g / H;
  for (int step = 0; step < L; ++step) {
    const int t = dir ? (L - 1 - step) : step;
    float acc = xp[(long long)t * 2 * G];
#pragma unroll 8
    for (int j = 0; j < H; ++j) acc = __builtin_fmaf(wh[(long long)j * G], h_s[j], acc);
    const float act = (gate == 2) ? tanhf(acc) : sigmoidf_(acc);
    g_s[g] = act;
    __syncthreads();
    if (g < H) {
      const float ig = g_s[g], fg = g_s[H + g], gg = g_s[2 * H + g], og = g_s[3 * H + g];
      c = fg * c + ig * gg;
      const float h = og * tanhf(c);
      h_s[g] = h;
      kk_st(ob + (long long)t * a.ldo + g, h);
    }
    __syncthreads();
  }
  // rows past the valid length are zero (what the B=1 reference call never sees)
  if (g < H)
    for (int t = L; t < a.Lmax; ++t) kk_st(ob + (long long)t * a.ldo + g, 0.f);
}

}  // namespace

int kk_launch_lstm(const KKLstmArgs& a, int B, int dtype, hipStream_t st) {
  if (B <= 0 || a.Lmax <= 0) return 0;
  if (a.H > 256 || a.H < 16 || (a.H & 15)) return kk_fail("lstm: H must be a multiple of 16, <= 256");
  dim3 grid(B, 2);
  if (dtype == KK_F32)
    hipLaunchKernelGGL(lstm_kernel<float>, grid, dim3(4 * a.H), 0, st, a);
  else
    hipLaunchKernelGGL(lstm_kernel<bf16_t>, grid, dim3(4 * a.H), 0, st, a);
  KK_CHECK_LAUNCH();
  return 0;
}
