// The matrix-core GEMV of the CSM single-token steps (mlx-audio_amd/csrc/kk_csm_gemvm.h) timed in ISOLATION on the MLP half of a depth-decoder
// layer: gate|up (K = 1024, N = 16384, RMSNorm prologue) -> down (K = 8192 in 8 slices, N = 1024, SwiGLU prologue, partial tiles) -> combine,
// as one graph of `n` dependent triples over a pool of distinct matrices (a frame never re-reads a matrix from a cache).  Prints the wall
// time per kernel of the chain and the in-kernel marks (span, workgroup 0's phases) of the two GEMVs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o gemvm_bench gemvm_bench.hip && ./gemvm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

#define KK_TS_PER_WG
namespace {
#include "../../mlx-audio_amd/csrc/kk_csm_gemvm.h"

__global__ __launch_bounds__(256) void combine_kernel(const float* part, int KS, long long pss, long long n, float* h) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float t = h[e];
  for (int ks = 0; ks < KS; ++ks) t += part[(long long)ks * pss + e];
  h[e] = t * 0.5f;
}
__global__ void fill_kernel(uint32_t* p, size_t n, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
}  // namespace

int main(int argc, char** argv) {
  const int D = 1024, I = 8192, M = 8, nmat = 24, n = 200, TSW = 8 + 3 * 256;
  hipStream_t st; CK(hipStreamCreate(&st));
  float *h, *nw, *gu, *part; uint16_t *wgu, *wdn; unsigned long long* ts;
  CK(hipMalloc(&h, M * D * 4)); CK(hipMalloc(&nw, D * 4)); CK(hipMalloc(&gu, (size_t)M * 2 * I * 4)); CK(hipMalloc(&part, (size_t)8 * M * D * 4));
  const size_t gub = (size_t)D * 2 * I, dnb = (size_t)I * D;  // elements per matrix
  CK(hipMalloc(&wgu, gub * 2 * nmat)); CK(hipMalloc(&wdn, dnb * 2 * nmat));
  CK(hipMalloc(&ts, (size_t)3 * n * TSW * 8));
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, (uint32_t*)wgu, gub * nmat / 2, 0x3a803b00u);  // small bf16 values
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, (uint32_t*)wdn, dnb * nmat / 2, 0x3a803b00u);
  hipLaunchKernelGGL(fill_kernel, dim3(32), dim3(256), 0, st, (uint32_t*)h, (size_t)M * D, 0x3f000000u);
  hipLaunchKernelGGL(fill_kernel, dim3(4), dim3(256), 0, st, (uint32_t*)nw, (size_t)D, 0x3f800000u);
  CK(hipStreamSynchronize(st));
  CK(hipFuncSetAttribute((const void*)gemvm_kernel<4, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)gemvm_kernel<2, 2, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const bool marks = argc > 2 ? atoi(argv[2]) != 0 : true;  // in-kernel marks (their atomics cost ~1 us per kernel)  // 0 gate|up -> down -> combine; 1 gate|up only; 2 gate|up -> down
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) {
    FGArgs a; memset(&a, 0, sizeof a);
    a.x = h; a.xrs = D; a.nw = nw; a.eps = 1e-5f; a.w = wgu + (size_t)(i % nmat) * gub; a.K = D; a.kper = D; a.N = 2 * I; a.M = M; a.out = gu; a.ors = 2 * I;
    a.ts = marks ? ts + (size_t)(3 * i) * TSW : nullptr; a.ts_id = 1;
    hipLaunchKernelGGL((gemvm_kernel<4, 1, 0, 1>), dim3(2 * I / 64, 1, 1), dim3(512), gm_lds_bytes(4, D), st, a);
    if (mode == 1) continue;
    memset(&a, 0, sizeof a);
    a.x = gu; a.xrs = 2 * I; a.w = wdn + (size_t)(i % nmat) * dnb; a.K = I; a.kper = I / 8; a.N = D; a.M = M; a.out = part; a.ors = D; a.pss = (long long)M * D;
    a.ts = marks ? ts + (size_t)(3 * i + 1) * TSW : nullptr; a.ts_id = 2;
    hipLaunchKernelGGL((gemvm_kernel<2, 2, 2, 1>), dim3(D / 32, 8, 1), dim3(512), gm_lds_bytes(2, I / 8), st, a);
    if (mode == 2) continue;
    hipLaunchKernelGGL(combine_kernel, dim3(M * D / 256), dim3(256), 0, st, part, 8, (long long)M * D, (long long)M * D, h);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> init((size_t)3 * n * TSW, 0);
  for (int i = 0; i < 3 * n; ++i) init[(size_t)i * TSW + 1] = ~0ull;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemcpy(ts, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<unsigned long long> t((size_t)3 * n * TSW);
  CK(hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost));
  printf("mode %d chain%s: %.2f us per iteration\n", mode, marks ? "" : " (no marks)", best * 1000.f / n);
  if (!marks) return 0;
  double sp[2] = {0, 0}, stg[2] = {0, 0}, con[2] = {0, 0}, gap[2] = {0, 0};
  int cnt = 0;
  for (int i = 20; i < n; ++i) {
    for (int k = 0; k < 2; ++k) {
      const unsigned long long* s = &t[(size_t)(3 * i + k) * TSW];
      sp[k] += (double)(s[2] - s[1]) / 100.0;
      stg[k] += (double)(s[3] - s[4]) / 100.0;
      con[k] += (double)(s[5] - s[4]) / 100.0;
    }
    const unsigned long long* a0 = &t[(size_t)(3 * i) * TSW];
    const unsigned long long* a1 = &t[(size_t)(3 * i + 1) * TSW];
    gap[1] += (double)(a1[1] - a0[2]) / 100.0;
    ++cnt;
  }
  printf("gate|up: span %.2f us, workgroup 0 staged at %.2f, weights consumed at %.2f\n", sp[0] / cnt, stg[0] / cnt, con[0] / cnt);
  printf("down   : span %.2f us, workgroup 0 staged at %.2f, weights consumed at %.2f, gap after gate|up %.2f\n", sp[1] / cnt, stg[1] / cnt, con[1] / cnt, gap[1] / cnt);
  {  // workgroup start / end distribution of one gate|up launch (relative to the earliest start), sorted
    const unsigned long long* s = &t[(size_t)(3 * 100) * TSW];
    std::vector<double> st_, en_;
    for (int w = 0; w < 256; ++w) { st_.push_back((double)(s[8 + 2 * w] - s[1]) / 100.0); en_.push_back((double)(s[9 + 2 * w] - s[1]) / 100.0); }
    std::sort(st_.begin(), st_.end()); std::sort(en_.begin(), en_.end());
    printf("gate|up workgroup starts  (us): min %.2f p25 %.2f p50 %.2f p75 %.2f max %.2f\n", st_[0], st_[64], st_[128], st_[192], st_[255]);
    {  // placement: workgroups per (XCC, SE, CU)
      std::vector<int> cnt(8 * 64 * 16, 0);
      int maxper = 0, used = 0;
      for (int w = 0; w < 256; ++w) {
        const unsigned long long v = s[8 + 2 * 256 + w];
        const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32) & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        int& c = cnt[(xcc * 64 + se * 2 + sh) * 16 + cu];
        if (c++ == 0) ++used;
        if (c > maxper) maxper = c;
      }
      printf("gate|up placement: 256 workgroups on %d distinct CUs, at most %d on one CU\n", used, maxper);
      // end time by CU occupancy
      double e1 = 0, e2 = 0; int n1 = 0, n2 = 0;
      for (int w = 0; w < 256; ++w) {
        const unsigned long long v = s[8 + 2 * 256 + w];
        const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32) & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const int c = cnt[(xcc * 64 + se * 2 + sh) * 16 + cu];
        const double en = (double)(s[9 + 2 * w] - s[1]) / 100.0;
        if (c == 1) { e1 += en; ++n1; } else { e2 += en; ++n2; }
      }
      printf("   mean end of workgroups alone on their CU: %.2f us (%d), sharing a CU: %.2f us (%d)\n", n1 ? e1 / n1 : 0.0, n1, n2 ? e2 / n2 : 0.0, n2);
    }
    printf("gate|up workgroup ends    (us): min %.2f p25 %.2f p50 %.2f p75 %.2f max %.2f\n", en_[0], en_[64], en_[128], en_[192], en_[255]);
  }
  return 0;
}
