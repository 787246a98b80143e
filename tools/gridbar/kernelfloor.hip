// Micro-benchmark: wall time per kernel of DEPENDENT small kernels in graph replay on gfx950 -- what a load -> compute -> store kernel
// costs beyond the 1.55 us of an empty launch, step by step (each kernel reads what its predecessor wrote).
//   hipcc --offload-arch=gfx950 -O3 -o kernelfloor kernelfloor.hip && ./kernelfloor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k_empty(float* p) { if (p == nullptr) p[0] = 0; }
// out[e] = in[e] + 1 (8192 elements: the residual stream of 8 rows x 1024)
__global__ __launch_bounds__(256) void k_copy(const float* in, float* out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  out[e] = in[e] + 1.0f;
}
// one thread: *p += 1
__global__ void k_one(float* p) { *p += 1.0f; }
// combine: out[e] = in[e] + sum of 16 slices
__global__ __launch_bounds__(256) void k_combine(const float* in, const float* part, float* out, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = part[(size_t)k * n + e];
  float t = in[e];
#pragma unroll
  for (int k = 0; k < 16; ++k) t += v[k];
  out[e] = t;
}
// staging: every workgroup reads the same 32 KB (8 x 1024 floats) into LDS, then writes 64 outputs
__global__ __launch_bounds__(512) void k_stage(const float* in, float* out) {
  __shared__ float xs[8192];
  const int tid = threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = ((const float4*)in)[tid + 512 * i];
#pragma unroll
  for (int i = 0; i < 4; ++i) ((float4*)xs)[tid + 512 * i] = v[i];
  __syncthreads();
  if (tid < 32) out[(blockIdx.x * 32 + tid) & 8191] = xs[tid * 7] + 1.0f;
}
// staging + weight stream: 128 KB per workgroup (16 x 16 B per lane, all requested before the staging), summed, 32 outputs
__device__ unsigned long long g_wgts[2 * 1024];  // per-workgroup start / end of the LAST k_stream launch (wall clock, 100 MHz)
template <int NLOAD, int PATTERN = 0>
__global__ __launch_bounds__(512) void k_stream(const float* in, const uint4* w, float* out) {
  __shared__ float xs[8192];
  extern __shared__ float dyn[];
  if (in == nullptr) dyn[threadIdx.x] = 0.f;
  const int tid = threadIdx.x;
  if (tid == 0) g_wgts[2 * blockIdx.x] = wall_clock64();
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* wp = (const u32x4*)w + (size_t)blockIdx.x * 512 * NLOAD + tid;
  u32x4 r[NLOAD];
  if (PATTERN == 1) {  // the fragment order of gemvm_kernel: wave w takes the 4-KB chunks w, w + 8, ...; 4 consecutive 1-KB loads per chunk
    const u32x4* wq = (const u32x4*)w + (size_t)blockIdx.x * 512 * NLOAD + (tid & 63);
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) r[i] = __builtin_nontemporal_load(wq + (size_t)(((wave + 8 * (i >> 2)) * 4 + (i & 3)) * 64));
  } else
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) r[i] = __builtin_nontemporal_load(wp + 512 * i);
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = ((const float4*)in)[tid + 512 * i];
#pragma unroll
  for (int i = 0; i < 4; ++i) ((float4*)xs)[tid + 512 * i] = v[i];
  __syncthreads();
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) s += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
  float t = xs[(tid * 5) & 8191] + (float)(s & 1);
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  __shared__ float red[8];
  if ((tid & 63) == 0) red[tid >> 6] = t;
  __syncthreads();
  if (tid < 32) out[(blockIdx.x * 32 + tid) & 8191] = red[tid & 7] * 0.0f + 1.0f;
  if (tid == 0) g_wgts[2 * blockIdx.x + 1] = wall_clock64();
}
#include <vector>
#include <algorithm>
static void print_wg_times(const char* what, int nwg) {
  std::vector<unsigned long long> t(2 * 1024);
  CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_wgts), t.size() * 8));
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < nwg; ++w) t0 = t[2 * w] < t0 ? t[2 * w] : t0;
  std::vector<double> st_, en_;
  for (int w = 0; w < nwg; ++w) { st_.push_back((double)(t[2 * w] - t0) / 100.0); en_.push_back((double)(t[2 * w + 1] - t0) / 100.0); }
  std::sort(st_.begin(), st_.end()); std::sort(en_.begin(), en_.end());
  printf("   %s: workgroup starts min %.2f p50 %.2f max %.2f | ends min %.2f p25 %.2f p50 %.2f p75 %.2f max %.2f us\n", what, st_[0], st_[nwg / 2], st_[nwg - 1], en_[0], en_[nwg / 4], en_[nwg / 2],
         en_[3 * nwg / 4], en_[nwg - 1]);
}

static hipStream_t st;
static hipEvent_t e0, e1;

void time_graph(const char* name, int n, const std::function<void(int)>& launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-64s %7.3f us each\n", name, best * 1000.f / n);
  fflush(stdout);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *a, *b, *part; uint4* w;
  CK(hipMalloc(&a, 8192 * 4)); CK(hipMalloc(&b, 8192 * 4)); CK(hipMalloc(&part, (size_t)16 * 8192 * 4));
  const size_t wbytes = (size_t)256 * 512 * 16 * 16 * 8;  // 8 distinct 33.5-MB matrices (a frame never re-reads a matrix from cache)
  CK(hipMalloc(&w, wbytes));
  CK(hipMemset(a, 0, 8192 * 4)); CK(hipMemset(b, 0, 8192 * 4)); CK(hipMemset(part, 0, (size_t)16 * 8192 * 4)); CK(hipMemset(w, 0, wbytes));
  const int n = 1000;
  auto pp = [&](int i, float*& in, float*& out) { in = (i & 1) ? b : a; out = (i & 1) ? a : b; };
  time_graph("empty, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, a); });
  time_graph("one thread: *p += 1", n, [&](int) { hipLaunchKernelGGL(k_one, dim3(1), dim3(1), 0, st, a); });
  time_graph("copy 8192 floats (32 x 256), reads the predecessor's output", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_copy, dim3(32), dim3(256), 0, st, in, out); });
  time_graph("combine 16 slices + residual (32 x 256)", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_combine, dim3(32), dim3(256), 0, st, in, part, out, 8192); });
  time_graph("stage 32 KB into LDS, 256 x 512", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stage, dim3(256), dim3(512), 0, st, in, out); });
  time_graph("stage 32 KB into LDS, 96 x 512", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stage, dim3(96), dim3(512), 0, st, in, out); });
  time_graph("stage + stream 32 KB / workgroup (4 loads per lane), 96 x 512", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stream<4>, dim3(96), dim3(512), 0, st, in, w + (size_t)(i & 7) * (wbytes / 16 / 8), out); });
  time_graph("stage + stream 64 KB / workgroup (8 loads per lane), 256 x 512", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stream<8>, dim3(256), dim3(512), 0, st, in, w + (size_t)(i & 7) * (wbytes / 16 / 8), out); });
  time_graph("stage + stream 128 KB / workgroup (16 loads per lane), 256 x 512", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stream<16>, dim3(256), dim3(512), 0, st, in, w + (size_t)(i & 7) * (wbytes / 16 / 8), out); });
  print_wg_times("128 KB / workgroup", 256);
  CK(hipFuncSetAttribute((const void*)k_stream<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  time_graph("stage + stream 128 KB / workgroup, + 44 KB dynamic LDS (76 KB)", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_stream<16, 0>), dim3(256), dim3(512), 44 * 1024, st, in, w + (size_t)(i & 7) * (wbytes / 16 / 8), out); });
  print_wg_times("76 KB LDS", 256);
  CK(hipMemset(w, 0x3b, wbytes));
  time_graph("stage + stream 128 KB / workgroup, non-zero data", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_stream<16, 0>), dim3(256), dim3(512), 0, st, in, w + (size_t)(i & 7) * (wbytes / 16 / 8), out); });
  print_wg_times("non-zero data", 256);
  time_graph("stage + stream 128 KB / workgroup, fragment-order addresses", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL((k_stream<16, 1>), dim3(256), dim3(512), 0, st, in, w + (size_t)(i & 7) * (wbytes / 16 / 8), out); });
  {  // the same stream from a 4-GB matrix pool, a different 33.5-MB matrix per launch (a frame touches 2.2 GB of weights: TLB reach?)
    uint4* big;
    const size_t one = (size_t)256 * 512 * 16 * 16, nmat = 119;
    CK(hipMalloc(&big, one * nmat));
    CK(hipMemset(big, 0, one * nmat));
    time_graph("stage + stream 128 KB / workgroup, 119 matrices in 4 GB", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stream<16>, dim3(256), dim3(512), 0, st, in, big + (size_t)((i * 37) % nmat) * (one / 16), out); });
    time_graph("stage + stream 32 KB / workgroup (96 x 512), 119 matrices in 4 GB", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stream<4>, dim3(96), dim3(512), 0, st, in, big + (size_t)((i * 37) % nmat) * (one / 16), out); });
    time_graph("stage + stream 128 KB / workgroup, 7 matrices in 4 GB (222 MB cycle)", n, [&](int i) { float *in, *out; pp(i, in, out); hipLaunchKernelGGL(k_stream<16>, dim3(256), dim3(512), 0, st, in, big + (size_t)(i % 7) * (one / 16), out); });
  }
  return 0;
}
