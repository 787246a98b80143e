// Micro-benchmark: cost of a device-wide barrier inside one persistent kernel on gfx950 (8 XCDs, one L2 each), with a realistic
// data exchange around it (every workgroup writes 32 floats, then reads everybody's: 32 KB).  Decides whether the CSM depth decoder
// (31 steps x 4 layers x 4-5 dependent GEMV phases per 80-ms frame, today ~840 launches at a ~6.6 us floor) should be ONE kernel.
//   hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr unsigned SPIN_LIMIT = 1u << 22;  // every wave leaves the spin: a lost barrier becomes an error flag, never a hang

// mode 0: fence by every thread + one counter;  mode 1: barrier only (no data);  mode 2: per-XCD counter then a global one
template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* err) {
  if (MODE != 1) __threadfence();
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { good = 0; *err = 1; break; }
    }
    ok = good;
  }
  __syncthreads();
  if (MODE != 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* ctr, float* buf, int iters, int* err, float* sink) {
  const unsigned nwg = gridDim.x;
  const int wg = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float* b = buf + (size_t)(it & 1) * nwg * 32;
    if (MODE != 1 && tid < 32) b[wg * 32 + tid] = (float)(it + 1);
    if (!grid_barrier<MODE>(ctr, (unsigned)(it + 1) * nwg, err)) return;
    if (MODE != 1) {
      float s = 0.f;
      for (int i = tid; i < (int)nwg * 32; i += 256) s += b[i];
      // every element must read it+1
      const float want = (float)(it + 1) * (float)((nwg * 32 + 255 - tid) / 256);
      if (s != want) *err = 2;
      acc += s;
    }
  }
  if (acc == -1.f) sink[0] = acc;
}

// empty kernel chain for the launch floor next to it
__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 0; }

template <int MODE>
void run(const char* name, int nwg, int iters) {
  unsigned* ctr; float* buf; int* err; float* sink;
  CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&buf, (size_t)2 * nwg * 32 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  int herr = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(ctr, 0, 256)); CK(hipMemset(err, 0, 4)); CK(hipMemset(buf, 0, (size_t)2 * nwg * 32 * 4));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bar_kernel<MODE>, dim3(nwg), dim3(256), 0, 0, ctr, buf, iters, err, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    if (herr) break;
  }
  printf("%-34s nwg %4d  %7.3f us per barrier  err %d\n", name, nwg, best * 1000.f / iters, herr);
  fflush(stdout);
  CK(hipFree(ctr)); CK(hipFree(buf)); CK(hipFree(err)); CK(hipFree(sink));
}

int main() {
  const int iters = 2000;
  for (int nwg : {64, 128, 256, 512}) {
    run<1>("barrier only", nwg, iters);
    run<0>("fence + barrier + 32 KB exchange", nwg, iters);
  }
  // launch floor: 2000 empty kernels back to back, eager and as one graph
  {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* p; CK(hipMalloc(&p, 4));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, p);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("empty kernels, eager: %.3f us each\n", ms * 1000.f / iters);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, p);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("empty kernels, graph replay: %.3f us each\n", ms * 1000.f / iters);
    }
  }
  return 0;
}
