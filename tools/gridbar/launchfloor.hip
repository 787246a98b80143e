// Micro-benchmark: what sets the per-kernel floor of a chain of dependent launches in graph replay on gfx950?  Empty kernels differing in
// ONE launch property each (dynamic LDS, VGPR allocation, grid size, kernarg size, scratch, distinct code objects).
//   hipcc --offload-arch=gfx950 -O3 -o launchfloor launchfloor.hip && ./launchfloor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Big { float* p; long long a[24]; };

__global__ __launch_bounds__(256) void k_plain(float* p) { if (p == nullptr) p[0] = 0; }
__global__ __launch_bounds__(256) void k_lds(float* p) { extern __shared__ float s[]; if (p == nullptr) p[0] = s[threadIdx.x]; }
__global__ __launch_bounds__(256) void k_vgpr(float* p) { asm volatile("v_mov_b32 v200, 0" ::: "v200"); if (p == nullptr) p[0] = 0; }
__global__ __launch_bounds__(256) void k_big(Big b) { if (b.p == nullptr) b.p[0] = (float)b.a[23]; }
__global__ __launch_bounds__(256) void k_scratch(float* p, int n) {
  volatile float a[64];
  for (int i = 0; i < n; ++i) a[i & 63] = (float)i;
  if (p == nullptr) p[0] = a[n & 63];
}
__global__ __launch_bounds__(256) void k_store(float* p) { p[(size_t)blockIdx.x * 256 + threadIdx.x] = 1.0f; }
template <int I> __global__ __launch_bounds__(256) void k_multi(float* p) { if (p == nullptr) p[0] = (float)I; }

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
  printf("%-52s %7.3f us each (graph replay, %d nodes)\n", name, best * 1000.f / n, n);
  fflush(stdout);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* p; CK(hipMalloc(&p, (size_t)4096 * 256 * 4));
  const int n = 1000;
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  time_graph("plain, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, p); });
  time_graph("plain, 1 x 64", n, [&](int) { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, p); });
  time_graph("plain, 1024 x 256", n, [&](int) { hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, st, p); });
  time_graph("plain, 4096 x 256", n, [&](int) { hipLaunchKernelGGL(k_plain, dim3(4096), dim3(256), 0, st, p); });
  time_graph("dynamic LDS 32 KB, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 32 * 1024, st, p); });
  time_graph("dynamic LDS 68 KB, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 68 * 1024, st, p); });
  time_graph("dynamic LDS 68 KB, 1024 x 256", n, [&](int) { hipLaunchKernelGGL(k_lds, dim3(1024), dim3(256), 68 * 1024, st, p); });
  time_graph("dynamic LDS 132 KB, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 132 * 1024, st, p); });
  time_graph("200 VGPRs, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_vgpr, dim3(256), dim3(256), 0, st, p); });
  time_graph("200 VGPRs, 1024 x 256", n, [&](int) { hipLaunchKernelGGL(k_vgpr, dim3(1024), dim3(256), 0, st, p); });
  Big b; b.p = p;
  time_graph("200-byte kernarg, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, st, b); });
  time_graph("scratch 256 B/lane, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_scratch, dim3(256), dim3(256), 0, st, p, 0); });
  time_graph("one store per thread, 256 x 256", n, [&](int) { hipLaunchKernelGGL(k_store, dim3(256), dim3(256), 0, st, p); });
  time_graph("one store per thread, 1024 x 256", n, [&](int) { hipLaunchKernelGGL(k_store, dim3(1024), dim3(256), 0, st, p); });
  time_graph("8 distinct kernels alternating, 256 x 256", n, [&](int i) {
    switch (i & 7) {
      case 0: hipLaunchKernelGGL(k_multi<0>, dim3(256), dim3(256), 0, st, p); break;
      case 1: hipLaunchKernelGGL(k_multi<1>, dim3(256), dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL(k_multi<2>, dim3(256), dim3(256), 0, st, p); break;
      case 3: hipLaunchKernelGGL(k_multi<3>, dim3(256), dim3(256), 0, st, p); break;
      case 4: hipLaunchKernelGGL(k_multi<4>, dim3(256), dim3(256), 0, st, p); break;
      case 5: hipLaunchKernelGGL(k_multi<5>, dim3(256), dim3(256), 0, st, p); break;
      case 6: hipLaunchKernelGGL(k_multi<6>, dim3(256), dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL(k_multi<7>, dim3(256), dim3(256), 0, st, p); break;
    }
  });
  time_graph("LDS 68 KB + 200 VGPRs mix (lds, vgpr alternating)", n, [&](int i) {
    if (i & 1) hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 68 * 1024, st, p);
    else hipLaunchKernelGGL(k_vgpr, dim3(256), dim3(256), 0, st, p);
  });
  return 0;
}
