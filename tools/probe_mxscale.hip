// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950): which lane's scale byte governs which operand bytes.
// A = e4m3 ones in ONE (lane, VGPR) cell, B = all ones; every scale byte 127 (2^0) except ONE lane of scale_a = 128 (2^1).
// D[row][0] = 4 * (1 or 2) tells whether that scale lane governs that cell.  Prints a table: cell (lane, vgpr) -> scale lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void probe(const int* acell, const int* sc, float* out) {
  // acell[0] = lane, acell[1] = vgpr holding ones; sc[0] = lane whose scale_a is 2
  const int lane = threadIdx.x;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
  for (int i = 0; i < 8; ++i) b[i] = 0x38383838;
  if (lane == acell[0]) a[acell[1]] = 0x38383838;
  const int sa = lane == sc[0] ? 128 : 127, sb = 127;
  v16f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}
int main() {
  int *d_cell, *d_sc;
  float* d_out;
  hipMalloc(&d_cell, 8); hipMalloc(&d_sc, 4); hipMalloc(&d_out, 64 * 16 * 4);
  static float h[64 * 16];
  const int lanes[4] = {0, 5, 32, 37};
  for (int li = 0; li < 4; ++li)
    for (int v = 0; v < 8; ++v) {
      int cell[2] = {lanes[li], v};
      hipMemcpy(d_cell, cell, 8, hipMemcpyHostToDevice);
      int gov = -1, cnt = 0;
      for (int s = 0; s < 64; ++s) {
        hipMemcpy(d_sc, &s, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_cell, d_sc, d_out);
        hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        float mx = 0;
        for (int i = 0; i < 64 * 16; ++i) mx = h[i] > mx ? h[i] : mx;
        if (mx > 4.5f) { gov = s; ++cnt; }
      }
      // which output row lights up (column 0 = lanes 0 and 32): D row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
      hipMemcpy(d_sc, &gov, 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_cell, d_sc, d_out);
      hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
      int row = -1;
      for (int l = 0; l < 64; l += 32)
        for (int r = 0; r < 16; ++r)
          if (h[l * 16 + r] != 0.f) row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      printf("A cell lane %2d vgpr %d -> D row %2d ; governed by scale_a lane %2d (%d lanes matched)\n", lanes[li], v, row, gov, cnt);
    }
  return 0;
}
