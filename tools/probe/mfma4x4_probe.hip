// layout probe of v_mfma_f32_4x4x1_16b_f32 (tuning aid): which (block, row, col) each lane / register holds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  // A = 1000 * lane, B = 1 -> D[.] = A-lane value contributing; then A = 1, B = 1000 * lane
  floatx4 c = {0.f, 0.f, 0.f, 0.f};
  floatx4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.0f, c, 0, 0, 0);
  floatx4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)l, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { out[(l * 4 + r) * 2] = d1[r]; out[(l * 4 + r) * 2 + 1] = d2[r]; }
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 2 * sizeof(float));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) printf("  r%d A-lane %2.0f B-lane %2.0f", r, h[(l * 4 + r) * 2], h[(l * 4 + r) * 2 + 1]);
    printf("\n");
  }
  return 0;
}
