// Sustained fp32 MFMA rate of the chip (tuning aid): every wave issues NI v_mfma_f32_32x32x2_f32 on ACC independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak [waves_per_simd] [acc]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int ACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  floatx16 acc[ACC];
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < ACC; ++i) s += acc[i][0];
  if (s == 12345.f) out[0] = s;
}
template <int ACC> void run(int wps, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * wps;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<ACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * ACC * 4096.0;
    printf("waves/SIMD=%d acc=%d: %.3f ms  %.1f TFLOP/s\n", wps, ACC, ms, flops / ms / 1e9);
  }
}
int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1, acc = argc > 2 ? atoi(argv[2]) : 2;
  const int iters = 20000 / wps;
  if (acc == 1) run<1>(wps, iters); else if (acc == 2) run<2>(wps, iters); else run<4>(wps, iters);
  return 0;
}
