// issue interval / dependent latency of the fp32 MFMA forms on gfx950: cycles per instruction of ONE wave with NACC independent
// accumulator chains (s_memtime around 4096 instructions).  build: hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef float fx16 __attribute__((ext_vector_type(16)));
template <int FORM, int NACC>
__global__ void k(float* out, unsigned long long* cyc, float a, float b) {
  constexpr int N = 4096;
  if constexpr (FORM == 0) {
    fx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = fx4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N / NACC; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
  } else if constexpr (FORM == 1) {
    fx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = fx4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N / NACC; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
  } else {
    fx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N / NACC; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
  }
}
template <int FORM, int NACC>
void run(const char* name, float* out, unsigned long long* cyc, int waves) {
  unsigned long long h = 0;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<FORM, NACC>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f, 0.5f);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-10s %2d chains, %d wave(s) per workgroup: %6.1f cycles per MFMA (wave 0)\n", name, NACC, waves, (double)h / 4096.0);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  for (int waves : {1, 4, 8}) {
    run<0, 1>("4x4x1", out, cyc, waves); run<0, 2>("4x4x1", out, cyc, waves); run<0, 4>("4x4x1", out, cyc, waves); run<0, 8>("4x4x1", out, cyc, waves); run<0, 16>("4x4x1", out, cyc, waves);
    run<1, 1>("16x16x4", out, cyc, waves); run<1, 2>("16x16x4", out, cyc, waves); run<1, 4>("16x16x4", out, cyc, waves); run<1, 8>("16x16x4", out, cyc, waves);
    run<2, 1>("32x32x2", out, cyc, waves); run<2, 2>("32x32x2", out, cyc, waves); run<2, 4>("32x32x2", out, cyc, waves);
  }
  return 0;
}
