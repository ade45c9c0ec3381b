// What a cross-stream dependency costs the PRODUCING stream: N x (kernel A, <sync op>, kernel B) on stream 0, the consumer on stream 1.
// Variants: 0 none (no dependency), 1 hipEventRecord + hipStreamWaitEvent, 2 hipExtLaunchKernelGGL stop event + hipStreamWaitEvent,
// 3 hipStreamWriteValue32 + hipStreamWaitValue32 (signal memory), 4 event with hipEventDisableSystemFence.
// build: hipcc --offload-arch=gfx950 -O2 sync_gap_probe.hip -o /tmp/sync_gap_probe ; run: /tmp/sync_gap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  float* buf; CK(hipMalloc(&buf, 1 << 24));
  uint32_t* flag; CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  const int N = 200, WG = 512, IT = 2000;   // ~10 us kernels
  hipEvent_t ev[2], evn[2];
  for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evn[i], hipEventDisableTiming | hipEventDisableSystemFence)); }
  for (int variant = 0; variant < 5; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int i = 0; i < N; ++i) {
        const uint32_t tick = (uint32_t)(variant * 100000 + rep * 1000 + i + 1);
        if (variant == 2) hipExtLaunchKernelGGL(spin, dim3(WG), dim3(256), 0, s0, nullptr, ev[i & 1], 0, buf, IT);
        else hipLaunchKernelGGL(spin, dim3(WG), dim3(256), 0, s0, buf, IT);
        if (variant == 1) { CK(hipEventRecord(ev[i & 1], s0)); CK(hipStreamWaitEvent(s1, ev[i & 1], 0)); }
        if (variant == 2) { CK(hipStreamWaitEvent(s1, ev[i & 1], 0)); }
        if (variant == 3) { CK(hipStreamWriteValue32(s0, flag, tick, 0)); CK(hipStreamWaitValue32(s1, flag, tick, hipStreamWaitValueGte, 0xffffffffu)); }
        if (variant == 4) { CK(hipEventRecord(evn[i & 1], s0)); CK(hipStreamWaitEvent(s1, evn[i & 1], 0)); }
        hipLaunchKernelGGL(spin, dim3(WG), dim3(256), 0, s0, buf + (1 << 20), IT);          // the producer stream goes on
        if (variant) hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, s1, buf + (2 << 20), 100);   // the consumer
      }
      CK(hipDeviceSynchronize());
      auto t1 = std::chrono::high_resolution_clock::now();
      printf("variant %d rep %d: %.2f us per (A, sync, B) pair\n", variant, rep, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
  }
  return 0;
}
