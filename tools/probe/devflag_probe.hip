// A cross-stream dependency WITHOUT an event on the producing stream: what it costs and whether the consumer sees the data.
//
//   stream 0:  A (WG workgroups write a tick-dependent pattern)  ->  C (the producing stream goes on)
//   stream 1:  wait  ->  B (reads everything A wrote, counts words that are not this round's)
//
// Variants of "wait":
//   0  none: B does not depend on A (lower bound of the loop)
//   1  hipEventRecord (no system fence) + hipStreamWaitEvent                      -- what the library does at a fork
//   2  A carries the event itself (hipExtLaunchKernelGGL stop event)             -- what the armed launches do
//   3  DEVICE FLAG: A's workgroups arrive on a per-XCD counter (workgroup i runs on XCD i % 8); the LAST workgroup of an XCD releases at agent
//      scope (one L2 write-back per XCD instead of one per workgroup) and arrives on a global counter; the eighth arrival stores the flag.
//      Stream 1 runs a one-wave kernel that polls the flag (bounded: it gives up after ~20 ms and raises an error word) in front of B.
//   4  as 3, but EVERY workgroup releases before it arrives (the textbook form)
// Reported: us per (A, wait, C) round on the host clock over N rounds, stale words seen by B, poll timeouts.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/devflag_probe.hip -o tools/probe/bin/devflag_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int WORDS_PER_WG = 4096;   // 16 KB per workgroup

struct Sync { unsigned xcd[8]; unsigned all; unsigned flag; unsigned err; unsigned stale; };

// mode 0: plain; 3: per-XCD last arriver releases; 4: every workgroup releases
__global__ void produce(unsigned* buf, unsigned tick, int iters, Sync* s, int mode, unsigned n_wg) {
  unsigned v = tick * 2654435761u + blockIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;     // ~ the work
  for (int i = threadIdx.x; i < WORDS_PER_WG; i += blockDim.x) buf[(size_t)blockIdx.x * WORDS_PER_WG + i] = tick + (v & 0u);
  if (mode == 0) return;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const unsigned x = blockIdx.x & 7u, mine = (n_wg - x + 7u) / 8u;    // workgroups dispatched to this XCD
  if (mode == 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  const unsigned prev = __hip_atomic_fetch_add(&s->xcd[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((prev + 1u) % mine != 0u) return;   // (the counters run on from round to round)
  // last workgroup of this XCD for this round: one write-back of the XCD's L2, then the global arrival
  if (mode == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  const unsigned a = __hip_atomic_fetch_add(&s->all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((a + 1u) % 8u == 0u) __hip_atomic_store(&s->flag, tick, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void poll(Sync* s, unsigned tick) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(&s->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < tick) {
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - t0 > 2000000LL) { atomicAdd(&s->err, 1u); return; }   // 100 MHz: 20 ms
  }
}

__global__ void consume(const unsigned* buf, unsigned tick, size_t n, Sync* s) {
  unsigned bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad += buf[i] != tick;
  if (bad) atomicAdd(&s->stale, bad);
}

__global__ void busy(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

int main(int argc, char** argv) {
  // argv[1] = "free": no back edge from B to the next A (B then reads data the next round may be overwriting: its stale count means nothing)
  // -- the round time is then what the PRODUCING stream needs for (A, its share of the wait, C)
  const bool back = !(argc > 1 && argv[1][0] == 'f');
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const int WG = 668, N = 200, IT = 3000;
  unsigned* buf; CK(hipMalloc(&buf, (size_t)WG * WORDS_PER_WG * 4));
  float* scratch; CK(hipMalloc(&scratch, 1 << 24));
  Sync* sy; CK(hipMalloc(&sy, sizeof(Sync)));
  hipEvent_t ev[2], evb[2];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  for (auto& e : evb) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  for (int variant = 0; variant < 5; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(sy, 0, sizeof(Sync)));
      CK(hipMemset(buf, 0, (size_t)WG * WORDS_PER_WG * 4));
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int i = 0; i < N; ++i) {
        const unsigned tick = (unsigned)(i + 1);
        const int mode = variant >= 3 ? variant : 0;
        if (variant == 2) hipExtLaunchKernelGGL(produce, dim3(WG), dim3(256), 0, s0, nullptr, ev[i & 1], 0, buf, tick, IT, sy, mode, (unsigned)WG);
        else hipLaunchKernelGGL(produce, dim3(WG), dim3(256), 0, s0, buf, tick, IT, sy, mode, (unsigned)WG);
        if (variant == 1) { CK(hipEventRecord(ev[i & 1], s0)); CK(hipStreamWaitEvent(s1, ev[i & 1], 0)); }
        if (variant == 2) CK(hipStreamWaitEvent(s1, ev[i & 1], 0));
        if (variant >= 3) hipLaunchKernelGGL(poll, dim3(1), dim3(64), 0, s1, sy, tick);
        hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, s0, scratch, 2000);                                   // C: the producing stream goes on
        if (variant) hipLaunchKernelGGL(consume, dim3(256), dim3(256), 0, s1, buf, tick, (size_t)WG * WORDS_PER_WG, sy);   // B
        if (variant && back) {   // the NEXT round's A overwrites buf: it must not start before this round's B is done (as the library's join does)
          CK(hipEventRecord(evb[i & 1], s1));
          CK(hipStreamWaitEvent(s0, evb[i & 1], 0));
        }
      }
      CK(hipDeviceSynchronize());
      auto t1 = std::chrono::high_resolution_clock::now();
      Sync h;
      CK(hipMemcpy(&h, sy, sizeof(Sync), hipMemcpyDeviceToHost));
      printf("%s variant %d rep %d: %.2f us per round, stale words %u, poll timeouts %u\n", back ? "chained" : "free   ", variant, rep,
             std::chrono::duration<double, std::micro>(t1 - t0).count() / N, h.stale, h.err);
    }
  }
  return 0;
}
