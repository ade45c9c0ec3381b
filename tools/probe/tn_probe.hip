// Ablation probe for the no-LDS weight-gradient GEMM (tuning aid, not part of the library): MODE 0 = the kernel as shipped,
// 1 = every load reads the same few lines (memory system out of the picture), 2 = loads only (no MFMA), 3 = MFMA + address VALU, no loads.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/tn_probe.hip -o tools/probe/tn_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float fx2 __attribute__((ext_vector_type(2)));
constexpr int TB = 128;
template <int MODE, int DEPTH, int WPB>
__global__ __launch_bounds__(64 * WPB) void k(const float* __restrict__ P, int ldp, const float* __restrict__ Q, int ldq, int T, int R, int Cc,
                                       int tok_per_split, int n_splits, float* __restrict__ part, const float* __restrict__ tn_zero) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) & 1, wc = wave & 1, li = lane & 31, kk = lane >> 5;
  const int ksub = wave >> 2;                       // WPB = 8: two token-interleaved halves
  const int ntc = (Cc + TB - 1) / TB, ntiles = ntc * ((R + TB - 1) / TB);
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  const int sp = (qid / ntiles) * 8 + xcd, tile = qid % ntiles;
  if (sp >= n_splits) return;
  const int c0 = (tile % ntc) * TB, r0 = (tile / ntc) * TB;
  const int t_begin = sp * tok_per_split + ksub * (tok_per_split / (WPB / 4));
  const int t_end = min(T, t_begin + tok_per_split / (WPB / 4));
  const int ra = r0 + wr * 64 + 2 * li, cb = c0 + wc * 64 + 2 * li;
  const bool rin = ra < R, cin = cb < Cc;
  const float* Pp = P + (rin ? ra : 0);
  const float* Qp = Q + (cin ? cb : 0);
  floatx16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  fx2 bsum = {0.f, 0.f};
  const int n_it = (max(t_end - t_begin, 0) + 1) >> 1;
  fx2 fa[DEPTH], fb[DEPTH], ga[DEPTH], gb[DEPTH];
  auto issue = [&](int it, fx2& a, fx2& b) {
    const int t = t_begin + 2 * it + kk;
    const bool in = rin && t < t_end;
    if (MODE == 3) { a = fx2{(float)t, 1.f}; b = fx2{(float)it, 2.f}; return; }
    const float* pa = in ? Pp + (long long)(MODE == 1 ? (t & 7) : t) * ldp : tn_zero;
    a = *(const fx2*)pa;
    b = *(const fx2*)(Qp + (long long)(MODE == 1 ? (t & 7) : min(t, T - 1)) * ldq);
  };
  auto consume = [&](int it, fx2 a, fx2 b) {
    if (MODE != 2) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
    } else { acc[0][0][0] += a.x * b.x; acc[0][0][1] += a.y * b.y; }
    bsum.x += a.x; bsum.y += a.y;
  };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) issue(u, fa[u], fb[u]);
  for (int base = 0; base < n_it; base += 2 * DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) { issue(base + DEPTH + u, ga[u], gb[u]); __builtin_amdgcn_sched_barrier(0); consume(base + u, fa[u], fb[u]); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) { issue(base + 2 * DEPTH + u, fa[u], fb[u]); __builtin_amdgcn_sched_barrier(0); consume(base + DEPTH + u, ga[u], gb[u]); __builtin_amdgcn_sched_barrier(0); }
  }
  float* out = part + ((long long)sp * (WPB / 4) + ksub) * R * Cc;
  if (cin) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = r0 + wr * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kk) + i;
        if (rr < R) *(fx2*)(out + (long long)rr * Cc + cb) = fx2{acc[i][0][r], acc[i][1][r] + bsum.x + bsum.y};
      }
  }
}
template <int MODE, int DEPTH, int WPB> void run(int T, int R, int Cc, int target) {
  float *P, *Q, *part, *z;
  hipMalloc(&P, (size_t)T * R * 4); hipMalloc(&Q, (size_t)T * Cc * 4); hipMalloc(&z, 256); hipMemset(z, 0, 256);
  hipMemset(P, 0, (size_t)T * R * 4); hipMemset(Q, 0, (size_t)T * Cc * 4);
  const int tiles = ((R + 127) / 128) * ((Cc + 127) / 128);
  int S = (target + tiles - 1) / tiles;
  int tps = (T + S - 1) / S; tps = (tps + 31) / 32 * 32;
  hipMalloc(&part, (size_t)S * (WPB / 4) * R * Cc * 4);
  dim3 grid(8 * ((S + 7) / 8) * tiles);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, DEPTH, WPB>), grid, dim3(64 * WPB), 0, 0, P, R, Q, Cc, T, R, Cc, tps, S, part, z);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("MODE %d DEPTH %d WPB %d target %d: T=%d R=%d C=%d S=%d tps=%d grid=%d  best %.1f us  %.1f TF/s\n", MODE, DEPTH, WPB, target, T, R, Cc, S, tps, grid.x, best * 1e3,
         2.0 * T * R * Cc / best / 1e9);
  hipFree(P); hipFree(Q); hipFree(part); hipFree(z);
}
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 21360;
  for (int target : {256, 512}) {
    run<0, 8, 4>(T, 512, 128, target); run<1, 8, 4>(T, 512, 128, target); run<2, 8, 4>(T, 512, 128, target); run<3, 8, 4>(T, 512, 128, target);
    run<0, 4, 4>(T, 512, 128, target); run<0, 8, 8>(T, 512, 128, target); run<1, 8, 8>(T, 512, 128, target);
  }
  run<0, 8, 4>(204800, 512, 128, 256); run<1, 8, 4>(204800, 512, 128, 256); run<3, 8, 4>(204800, 512, 128, 256);
  return 0;
}
