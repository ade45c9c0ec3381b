// Round 5 probe: where does the 128 x 128 LDS-DMA weight-gradient kernel lose its time?  One workgroup streams `tok` token rows of a
// [T, 128] P tile and a [T, ldq] Q tile through a 3-stage LDS ring exactly as gemm_tn_big_kernel does (global_load_lds_dwordx4, 8 pieces
// per thread and stage, counted vmcnt + raw s_barrier) and runs the same 64 MFMAs per stage.
//   MODE 0 = both, 1 = DMA only (no MFMA, no fragment reads), 2 = MFMA + fragment reads only (no DMA), 3 = register staging
//   (global_load_dwordx4 -> ds_write_b128) + MFMA
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/ldsdma_probe.hip -o tools/probe/bin/ldsdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int BT = 128, BTK = 32, BTB = 3;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ P, const float* __restrict__ Q, int ldp, int ldq, int tok, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, hi = lane >> 5, fcol = lane & 31;
  const long long t_begin = (long long)blockIdx.x * tok;
  const float* Pg = P + fcol * 4;
  const float* Qg = Q + fcol * 4;
  const int nt = tok / BTK;
  auto issue = [&](int buf, long long t0) {
    float* Pl = smem + buf * (2 * BTK * BT);
    float* Ql = Pl + BTK * BT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave + 4 * i;
      const long long t = t0 + 2 * piece + hi;
      if (MODE == 3) {
        const float4 a = *(const float4*)(Pg + t * ldp), b = *(const float4*)(Qg + t * ldq);
        *(float4*)(Pl + piece * 2 * BT + lane * 4) = a;
        *(float4*)(Ql + piece * 2 * BT + lane * 4) = b;
      } else {
        __builtin_amdgcn_global_load_lds((gptr_t)(Pg + t * ldp), (lptr_t)(Pl + piece * 2 * BT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(Qg + t * ldq), (lptr_t)(Ql + piece * 2 * BT), 16, 0, 0);
      }
    }
  };
  floatx16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (MODE != 2) {
    issue(0, t_begin);
    if (nt > 1) issue(1, t_begin + BTK);
    if (MODE == 3) __syncthreads();
    else { if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  const int pa = hi * BT + wr * 64 + fcol, qa = BTK * BT + hi * BT + wc * 64 + fcol;
  int buf = 0;
  for (int s = 0; s < nt; ++s) {
    const bool more = s + 2 < nt;
    if (MODE != 2 && more) issue(buf >= 1 ? buf - 1 : BTB - 1, t_begin + (long long)(s + 2) * BTK);
    if (MODE != 1) {
      const float* Pb = smem + buf * (2 * BTK * BT) + pa;
      const float* Qb = smem + buf * (2 * BTK * BT) + qa;
      float a0 = Pb[0], a1 = Pb[32], b0 = Qb[0], b1 = Qb[32];
#pragma unroll
      for (int kk = 0; kk < BTK / 2; ++kk) {
        float a0n = 0.f, a1n = 0.f, b0n = 0.f, b1n = 0.f;
        if (kk + 1 < BTK / 2) { a0n = Pb[(kk + 1) * 2 * BT]; a1n = Pb[(kk + 1) * 2 * BT + 32]; b0n = Qb[(kk + 1) * 2 * BT]; b1n = Qb[(kk + 1) * 2 * BT + 32]; }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
      }
    }
    if (MODE == 3) __syncthreads();
    else if (MODE != 2) { if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    buf = buf + 1 < BTB ? buf + 1 : 0;
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (MODE == 1) s = smem[tid];
  out[(long long)blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(int nwg, int tok, int ldq, int lds_kb) {
  const long long T = (long long)nwg * tok;
  float *P, *Q, *out;
  hipMalloc(&P, (size_t)T * 128 * 4); hipMalloc(&Q, (size_t)T * ldq * 4); hipMalloc(&out, (size_t)nwg * 256 * 4);
  hipMemset(P, 0, (size_t)T * 128 * 4); hipMemset(Q, 0, (size_t)T * ldq * 4);
  const int lds = lds_kb * 1024;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), lds, 0, P, Q, 128, ldq, tok, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double bytes = (double)T * 256 * 4, flop = 2.0 * T * 128 * 128;
  printf("MODE %d  wgs %4d  tok/wg %5d  ldq %4d  lds %3d KB: %8.1f us  %6.2f us/stage  %6.2f TB/s  %6.1f TF/s\n", MODE, nwg, tok, ldq, lds_kb, best * 1e3,
         best * 1e3 / (tok / BTK), MODE == 2 ? 0.0 : bytes / best / 1e9, MODE == 1 ? 0.0 : flop / best / 1e9);
  hipFree(P); hipFree(Q); hipFree(out);
}
int main() {
  for (int nwg : {4, 64, 128, 256, 512}) {
    run<0>(nwg, 1024, 128, 96); run<1>(nwg, 1024, 128, 96); run<2>(nwg, 1024, 128, 96); run<3>(nwg, 1024, 128, 96);
  }
  run<0>(256, 1024, 512, 96); run<1>(256, 1024, 512, 96); run<3>(256, 1024, 512, 96);
  run<0>(512, 1024, 128, 64); run<1>(512, 1024, 128, 64); run<3>(512, 1024, 128, 64);   // two workgroups per CU
  run<1>(1024, 1024, 128, 48); run<1>(2048, 512, 128, 48);                               // three per CU: DMA supply with more streams
  return 0;
}
