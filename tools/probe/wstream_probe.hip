// How fast can ONE workgroup pull an L2-resident weight set (640 KB, shared by every workgroup) into registers, by access pattern?
// Decides the B-operand mapping of the fused last-row kernels (lastrow.hip): a lane that owns an output COLUMN of W [N][K] reads 16 B of
// its own row per instruction (64 different 128-B lines per wave instruction); a lane that owns VW consecutive columns of W^T [K][N]
// reads a fully coalesced 1 KB per wave instruction.
//   mode 0: lane = row, float4 along k            (64 lines x 16 B per instruction)
//   mode 1: lane & 31 = row, lane >> 5 = k half   (32 lines x 32 B per instruction; rc_gemm's pattern)
//   mode 2: contiguous float4 (1 KB / instruction)   mode 3: contiguous float2   mode 4: contiguous dword
// build: hipcc --offload-arch=gfx950 -O3 wstream_probe.hip -o /tmp/wstream_probe ; run: /tmp/wstream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef float fx2 __attribute__((ext_vector_type(2)));
constexpr int NF = 160 * 1024;   // floats in the weight set (640 KB)
constexpr int K = 128;           // row length of the [N][K] view

template <int MODE>
__global__ __launch_bounds__(512) void stream_kernel(const float* __restrict__ W, float* __restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  fx4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE == 0) {
    // N = NF / K = 1280 rows; wave w takes row tiles w, w + 8, ...: 20 tiles of 64 rows, each K / 4 = 32 float4 per lane
    for (int t = wave; t < NF / K / 64; t += 8) {
      const float* p = W + (long long)(t * 64 + lane) * K;
#pragma unroll 8
      for (int s = 0; s < K / 4; ++s) acc += *(const fx4*)(p + 4 * s);
    }
  } else if constexpr (MODE == 1) {
    for (int t = wave; t < NF / K / 32; t += 8) {
      const float* p = W + (long long)(t * 32 + (lane & 31)) * K + 4 * (lane >> 5);
#pragma unroll 8
      for (int s = 0; s < K / 8; ++s) acc += *(const fx4*)(p + 8 * s);
    }
  } else if constexpr (MODE == 2) {
#pragma unroll 8
    for (int i = tid; i < NF / 4; i += 512) acc += *(const fx4*)(W + 4LL * i);
  } else if constexpr (MODE == 3) {
#pragma unroll 8
    for (int i = tid; i < NF / 2; i += 512) { const fx2 v = *(const fx2*)(W + 2LL * i); acc.x += v.x; acc.y += v.y; }
  } else if constexpr (MODE == 4) {
#pragma unroll 8
    for (int i = tid; i < NF; i += 512) acc.x += W[i];
  } else {
    // MODE 5 / 6: lastrow.hip's pattern -- buffer loads of 1 KB k-rows, row stride 2 KB (a [K][512] matrix, this wave's 256-column tile),
    // 32 rows requested back to back, then consumed; mode 6: 8 rows at a time
    constexpr int DEPTH = MODE == 5 ? 32 : 8;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    const int t1 = wave & 1, k1 = wave >> 1;            // 2 column tiles x 4 K-parts of 32 rows = one [128][512] matrix = 256 KB
    for (int m = 0; m < NF / (128 * 512); ++m) {        // (NF = 2.5 such matrices: two full ones)
      for (int r0 = 0; r0 < 32; r0 += DEPTH) {
        fx4 w[DEPTH];
#pragma unroll
        for (int r = 0; r < DEPTH; ++r)
          w[r] = __builtin_bit_cast(fx4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((m * 128 + k1 * 32 + r0 + r) * 512 + t1 * 256) * 4, 0));
#pragma unroll
        for (int r = 0; r < DEPTH; ++r) acc += w[r];
      }
    }
  }
  const float s = (acc.x + acc.y) + (acc.z + acc.w);
  if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static int run(const float* W, float* out, int G, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(G), dim3(512), 0, 0, W, out);
  const int N = 100;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(G), dim3(512), 0, 0, W, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / N;
  printf("mode %d (%-28s) G=%3d: %7.2f us / launch, %6.1f GB/s per workgroup, %6.2f TB/s chip\n", MODE, what, G, us,
         NF * 4.0 / us * 1e-3, (double)G * NF * 4.0 / us * 1e-6);
  return 0;
}

int main() {
  float *W, *out;
  CK(hipMalloc(&W, NF * sizeof(float)));
  CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  CK(hipMemset(W, 0, NF * sizeof(float)));
  for (int G : {32, 64, 128, 256}) {
    if (run<0>(W, out, G, "lane=row float4 (64 lines)")) return 1;
    if (run<1>(W, out, G, "32 rows x 32 B (rc_gemm)")) return 1;
    if (run<2>(W, out, G, "contiguous float4")) return 1;
    if (run<3>(W, out, G, "contiguous float2")) return 1;
    if (run<4>(W, out, G, "contiguous dword")) return 1;
    if (run<5>(W, out, G, "buffer 1 KB rows x32 (512 KB)")) return 1;
    if (run<6>(W, out, G, "buffer 1 KB rows x8 (512 KB)")) return 1;
  }
  return 0;
}
