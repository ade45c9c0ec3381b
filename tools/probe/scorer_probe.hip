// Round 5 probe: where does the fused gather-dot (scorer_loss_fwd_kernel<32,8,1024>: 0.70-0.72 of the HBM peak) lose the
// 10-13 % it is short of the random-row ceiling (randrow_probe: 0.82)?  The same 4 100 096 random 512-byte rows of a 51.2 GB table, read by
//   V0  the ceiling kernel: grid-stride lane groups, rows folded into a register sum (randrow_probe's rr_kernel<8, nt>)
//   V1  the scorer's SHAPE: one 1024-thread workgroup per batch row b, its G = 1001 ids walked in trips of 32 groups x 8 rows, ids by
//       vector loads at the head of every trip -- still no dot product, no stores but one per thread
//   V2  V1 + the dot with the row's user vector, the 32-lane sum, the score store and the LDS copy of the scores (the real kernel's work)
//   V3  V2 with the ids of the NEXT trip fetched by scalar loads into SGPRs while this trip's rows are in flight (VERDICT r4's suggestion:
//       a one-trip-ahead id stream at zero VGPR cost)
//   V4  one row per WAVE: 64 lanes x 8 bytes, the id wave-uniform (scalar load, row base in SGPRs), 8 rows in flight per wave
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/scorer_probe.hip -o tools/probe/bin/scorer_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef float fx2 __attribute__((ext_vector_type(2)));
constexpr int U = 8;

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float sum32(float v) {
  v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
  v += __shfl_xor(v, 16, 64);
  return v;
}
__device__ __forceinline__ float sum64(float v) { v = sum32(v); v += __shfl_xor(v, 32, 64); return v; }

__global__ __launch_bounds__(256) void v0(const fx4* __restrict__ table, const long long* __restrict__ ids, long long n, float* __restrict__ out) {
  const int t = threadIdx.x & 31;
  const long long groups = (long long)gridDim.x * (blockDim.x >> 5), g0 = (long long)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
  fx4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long i = g0 * U; i < n; i += groups * U) {
    long long id[U];
#pragma unroll
    for (int q = 0; q < U; ++q) id[q] = ids[i + q < n ? i + q : n - 1];
    fx4 e[U];
#pragma unroll
    for (int q = 0; q < U; ++q) e[q] = __builtin_nontemporal_load(&table[id[q] * 32 + t]);
#pragma unroll
    for (int q = 0; q < U; ++q) acc += e[q];
  }
  out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// MODE 1: shape only; 2: + dot, group sum, score store, LDS copy; 3: 2 + scalar next-trip ids; 4: 2 without the global score store;
// 5: 2 with a TRANSPOSED reduction (8 rows x 32 lanes -> one row total per lane in 9 exchanges instead of 40) and one 32-byte store per group
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void v123(const fx4* __restrict__ table, const long long* __restrict__ ids, int G, const fx4* __restrict__ user,
                                             float* __restrict__ scores, float* __restrict__ out) {
  extern __shared__ float sc[];
  const int b = blockIdx.x, t = threadIdx.x & 31, g0 = threadIdx.x >> 5, groups = NT / 32;
  const long long* my = ids + (long long)b * G;
  const fx4 u = user[(long long)b * 32 + t];
  fx4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 3) {
    // the wave's two lane groups walk 16 consecutive ids per trip: a wave-uniform base -> scalar loads
    const int w16 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 16;
    long long nid[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) nid[q] = my[min(w16 + q, G - 1)];
    for (int gb0 = 0; gb0 < G; gb0 += groups * U) {
      long long cur[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) cur[q] = nid[q];
      const int nb = gb0 + groups * U + w16;
#pragma unroll
      for (int q = 0; q < 16; ++q) nid[q] = my[min(nb + q, G - 1)];      // next trip's ids: in flight under this trip's rows
      const bool hi = (threadIdx.x & 32) != 0;
      fx4 e[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const long long id = hi ? cur[8 + q] : cur[q];
        e[q] = __builtin_nontemporal_load(&table[id * 32 + t]);
      }
      const int gb = gb0 + w16 + (hi ? 8 : 0);
#pragma unroll
      for (int q = 0; q < U; ++q) {
        float s = (e[q].x * u.x + e[q].y * u.y) + (e[q].z * u.z + e[q].w * u.w);
        s = sum32(s);
        if (t == 0 && gb + q < G) { sc[gb + q] = s; scores[(long long)b * G + gb + q] = s; }
      }
    }
  } else {
    for (int gb = g0 * U; gb < G; gb += groups * U) {
      long long id[U];
#pragma unroll
      for (int q = 0; q < U; ++q) id[q] = (gb + q < G) ? my[gb + q] : 0;
      fx4 e[U];
#pragma unroll
      for (int q = 0; q < U; ++q) e[q] = __builtin_nontemporal_load(&table[id[q] * 32 + t]);
      if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < U; ++q) acc += e[q];
      } else if (MODE == 5) {
        float v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = (e[q].x * u.x + e[q].y * u.y) + (e[q].z * u.z + e[q].w * u.w);
        const bool b0 = t & 1, b1 = t & 2, b2 = t & 4;
        float r4[4], r2[2], r1;
#pragma unroll
        for (int i = 0; i < 4; ++i) r4[i] = (b0 ? v[4 + i] : v[i]) + dpp<0xB1>(b0 ? v[i] : v[4 + i]);          // lanes l, l ^ 1
#pragma unroll
        for (int i = 0; i < 2; ++i) r2[i] = (b1 ? r4[2 + i] : r4[i]) + dpp<0x4E>(b1 ? r4[i] : r4[2 + i]);      // l ^ 2
        r1 = (b2 ? r2[1] : r2[0]) + __shfl_xor(b2 ? r2[0] : r2[1], 4, 64);                                     // l ^ 4
        r1 += __shfl_xor(r1, 8, 64);
        r1 += __shfl_xor(r1, 16, 64);
        const int row = 4 * (t & 1) + 2 * ((t >> 1) & 1) + ((t >> 2) & 1);     // the row whose total this lane holds
        if (t < 8 && gb + row < G) { sc[gb + row] = r1; scores[(long long)b * G + gb + row] = r1; }
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q) {
          float s = (e[q].x * u.x + e[q].y * u.y) + (e[q].z * u.z + e[q].w * u.w);
          s = sum32(s);
          if (t == 0 && gb + q < G) { sc[gb + q] = s; if (MODE != 4) scores[(long long)b * G + gb + q] = s; }
        }
      }
    }
  }
  if (MODE == 1) out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + u.x;
  __syncthreads();
  if (MODE != 1 && threadIdx.x == 0) out[b] = sc[0];
}

// one row per wave: 64 lanes x 8 bytes; ids wave-uniform
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void v4(const fx2* __restrict__ table, const long long* __restrict__ ids, int G, const fx2* __restrict__ user,
                                                 float* __restrict__ scores, float* __restrict__ out) {
  extern __shared__ float sc[];
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long* my = ids + (long long)b * G;
  const fx2 u = user[(long long)b * 64 + lane];
  long long nid[U];
#pragma unroll
  for (int q = 0; q < U; ++q) nid[q] = my[min(w * U + q, G - 1)];
  for (int gb = w * U; gb < G; gb += WAVES * U) {
    long long cur[U];
#pragma unroll
    for (int q = 0; q < U; ++q) cur[q] = nid[q];
#pragma unroll
    for (int q = 0; q < U; ++q) nid[q] = my[min(gb + WAVES * U + q, G - 1)];
    fx2 e[U];
#pragma unroll
    for (int q = 0; q < U; ++q) e[q] = __builtin_nontemporal_load(&table[cur[q] * 64 + lane]);
#pragma unroll
    for (int q = 0; q < U; ++q) {
      float s = e[q].x * u.x + e[q].y * u.y;
      s = sum64(s);
      if (lane == 0 && gb + q < G) { sc[gb + q] = s; scores[(long long)b * G + gb + q] = s; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[b] = sc[0];
}

__global__ void fill_random(unsigned* p, long long n) {   // (zero-filled tables read faster than real data on this part: DVFS, MI355X_MICROARCH.md)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = 0x3C000000u | (x & 0x03FFFFFFu);   // floats in [2^-7, 2^-3): finite, every bit of the mantissa in use
  }
}

template <typename F> static float timeit(F launch, int reps = 8) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const bool random_data = argc > 1;
  const int B = 4096, G = 1001;
  const long long n = (long long)B * G, rows = 100000000LL;
  fx4* table; CK(hipMalloc(&table, rows * 512)); CK(hipMemset(table, 0, rows * 512));
  if (random_data) { hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, (unsigned*)table, rows * 128); CK(hipDeviceSynchronize()); }
  printf("table contents: %s\n", random_data ? "random floats" : "zeros");
  std::vector<long long> h(n);
  unsigned long long s = 88172645463325252ULL;
  for (long long i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (long long)(s % (unsigned long long)(rows - 1)) + 1; }
  long long* ids; CK(hipMalloc(&ids, n * 8)); CK(hipMemcpy(ids, h.data(), n * 8, hipMemcpyHostToDevice));
  fx4* user; CK(hipMalloc(&user, (size_t)B * 512)); CK(hipMemset(user, 0, (size_t)B * 512));
  float *scores, *out; CK(hipMalloc(&scores, n * 4)); CK(hipMalloc(&out, 256LL * 16384 * 4));
  const double bytes = (double)n * (512 + 8);
  auto rep = [&](const char* tag, float ms) { printf("%-58s %7.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", tag, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0); };
  rep("V0 ceiling: grid-stride groups, U=8, nt (2048 wgs)", timeit([&] { hipLaunchKernelGGL(v0, dim3(2048), dim3(256), 0, 0, table, ids, n, out); }));
  const size_t lds = (G + 16) * 4;
  rep("V1 scorer shape (1024-thread wg per b), no dot", timeit([&] { hipLaunchKernelGGL((v123<1, 1024>), dim3(B), dim3(1024), lds, 0, table, ids, G, user, scores, out); }));
  rep("V2 + dot, 32-lane sum, score store, LDS copy", timeit([&] { hipLaunchKernelGGL((v123<2, 1024>), dim3(B), dim3(1024), lds, 0, table, ids, G, user, scores, out); }));
  rep("V3 + next trip's ids by scalar loads (SGPRs)", timeit([&] { hipLaunchKernelGGL((v123<3, 1024>), dim3(B), dim3(1024), lds, 0, table, ids, G, user, scores, out); }));
  rep("V2 at 256 threads per wg (the library's launch for B > 512)", timeit([&] { hipLaunchKernelGGL((v123<2, 256>), dim3(B), dim3(256), lds, 0, table, ids, G, user, scores, out); }));
  rep("V2 without the global score store (LDS copy only)", timeit([&] { hipLaunchKernelGGL((v123<4, 1024>), dim3(B), dim3(1024), lds, 0, table, ids, G, user, scores, out); }));
  rep("V5 transposed reduction, one 32-byte score store per group", timeit([&] { hipLaunchKernelGGL((v123<5, 1024>), dim3(B), dim3(1024), lds, 0, table, ids, G, user, scores, out); }));
  rep("V5 at 256 threads per wg", timeit([&] { hipLaunchKernelGGL((v123<5, 256>), dim3(B), dim3(256), lds, 0, table, ids, G, user, scores, out); }));
  rep("V4 row per wave (64 x 8 B), scalar ids, 16 waves per wg", timeit([&] { hipLaunchKernelGGL(v4<16>, dim3(B), dim3(1024), lds, 0, (const fx2*)table, ids, G, (const fx2*)user, scores, out); }));
  rep("V4 row per wave, 8 waves per wg", timeit([&] { hipLaunchKernelGGL(v4<8>, dim3(B), dim3(512), lds, 0, (const fx2*)table, ids, G, (const fx2*)user, scores, out); }));
  rep("V4 row per wave, 4 waves per wg", timeit([&] { hipLaunchKernelGGL(v4<4>, dim3(B), dim3(256), lds, 0, (const fx2*)table, ids, G, (const fx2*)user, scores, out); }));
  return 0;
}
