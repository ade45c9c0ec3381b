// What is the ceiling of RANDOM 512-byte row reads out of a table far larger than any cache or TLB reach on this part?  (VERDICT r3 item 8:
// the fused gather-dot reads 2.1 GB of random rows at 0.69-0.73 of the 8 TB/s HBM peak; is the rest the kernel's or the part's?)
// The kernel does the MINIMUM a row consumer can do: a lane group of 32 lanes reads one 512-B row (one 16-byte load per lane), U rows in
// flight per group, and folds it into a register sum (no dot product, no LDS, no stores but one per thread at the end).  Swept: rows in
// flight per group (U), threads per workgroup, workgroups per CU (grid), non-temporal vs plain loads, ids random or SORTED (sorted =
// the same bytes with page locality: what a TLB-friendly order would buy), table size (51.2 GB = C5's table; 3.2 GB: TLB reach matters less).
// build: hipcc --offload-arch=gfx950 -O3 randrow_probe.hip -o bin/randrow_probe ; run: bin/randrow_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float fx4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void rr_kernel(const fx4* __restrict__ table, const long long* __restrict__ ids, long long n,
                                                 float* __restrict__ out) {
  const int t = threadIdx.x & 31;
  const long long groups = (long long)gridDim.x * (blockDim.x >> 5), g0 = (long long)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
  fx4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long i = g0 * U; i < n; i += groups * U) {
    long long id[U];
#pragma unroll
    for (int q = 0; q < U; ++q) id[q] = ids[i + q < n ? i + q : n - 1];
    fx4 e[U];
#pragma unroll
    for (int q = 0; q < U; ++q) e[q] = NT ? __builtin_nontemporal_load(&table[id[q] * 32 + t]) : table[id[q] * 32 + t];
#pragma unroll
    for (int q = 0; q < U; ++q) acc += e[q];
  }
  out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int U, bool NT>
static float run(const fx4* table, const long long* ids, long long n, float* out, int grid, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rr_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, table, ids, n, out);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rr_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, table, ids, n, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  return best;
}

int main() {
  const long long n = 4100096;                       // lookups (the gather_roofline leg's: B = 512 x 8008 candidates)
  for (long long rows : {100000000LL, 6250000LL}) {  // 51.2 GB (C5) and 3.2 GB
    fx4* table;
    CK(hipMalloc(&table, rows * 512));
    CK(hipMemset(table, 0, rows * 512));
    std::vector<long long> h(n);
    unsigned long long s = 88172645463325252ULL;
    for (long long i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (long long)(s % (unsigned long long)(rows - 1)) + 1; }
    long long *ids, *ids_sorted;
    CK(hipMalloc(&ids, n * 8)); CK(hipMalloc(&ids_sorted, n * 8));
    CK(hipMemcpy(ids, h.data(), n * 8, hipMemcpyHostToDevice));
    std::sort(h.begin(), h.end());
    CK(hipMemcpy(ids_sorted, h.data(), n * 8, hipMemcpyHostToDevice));
    float* out;
    CK(hipMalloc(&out, 256LL * 16384 * 4));
    const double bytes = (double)n * (512 + 8);
    printf("table %.1f GB, %lld random 512-B rows per launch (%.2f GB), best of 8 launches\n", rows * 512 / 1e9, n, bytes / 1e9);
    for (int wg_per_cu : {2, 4, 8, 16}) {
      const int grid = 256 * wg_per_cu;
#define ROW(U, NT, IDS, tag) { const float ms = run<U, NT>(table, IDS, n, out, grid, 8); \
        printf("  %2d wg/CU  U=%2d  %s  ids %-6s  %7.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", wg_per_cu, U, NT ? "nt   " : "plain", tag, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0); }
      ROW(4, true, ids, "random") ROW(8, true, ids, "random") ROW(16, true, ids, "random") ROW(8, false, ids, "random") ROW(16, false, ids, "random")
      ROW(8, true, ids_sorted, "sorted") ROW(16, true, ids_sorted, "sorted")
#undef ROW
    }
    hipFree(table); hipFree(ids); hipFree(ids_sorted); hipFree(out);
  }
  return 0;
}
