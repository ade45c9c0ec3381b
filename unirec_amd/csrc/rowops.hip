// Row-wise HBM-bound kernels: embedding gather, fused gather+pos+LayerNorm, LayerNorm backward.
// One row is owned by a group of TPR consecutive lanes (TPR | 64), each lane holding float4 column
// chunks c = t, t+TPR, ...; row reductions are xor-shuffles inside the group (no LDS).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace ur {

static inline int pick_tpr(int d) {
  int d4 = d / 4, t = 4;
  while (t < d4 && t < 32) t <<= 1;
  return t;  // 4, 8, 16 or 32; rows wider than 4*TPR floats loop (VPT <= MAXV)
}
constexpr int MAXV = 4;  // float4 chunks per lane: d <= 4*32*4 = 512
typedef float vf4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ gather
// UNROLL rows in flight per lane group: random 512-B row reads need many outstanding loads per CU.
template <typename IdxT, int TPR, int UNROLL>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table, const IdxT* __restrict__ idx,
                                                     long long n, int d4, float4* __restrict__ out, long long n_rows) {
  const int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const long long stride = (long long)gridDim.x * groups * UNROLL;
  for (long long base = ((long long)blockIdx.x * groups + g) * UNROLL; base < n; base += stride) {
    long long id[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) id[u] = (base + u < n) ? UR_ROW((long long)idx[base + u], n_rows) : 0;
    for (int c = t; c < d4; c += TPR) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const vf4 t4 = __builtin_nontemporal_load((const vf4*)&table[id[u] * d4 + c]);   // read once: streaming hint
        v[u] = make_float4(t4.x, t4.y, t4.z, t4.w);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (base + u < n) __builtin_nontemporal_store(*(const vf4*)&v[u], (vf4*)&out[(base + u) * d4 + c]);
    }
  }
}

template <typename IdxT>
static int launch_gather(const float* table, const void* idx, long long n, int d, float* out, hipStream_t st, long long n_rows) {
  const int d4 = d / 4;
  const int tpr = pick_tpr(d);
  const int groups = 256 / tpr;
  constexpr int U = 4;
  long long blocks = (n + (long long)groups * U - 1) / ((long long)groups * U);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
#define GO(T) hipLaunchKernelGGL((gather_kernel<IdxT, T, U>), dim3((unsigned)blocks), dim3(256), 0, st, \
                                 (const float4*)table, (const IdxT*)idx, n, d4, (float4*)out, n_rows)
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int gather_rows(const float* table, const void* idx, int idx_bytes, long long n, int d, float* out, hipStream_t st, long long n_rows) {
  if (n == 0) return UR_OK;
  ProfScope ps(PC_GATHER, st, (double)n * (d * 4.0 + idx_bytes));  // algorithmic READ bytes
  return idx_bytes == 8 ? launch_gather<long long>(table, idx, n, d, out, st, n_rows) : launch_gather<int>(table, idx, n, d, out, st, n_rows);
}

// ------------------------------------------------------------------------- gather + pos + LayerNorm
// x0[b,l,:] = LN(E[item_seq[b,l]] + P[l])  (sasrec.py:60-69). Writes y, xhat, rstd.
template <int TPR>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int* __restrict__ seq, const float4* __restrict__ table,
                                                           const float4* __restrict__ pos, const float4* __restrict__ gamma,
                                                           const float4* __restrict__ beta, float eps, int M, int L, int d4,
                                                           float4* __restrict__ y, float4* __restrict__ xhat,
                                                           float* __restrict__ rstd_out, const int* __restrict__ tok,
                                                           const int* __restrict__ m_dev, DropSpec drop, long long n_rows) {
  const int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const float inv_d = 1.0f / (float)(d4 * 4);
  if (m_dev) M = min(M, *m_dev);
  for (int row = blockIdx.x * groups + g; row < M; row += gridDim.x * groups) {
    const int full = tok ? tok[row] : row;   // position of this (compact) row in the padded [B, L] token grid
    const long long id = UR_ROW(seq[full], n_rows);
    const int l = full % L;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        float4 a = table[id * d4 + c];
        if (pos) {
          const float4 p = pos[(long long)l * d4 + c];
          a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        v[k] = a;
        s += (a.x + a.y) + (a.z + a.w);
      }
    }
    const float mean = group_sum<TPR>(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
        q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
      }
    }
    const float rstd = 1.0f / sqrtf(group_sum<TPR>(q) * inv_d + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        const float4 gm = gamma[c], bt = beta[c];
        float4 h, o;
        h.x = v[k].x * rstd; h.y = v[k].y * rstd; h.z = v[k].z * rstd; h.w = v[k].w * rstd;
        o.x = h.x * gm.x + bt.x; o.y = h.y * gm.y + bt.y; o.z = h.z * gm.z + bt.z; o.w = h.w * gm.w + bt.w;
        xhat[(long long)row * d4 + c] = h;
        if (drop.thresh) o = drop4(o, mix32((unsigned)full ^ drop.key), (unsigned)(c * 4), drop);
        y[(long long)row * d4 + c] = o;
      }
    }
    if (t == 0) rstd_out[row] = rstd;
  }
}

int embed_ln_fwd(const int* seq, const float* table, const float* pos, const float* gamma, const float* beta,
                 float eps, int M, int L, int d, float* y, float* xhat, float* rstd, hipStream_t st, const int* tok,
                 const int* m_dev, const DropSpec* drop, long long n_rows) {
  const DropSpec ds = drop ? *drop : DropSpec{};
  ProfScope ps(PC_ROWOPS, st, (double)M * d * 4.0 * 3);
  const int tpr = pick_tpr(d), groups = 256 / tpr;
  int blocks = cdiv(M, groups);
  if (blocks > 4096) blocks = 4096;
#define GO(T) hipLaunchKernelGGL((embed_ln_fwd_kernel<T>), dim3(blocks), dim3(256), 0, st, seq, (const float4*)table, \
                                 (const float4*)pos, (const float4*)gamma, (const float4*)beta, eps, M, L, d / 4,      \
                                 (float4*)y, (float4*)xhat, rstd, tok, m_dev, ds, n_rows)
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// ------------------------------------------------------------------------- stand-alone dropout
// out[r,:] = dropout(x[r,:]) (out may alias x): for the encoders whose dropout sits on a plain tensor (GRU: the gathered
// embeddings, gru.py:29; AttHist: the pooled output, modules.py:242).  The backward applies the same call to the gradient.
__global__ __launch_bounds__(256) void drop_rows_kernel(const float4* x, long long rows, int d4, DropSpec drop, float4* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * d4) return;
  const int r = (int)(i / d4), c = (int)(i % d4);
  out[i] = drop4(x[i], drop_rowkey(drop, r), (unsigned)(c * 4), drop);
}
int drop_rows(const float* x, long long rows, int d, const DropSpec& drop, float* out, hipStream_t st) {
  if (!drop.thresh) {
    if (out != x) UR_HIP(hipMemcpyAsync(out, x, (size_t)rows * d * sizeof(float), hipMemcpyDeviceToDevice, st));
    return UR_OK;
  }
  ProfScope ps(PC_ROWOPS, st, (double)rows * d * 4.0 * 2);
  hipLaunchKernelGGL(drop_rows_kernel, dim3(cdiv(rows * (d / 4), 256)), dim3(256), 0, st, (const float4*)x, rows, d / 4, drop, (float4*)out);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// ------------------------------------------------------------------------- residual + LayerNorm fwd
// y = LN(x + res) (res may be null); fallback for rows wider than the GEMM-fused epilogue supports.
template <int TPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ res,
                                                     const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                                     float eps, int M, int d4, float4* __restrict__ y,
                                                     float4* __restrict__ xhat, float* __restrict__ rstd_out) {
  const int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const float inv_d = 1.0f / (float)(d4 * 4);
  for (int row = blockIdx.x * groups + g; row < M; row += gridDim.x * groups) {
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        float4 a = x[(long long)row * d4 + c];
        if (res) {
          const float4 r = res[(long long)row * d4 + c];
          a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
        }
        v[k] = a;
        s += (a.x + a.y) + (a.z + a.w);
      }
    }
    const float mean = group_sum<TPR>(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
        q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
      }
    }
    const float rstd = 1.0f / sqrtf(group_sum<TPR>(q) * inv_d + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        const float4 gm = gamma[c], bt = beta[c];
        float4 h, o;
        h.x = v[k].x * rstd; h.y = v[k].y * rstd; h.z = v[k].z * rstd; h.w = v[k].w * rstd;
        o.x = h.x * gm.x + bt.x; o.y = h.y * gm.y + bt.y; o.z = h.z * gm.z + bt.z; o.w = h.w * gm.w + bt.w;
        xhat[(long long)row * d4 + c] = h;
        y[(long long)row * d4 + c] = o;
      }
    }
    if (t == 0) rstd_out[row] = rstd;
  }
}

int ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps, int M, int d, float* y,
           float* xhat, float* rstd, hipStream_t st) {
  ProfScope ps(PC_ROWOPS, st, (double)M * d * 4.0 * 4);
  const int tpr = pick_tpr(d), groups = 256 / tpr;
  int blocks = cdiv(M, groups);
  if (blocks > 4096) blocks = 4096;
#define GO(T) hipLaunchKernelGGL((ln_fwd_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float4*)x, (const float4*)res, \
                                 (const float4*)gamma, (const float4*)beta, eps, M, d / 4, (float4*)y, (float4*)xhat, rstd)
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// ------------------------------------------------------------------------------- LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// dx_out = dx (+ add_in when non-null: the residual branch's gradient; add_in may alias dx_out)
// zero_rows_of_id0: when seq != null, rows whose id is 0 are written as zeros (padding_idx).
// Per-block partials of dgamma = sum dy*xhat and dbeta = sum dy go to part[blk][2][d].
template <int TPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float4* __restrict__ gamma,
                                                     const float4* add_in, const int* __restrict__ seq, int M, int d4,
                                                     float4* dx_out, float* __restrict__ part, const int* __restrict__ m_dev,
                                                     const int* __restrict__ out_rows, DropSpec in_drop, DropSpec out_drop,
                                                     float4* __restrict__ dx_drop) {
  constexpr int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const float inv_d = 1.0f / (float)(d4 * 4);
  if (m_dev) M = min(M, *m_dev);
  float4 dg[MAXV], db[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) dg[k] = db[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int row = blockIdx.x * groups + g; row < M; row += gridDim.x * groups) {
    float4 gy[MAXV], h[MAXV];
    float s1 = 0.f, s2 = 0.f;
    const unsigned rk_in = in_drop.thresh ? drop_rowkey(in_drop, row) : 0u;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        float4 y = dy[(long long)row * d4 + c];
        if (in_drop.thresh) y = drop4(y, rk_in, (unsigned)(c * 4), in_drop);
        h[k] = xhat[(long long)row * d4 + c];
        const float4 gm = gamma[c];
        dg[k].x += y.x * h[k].x; dg[k].y += y.y * h[k].y; dg[k].z += y.z * h[k].z; dg[k].w += y.w * h[k].w;
        db[k].x += y.x; db[k].y += y.y; db[k].z += y.z; db[k].w += y.w;
        gy[k].x = y.x * gm.x; gy[k].y = y.y * gm.y; gy[k].z = y.z * gm.z; gy[k].w = y.w * gm.w;
        s1 += (gy[k].x + gy[k].y) + (gy[k].z + gy[k].w);
        s2 += (gy[k].x * h[k].x + gy[k].y * h[k].y) + (gy[k].z * h[k].z + gy[k].w * h[k].w);
      }
    }
    const float m1 = group_sum<TPR>(s1) * inv_d;
    const float m2 = group_sum<TPR>(s2) * inv_d;
    const float r = rstd[row];
    const bool zero = seq != nullptr && seq[row] == 0;
    const unsigned rk_out = out_drop.thresh ? drop_rowkey(out_drop, row) : 0u;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        float4 o;
        o.x = r * (gy[k].x - m1 - h[k].x * m2);
        o.y = r * (gy[k].y - m1 - h[k].y * m2);
        o.z = r * (gy[k].z - m1 - h[k].z * m2);
        o.w = r * (gy[k].w - m1 - h[k].w * m2);
        if (add_in) {
          const float4 a = add_in[(long long)row * d4 + c];
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        if (zero) o = make_float4(0.f, 0.f, 0.f, 0.f);
        dx_out[(long long)(out_rows ? out_rows[row] : row) * d4 + c] = o;
        if (out_drop.thresh) dx_drop[(long long)row * d4 + c] = drop4(o, rk_out, (unsigned)(c * 4), out_drop);
      }
    }
  }
  // block-level reduction of dgamma/dbeta over the `groups` row groups (fixed order => deterministic)
  extern __shared__ __attribute__((aligned(16))) float red[];   // [groups][2][d + 4]: sized for the actual row width (occupancy)
  const int d = d4 * 4, rs = d + 4;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = t + k * TPR;
    if (c < d4) {
      *(float4*)&red[(g * 2 + 0) * rs + c * 4] = dg[k];
      *(float4*)&red[(g * 2 + 1) * rs + c * 4] = db[k];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += 256) {
    const int which = i / d, col = i % d;
    float acc = 0.f;
#pragma unroll
    for (int gg = 0; gg < groups; ++gg) acc += red[(gg * 2 + which) * rs + col];
    part[((long long)blockIdx.x * 2 + which) * d + col] = acc;
  }
}

// out[which*d + col] = sum_blk part[blk][which][col]   (which in {0: dgamma, 1: dbeta})
// 1024 threads = 64 columns x 16 row-slices: each thread sums nblk/16 partials, LDS combines the slices in
// a fixed order (deterministic).  One block per 64 columns.
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ part, int nblk, int width,
                                                               float* __restrict__ out0, float* __restrict__ out1, int d) {
  __shared__ float red[16][65];
  const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  float acc = 0.f;
  if (i < width)
    for (int b = s; b < nblk; b += 16) acc += part[(long long)b * width + i];
  red[s][c] = acc;
  __syncthreads();
  if (s == 0 && i < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    if (i < d) out0[i] = t;
    else out1[i - d] = t;
  }
}

int ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, const float* add_in,
           const int* seq, int M, int d, float* dx, float* dgamma, float* dbeta, float* part_ws, hipStream_t st,
           ReduceBatch* defer, const int* m_dev, const int* out_rows, const DropSpec* in_drop, const DropSpec* out_drop,
           float* dx_drop) {
  ProfScope ps(PC_ROWOPS, st, (double)M * d * 4.0 * 3);
  const DropSpec di = in_drop ? *in_drop : DropSpec{}, d_o = out_drop ? *out_drop : DropSpec{};
  if (d_o.thresh && (!dx_drop || out_rows)) return fail(UR_ERR_ARG, "ln_bwd: out_drop needs dx_drop and no out_rows");
  const int tpr = pick_tpr(d), groups = 256 / tpr;
  int blocks = cdiv(M, groups * 4);   // 4 rows per lane group
  if (blocks > LN_BWD_MAX_BLOCKS) blocks = LN_BWD_MAX_BLOCKS;
  const size_t lds = (size_t)groups * 2 * (d + 4) * sizeof(float);
  if (blocks < 1) blocks = 1;
#define GO(T) hipLaunchKernelGGL((ln_bwd_kernel<T>), dim3(blocks), dim3(256), lds, st, (const float4*)dy, (const float4*)xhat, \
                                 rstd, (const float4*)gamma, (const float4*)add_in, seq, M, d / 4, (float4*)dx, part_ws, m_dev, \
                                 out_rows, di, d_o, (float4*)dx_drop)
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  if (defer) {   // part_ws[blk][2d] = per-block (dgamma | dbeta) partial sums
    if (defer->full(2)) {
      int rc = reduce_batch(*defer, st);
      if (rc) return rc;
    }
    defer->add(part_ws, 2 * d, blocks, d, d, dgamma, d);
    defer->add(part_ws + d, 2 * d, blocks, d, d, dbeta, d);
    return UR_OK;
  }
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(cdiv(2 * d, 64)), dim3(1024), 0, st, part_ws, blocks, 2 * d, dgamma, dbeta, d);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// --------------------------------------------------------------- position-embedding gradient
// dpos[l,:] = sum_b dx[b,l,:]   (position_embedding has no padding index: sasrec.py:25)
__global__ __launch_bounds__(256) void pos_grad_kernel(const float* __restrict__ dx, int B, int L, int d,
                                                       float* __restrict__ dpos) {
  const int l = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dx[((long long)b * L + l) * d + c];
    dpos[(long long)l * d + c] = acc;
  }
}

int pos_grad(const float* dx, int B, int L, int d, float* dpos, hipStream_t st) {
  hipLaunchKernelGGL(pos_grad_kernel, dim3(L), dim3(d < 256 ? ((d + 63) / 64) * 64 : 256), 0, st, dx, B, L, d, dpos);
  UR_LAUNCH_CHECK();
  return UR_OK;
}


// ------------------------------------------------------------------------------- pooled history (AvgHist / SVD++)
// user_emb[b,:] = (seq_len[b] + 1)^(-alpha) * sum_l E[item_seq[b,l],:]   (+ base[b,:] when non-null)
// unirec/model/sequential/avghist.py:35-42, svdplusplus.py:32-40.  One lane group per row, float4 per lane.
template <int TPR>
__global__ __launch_bounds__(256) void pool_rows_fwd_kernel(const float4* __restrict__ table, const int* __restrict__ seq,
                                                            const long long* __restrict__ seq_len, const float4* __restrict__ base,
                                                            float alpha, int B, int L, int d4, float4* __restrict__ out, long long n_rows) {
  constexpr int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int b = blockIdx.x * groups + g;
  if (b >= B) return;
  const float coef = powf((float)(seq_len[b] + 1), -alpha);
  float4 acc[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < L; ++l) {   // fixed summation order over positions
    const long long id = UR_ROW(seq[(long long)b * L + l], n_rows);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = t + k * TPR;
      if (c < d4) {
        const float4 e = table[id * d4 + c];
        acc[k].x += e.x; acc[k].y += e.y; acc[k].z += e.z; acc[k].w += e.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = t + k * TPR;
    if (c < d4) {
      float4 o = make_float4(coef * acc[k].x, coef * acc[k].y, coef * acc[k].z, coef * acc[k].w);
      if (base) {
        const float4 u = base[(long long)b * d4 + c];
        o.x = u.x + o.x; o.y = u.y + o.y; o.z = u.z + o.z; o.w = u.w + o.w;
      }
      out[(long long)b * d4 + c] = o;
    }
  }
}
// rows[b*L + l, :] = (seq_len[b] + 1)^(-alpha) * d_user[b,:]: the gradient of every gathered history row
__global__ __launch_bounds__(256) void pool_rows_bwd_kernel(const float4* __restrict__ d_user, const long long* __restrict__ seq_len,
                                                            float alpha, int B, int L, int d4, float4* __restrict__ rows) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * L * d4) return;
  const long long c = i % d4, row = i / d4, b = row / L;
  const float coef = powf((float)(seq_len[b] + 1), -alpha);
  const float4 g = d_user[b * d4 + c];
  rows[i] = make_float4(coef * g.x, coef * g.y, coef * g.z, coef * g.w);
}

}  // namespace ur

extern "C" int ur_embedding_gather_f32(const float* table, int64_t n_rows, int d, const void* idx, int idx_bytes,
                                       int64_t n, float* out, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(table && ((out && idx) || n == 0), UR_ERR_ARG, "ur_embedding_gather_f32: null pointer");
  UR_REQUIRE(d > 0 && d % 4 == 0 && d <= 512, UR_ERR_ARG, "ur_embedding_gather_f32: d=%d must be a multiple of 4, <= 512", d);
  UR_REQUIRE(idx_bytes == 4 || idx_bytes == 8, UR_ERR_ARG, "ur_embedding_gather_f32: idx_bytes=%d (4 or 8)", idx_bytes);
  UR_REQUIRE(n >= 0 && n_rows > 0, UR_ERR_ARG, "ur_embedding_gather_f32: n=%lld n_rows=%lld", (long long)n, (long long)n_rows);
  return ur::gather_rows(table, idx, idx_bytes, n, d, out, ur::as_stream(stream), n_rows);
}

extern "C" int ur_pool_rows_fwd(const float* table, int64_t n_rows, int32_t d, const int32_t* item_seq, const int64_t* seq_len,
                                const float* base, float alpha, int32_t B, int32_t L, float* user_emb, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(table && item_seq && seq_len && user_emb, UR_ERR_ARG, "ur_pool_rows_fwd: null pointer");
  UR_REQUIRE(d > 0 && d % 4 == 0 && d <= 512 && B > 0 && L > 0 && n_rows > 0, UR_ERR_ARG, "ur_pool_rows_fwd: d=%d B=%d L=%d", d, B, L);
  hipStream_t st = ur::as_stream(stream);
  ur::ProfScope ps(ur::PC_ROWOPS, st, (double)B * L * d * 4.0);
  const int tpr = ur::pick_tpr(d), groups = 256 / tpr;
#define GO(T) hipLaunchKernelGGL((ur::pool_rows_fwd_kernel<T>), dim3(ur::cdiv(B, groups)), dim3(256), 0, st, (const float4*)table, item_seq, \
                                 (const long long*)seq_len, (const float4*)base, alpha, B, L, d / 4, (float4*)user_emb, (long long)n_rows)
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_pool_rows_bwd(const float* d_user_emb, const int64_t* seq_len, float alpha, int32_t B, int32_t L, int32_t d,
                                float* d_rows, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(d_user_emb && seq_len && d_rows, UR_ERR_ARG, "ur_pool_rows_bwd: null pointer");
  UR_REQUIRE(d > 0 && d % 4 == 0 && B > 0 && L > 0, UR_ERR_ARG, "ur_pool_rows_bwd: d=%d B=%d L=%d", d, B, L);
  hipStream_t st = ur::as_stream(stream);
  ur::ProfScope ps(ur::PC_ROWOPS, st, (double)B * L * d * 4.0);
  hipLaunchKernelGGL(ur::pool_rows_bwd_kernel, dim3(ur::cdiv((long long)B * L * (d / 4), 256)), dim3(256), 0, st, (const float4*)d_user_emb,
                     (const long long*)seq_len, alpha, B, L, d / 4, (float4*)d_rows);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
