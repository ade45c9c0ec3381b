// AttHist user encoder: attention pooling of the history, no sequence order and NO mask.
// Reference: unirec/model/sequential/atthist.py:9-23 + AttentionMergeLayer (unirec/model/modules.py:226-244):
//   z[b,l,:] = E[item_seq[b,l]] W^T + bias        s[b,l] = z[b,l,:] . h        p = softmax_l(s)  (padding positions take part)
//   user_emb[b,:] = sum_l p[b,l] z[b,l,:]
// Built on the shared kernels: row gather, gemm_nt (+bias) for z, gemm_tn for dW/db, gemm_nt on W^T for dx, the deferred
// reduction for dh; the merge itself (dots with h, softmax over L, weighted sum, and its backward) is one wave per row b.
#include "common.h"
#include "kernels.h"

namespace ur {

// one wave per sequence: lanes over the d/4 float4 columns (strided), loop over positions
__global__ __launch_bounds__(256) void atthist_merge_fwd_kernel(const float4* __restrict__ z, const float4* __restrict__ h, int B, int L,
                                                                int d4, float* __restrict__ p_out, float4* __restrict__ out) {
  extern __shared__ float sh[];   // [4][L] scores -> probabilities
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + w;
  if (b >= B) return;
  float* s = sh + w * L;
  for (int l = 0; l < L; ++l) {
    float acc = 0.f;
    for (int c = lane; c < d4; c += 64) {
      const float4 a = z[((long long)b * L + l) * d4 + c], q = h[c];
      acc += (a.x * q.x + a.y * q.y) + (a.z * q.z + a.w * q.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) s[l] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  float m = -INFINITY;
  for (int l = lane; l < L; l += 64) m = fmaxf(m, s[l]);
  m = group_max<64>(m);
  float sum = 0.f;
  for (int l = lane; l < L; l += 64) sum += expf(s[l] - m);
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  for (int l = lane; l < L; l += 64) {
    const float pv = expf(s[l] - m) / sum;
    s[l] = pv;
    p_out[(long long)b * L + l] = pv;
  }
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < d4; c += 64) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {   // fixed order over positions
      const float pv = s[l];
      const float4 a = z[((long long)b * L + l) * d4 + c];
      o.x = fmaf(pv, a.x, o.x); o.y = fmaf(pv, a.y, o.y); o.z = fmaf(pv, a.z, o.z); o.w = fmaf(pv, a.w, o.w);
    }
    out[(long long)b * d4 + c] = o;
  }
}

// dz[b,l,:] = p_l * g + ds_l * h,  ds_l = p_l (g . z_l - sum_j p_j g . z_j);   dh_part[b,:] = sum_l ds_l z_l
__global__ __launch_bounds__(256) void atthist_merge_bwd_kernel(const float4* __restrict__ z, const float4* __restrict__ h,
                                                                const float* __restrict__ p, const float4* __restrict__ g, int B, int L,
                                                                int d4, float4* __restrict__ dz, float4* __restrict__ dh_part) {
  extern __shared__ float sh[];   // [4][L] ds
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + w;
  if (b >= B) return;
  float* ds = sh + w * L;
  float tot = 0.f;
  for (int l = 0; l < L; ++l) {
    float acc = 0.f;
    for (int c = lane; c < d4; c += 64) {
      const float4 a = z[((long long)b * L + l) * d4 + c], q = g[(long long)b * d4 + c];
      acc += (a.x * q.x + a.y * q.y) + (a.z * q.z + a.w * q.w);
    }
    acc = wave_sum(acc);   // dp_l
    const float pv = p[(long long)b * L + l];
    if (lane == 0) ds[l] = acc;
    tot = fmaf(pv, acc, tot);
  }
  __builtin_amdgcn_wave_barrier();
  for (int l = lane; l < L; l += 64) ds[l] = p[(long long)b * L + l] * (ds[l] - tot);
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < d4; c += 64) {
    const float4 q = g[(long long)b * d4 + c], hh = h[c];
    float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
      const float pv = p[(long long)b * L + l], dsl = ds[l];
      const float4 a = z[((long long)b * L + l) * d4 + c];
      dz[((long long)b * L + l) * d4 + c] = make_float4(fmaf(pv, q.x, dsl * hh.x), fmaf(pv, q.y, dsl * hh.y), fmaf(pv, q.z, dsl * hh.z),
                                                         fmaf(pv, q.w, dsl * hh.w));
      dh.x = fmaf(dsl, a.x, dh.x); dh.y = fmaf(dsl, a.y, dh.y); dh.z = fmaf(dsl, a.z, dh.z); dh.w = fmaf(dsl, a.w, dh.w);
    }
    dh_part[(long long)b * d4 + c] = dh;
  }
}

struct AttHistWs {
  float *x, *z, *p, *dz, *wT, *dh_part, *tn_ws, *g_out;
  long long floats;
};
static AttHistWs atthist_carve(const UrAttHistCfg& c, float* base) {
  AttHistWs w;
  long long o = 0;
  auto take = [&](long long n) {
    float* ptr = base ? base + o : nullptr;
    o += (n + 63) & ~63LL;
    return ptr;
  };
  const long long M = (long long)c.B * c.L, d = c.d;
  w.x = take(M * d); w.z = take(M * d); w.p = take(M); w.dz = take(M * d); w.wT = take(d * d); w.dh_part = take((long long)c.B * d);
  w.tn_ws = take(gemm_tn_ws_floats((int)M, (int)d, (int)d));
  w.g_out = take((long long)c.B * d);
  w.floats = o;
  return w;
}
static int atthist_check(const UrAttHistCfg* c) {
  UR_REQUIRE(c != nullptr, UR_ERR_ARG, "atthist: null cfg");
  UR_REQUIRE(c->B > 0 && c->L > 0 && c->d > 0 && c->d % 4 == 0 && c->d <= 512, UR_ERR_ARG, "atthist: B=%d L=%d d=%d", c->B, c->L, c->d);
  UR_REQUIRE((long long)c->B * c->L < (1LL << 31) && c->L <= 4096, UR_ERR_ARG, "atthist: B*L / L too large");
  UR_REQUIRE(c->p_drop >= 0.f && c->p_drop < 1.f, UR_ERR_ARG, "atthist: dropout_prob %g not in [0, 1)", (double)c->p_drop);
  return UR_OK;
}

}  // namespace ur

using namespace ur;

extern "C" int64_t ur_atthist_param_layout(const UrAttHistCfg* cfg, int64_t* offsets_out) {
  int rc = atthist_check(cfg);
  if (rc) return rc;
  const int64_t d = cfg->d;
  if (offsets_out) { offsets_out[0] = 0; offsets_out[1] = d * d; offsets_out[2] = d * d + d; }
  return d * d + 2 * d;
}

extern "C" int64_t ur_atthist_workspace_bytes(const UrAttHistCfg* cfg) {
  int rc = atthist_check(cfg);
  if (rc) return rc;
  return atthist_carve(*cfg, nullptr).floats * (int64_t)sizeof(float);
}

extern "C" int ur_atthist_fwd(const UrAttHistCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                              const int32_t* item_seq, float* user_emb, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  int rc = atthist_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(item_table && dense && item_seq && user_emb && ws && n_items > 0, UR_ERR_ARG, "ur_atthist_fwd: null pointer");
  const UrAttHistCfg& c = *cfg;
  hipStream_t st = as_stream(stream);
  AttHistWs w = atthist_carve(c, (float*)ws);
  const int M = c.B * c.L, d = c.d;
  const float *W = dense, *bias = dense + (long long)d * d, *h = dense + (long long)d * d + d;
  if ((rc = gather_rows(item_table, item_seq, 4, M, d, w.x, st, n_items))) return rc;
  GemmArgs g{};
  g.A = w.x; g.lda = d; g.W = W; g.ldw = d; g.C = w.z; g.ldc = d; g.M = M; g.N = d; g.K = d; g.bias = bias;
  if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
  ProfScope ps(PC_ROWOPS, st, (double)M * d * 4.0 * 2);
  hipLaunchKernelGGL(atthist_merge_fwd_kernel, dim3(cdiv(c.B, 4)), dim3(256), 4 * c.L * sizeof(float), st, (const float4*)w.z, (const float4*)h,
                     c.B, c.L, d / 4, w.p, (float4*)user_emb);
  UR_LAUNCH_CHECK();
  if (c.p_drop > 0.f) return drop_rows(user_emb, c.B, d, drop_spec(c.p_drop, c.drop_seed, c.drop_step, 0), user_emb, st);
  return UR_OK;
}

extern "C" int ur_atthist_bwd(const UrAttHistCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                              const int32_t* item_seq, const float* d_user_emb, void* ws, float* dense_grad, float* d_emb_rows,
                              void* stream) {
  UR_TRACE_SCOPE();
  int rc = atthist_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(dense && item_seq && d_user_emb && ws && dense_grad && d_emb_rows, UR_ERR_ARG, "ur_atthist_bwd: null pointer");
  (void)item_table; (void)n_items;
  const UrAttHistCfg& c = *cfg;
  hipStream_t st = as_stream(stream);
  AttHistWs w = atthist_carve(c, (float*)ws);
  const int M = c.B * c.L, d = c.d;
  const float *W = dense, *h = dense + (long long)d * d + d;
  float *dW = dense_grad, *db = dense_grad + (long long)d * d, *dh = dense_grad + (long long)d * d + d;
  if (c.p_drop > 0.f) {   // gradient of the merge output = dropout-masked gradient of user_emb
    if ((rc = drop_rows(d_user_emb, c.B, d, drop_spec(c.p_drop, c.drop_seed, c.drop_step, 0), w.g_out, st))) return rc;
    d_user_emb = w.g_out;
  }
  {
    ProfScope ps(PC_ROWOPS, st, (double)M * d * 4.0 * 3);
    hipLaunchKernelGGL(atthist_merge_bwd_kernel, dim3(cdiv(c.B, 4)), dim3(256), 4 * c.L * sizeof(float), st, (const float4*)w.z, (const float4*)h,
                       w.p, (const float4*)d_user_emb, c.B, c.L, d / 4, (float4*)w.dz, (float4*)w.dh_part);
    UR_LAUNCH_CHECK();
  }
  ReduceBatch rb;
  rb.add(w.dh_part, d, c.B, d, d, dh, d);                 // dh = sum_b dh_part[b,:]  (fixed order)
  if ((rc = gemm_tn(w.dz, d, w.x, d, M, d, d, 0, 0, dW, d, db, w.tn_ws, st, &rb))) return rc;   // dW = dz^T x, db = colsum(dz)
  if ((rc = reduce_batch(rb, st))) return rc;
  if ((rc = transpose(W, d, d, w.wT, st))) return rc;
  GemmArgs g{};                                              // d x = dz W
  g.A = w.dz; g.lda = d; g.W = w.wT; g.ldw = d; g.C = d_emb_rows; g.ldc = d; g.M = M; g.N = d; g.K = d;
  return gemm_nt(g, PRO_NONE, EPI_NONE, st);
}
