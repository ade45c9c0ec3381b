// GRU4Rec-style user encoder (unirec/model/sequential/gru.py:13-35; the arithmetic is torch.nn.GRU's, 1 layer,
// batch_first, h0 = 0, gate order r,z,n):
//   x_t = E[item_seq[:,t]]                      all L steps run, including the left padding (zero rows)
//   gi  = x W_ih^T + b_ih                        ONE MFMA GEMM over all B*L tokens        (gemm_nt)
//   gh  = h_{t-1} W_hh^T + b_hh                  per step: [B,H] x [H,3H]                  (gemm_nt)
//   r = s(gi_r + gh_r); z = s(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h_t = (1-z) n + z h_{t-1}   (cell kernel)
//   user_emb = h_{L-1} W_d^T + b_d               only the last step is projected (gru.py:31-33 projects all L and slices)
// Activations are kept TIME-MAJOR ([L][B][.]) so that every step's matrices are contiguous and the weight gradients
// are two big token-dimension GEMMs after the backward sweep.  Backward = BPTT with saved gates.
#include "common.h"
#include "kernels.h"

namespace ur {

struct GruLayout { long long w_ih, w_hh, b_ih, b_hh, w_d, b_d, total; };
static GruLayout gru_layout(const UrGruCfg& c) {
  GruLayout l;
  long long o = 0;
  const long long d = c.d, H = c.H;
  l.w_ih = o; o += 3 * H * d;
  l.w_hh = o; o += 3 * H * H;
  l.b_ih = o; o += 3 * H;
  l.b_hh = o; o += 3 * H;
  l.w_d = o; o += d * H;
  l.b_d = o; o += d;
  l.total = o;
  return l;
}

struct GruWs {
  int* seq_tm;
  float *x, *gi, *gh, *h_all, *r, *z, *n, *hn;          // saved by forward
  float *dh, *dh_carry, *dgi, *dgh, *dx_tm, *w_ihT, *w_hhT, *w_dT, *tn_ws;
  long long total_floats;
};
static GruWs gru_carve(const UrGruCfg& c, float* base) {
  GruWs w;
  long long o = 0;
  auto take = [&](long long n) {
    float* p = base ? base + o : nullptr;
    o += (n + 63) & ~63LL;
    return p;
  };
  const long long B = c.B, L = c.L, d = c.d, H = c.H, M = B * L;
  w.seq_tm = (int*)take(M);
  w.x = take(M * d); w.gi = take(M * 3 * H); w.gh = take(B * 3 * H); w.h_all = take((L + 1) * B * H);
  w.r = take(M * H); w.z = take(M * H); w.n = take(M * H); w.hn = take(M * H);
  w.dh = take(B * H); w.dh_carry = take(B * H); w.dgi = take(M * 3 * H); w.dgh = take(M * 3 * H); w.dx_tm = take(M * d);
  w.w_ihT = take(3 * H * d); w.w_hhT = take(3 * H * H); w.w_dT = take(d * H);
  long long tn = gemm_tn_ws_floats((int)M, (int)(3 * H), (int)H);
  if (gemm_tn_ws_floats((int)M, (int)(3 * H), (int)d) > tn) tn = gemm_tn_ws_floats((int)M, (int)(3 * H), (int)d);
  if (gemm_tn_ws_floats((int)B, (int)d, (int)H) > tn) tn = gemm_tn_ws_floats((int)B, (int)d, (int)H);
  w.tn_ws = take(tn);
  w.total_floats = o;
  return w;
}

static int gru_check(const UrGruCfg* c) {
  UR_REQUIRE(c != nullptr, UR_ERR_ARG, "gru: null cfg");
  UR_REQUIRE(c->B > 0 && c->L > 0, UR_ERR_ARG, "gru: B=%d L=%d", c->B, c->L);
  UR_REQUIRE(c->d > 0 && c->d % 4 == 0 && c->d <= 512, UR_ERR_ARG, "gru: embedding_size d=%d must be a multiple of 4, <= 512", c->d);
  UR_REQUIRE(c->H > 0 && c->H % 4 == 0, UR_ERR_ARG, "gru: hidden_size H=%d must be a multiple of 4", c->H);
  UR_REQUIRE((long long)c->B * c->L < (1LL << 31), UR_ERR_ARG, "gru: B*L too large");
  UR_REQUIRE(c->p_drop >= 0.f && c->p_drop < 1.f, UR_ERR_ARG, "gru: dropout_prob %g not in [0, 1)", (double)c->p_drop);
  return UR_OK;
}

// seq_tm[t*B + b] = seq[b*L + t]
__global__ void ids_time_major_kernel(const int* __restrict__ seq, int B, int L, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  const int t = i / B, b = i % B;
  out[i] = seq[b * L + t];
}
// dst[(b*L + t), :] = src[(t*B + b), :]
__global__ void rows_batch_major_kernel(const float4* __restrict__ src, int B, int L, int d4, float4* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * L * d4) return;
  const int c = (int)(i % d4);
  const long long row = i / d4, b = row / L, t = row % L;
  dst[i] = src[(t * B + b) * d4 + c];
}

__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev, int B, int H, float* __restrict__ h_out,
                                                           float* __restrict__ r_s, float* __restrict__ z_s, float* __restrict__ n_s,
                                                           float* __restrict__ hn_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i % H;
  const float* gib = gi + (long long)b * 3 * H;
  const float* ghb = gh + (long long)b * 3 * H;
  const float r = 1.0f / (1.0f + expf(-(gib[j] + ghb[j])));
  const float z = 1.0f / (1.0f + expf(-(gib[H + j] + ghb[H + j])));
  const float hn = ghb[2 * H + j];
  const float n = tanhf(gib[2 * H + j] + r * hn);
  h_out[i] = (1.0f - z) * n + z * h_prev[i];
  r_s[i] = r; z_s[i] = z; n_s[i] = n; hn_s[i] = hn;
}

__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ r_s,
                                                           const float* __restrict__ z_s, const float* __restrict__ n_s,
                                                           const float* __restrict__ hn_s, const float* __restrict__ h_prev, int B,
                                                           int H, float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dh_carry) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i % H;
  const float g = dh[i], r = r_s[i], z = z_s[i], n = n_s[i], hn = hn_s[i];
  const float dn = g * (1.0f - z);
  const float dz = g * (h_prev[i] - n);
  const float dan = dn * (1.0f - n * n);
  const float daz = dz * z * (1.0f - z);
  const float dar = dan * hn * r * (1.0f - r);
  float* gi = dgi + (long long)b * 3 * H;
  float* gh = dgh + (long long)b * 3 * H;
  gi[j] = dar; gi[H + j] = daz; gi[2 * H + j] = dan;
  gh[j] = dar; gh[H + j] = daz; gh[2 * H + j] = dan * r;
  dh_carry[i] = g * z;
}

}  // namespace ur

using namespace ur;

extern "C" int64_t ur_gru_param_layout(const UrGruCfg* cfg, int64_t* offsets_out) {
  int rc = gru_check(cfg);
  if (rc) return rc;
  const GruLayout l = gru_layout(*cfg);
  if (offsets_out) {
    offsets_out[0] = l.w_ih; offsets_out[1] = l.w_hh; offsets_out[2] = l.b_ih; offsets_out[3] = l.b_hh;
    offsets_out[4] = l.w_d; offsets_out[5] = l.b_d;
  }
  return l.total;
}

extern "C" int64_t ur_gru_workspace_bytes(const UrGruCfg* cfg) {
  int rc = gru_check(cfg);
  if (rc) return rc;
  return gru_carve(*cfg, nullptr).total_floats * (int64_t)sizeof(float);
}

extern "C" int ur_gru_fwd(const UrGruCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                          const int32_t* item_seq, float* user_emb, void* ws, void* stream) {
  int rc = gru_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(item_table && dense && item_seq && user_emb && ws && n_items > 0, UR_ERR_ARG, "ur_gru_fwd: null pointer");
  const UrGruCfg& c = *cfg;
  hipStream_t st = as_stream(stream);
  const GruLayout lay = gru_layout(c);
  GruWs w = gru_carve(c, (float*)ws);
  const int B = c.B, L = c.L, d = c.d, H = c.H, M = B * L;
  {
    ProfScope ps(PC_GRU, st, 0);
    hipLaunchKernelGGL(ids_time_major_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, item_seq, B, L, w.seq_tm);
    UR_LAUNCH_CHECK();
  }
  if ((rc = gather_rows(item_table, w.seq_tm, 4, M, d, w.x, st))) return rc;
  if (c.p_drop > 0.f && (rc = drop_rows(w.x, M, d, drop_spec(c.p_drop, c.drop_seed, c.drop_step, 0), w.x, st))) return rc;
  GemmArgs g{};
  g.A = w.x; g.lda = d; g.W = dense + lay.w_ih; g.ldw = d; g.C = w.gi; g.ldc = 3 * H; g.M = M; g.N = 3 * H; g.K = d; g.bias = dense + lay.b_ih;
  if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
  UR_HIP(hipMemsetAsync(w.h_all, 0, sizeof(float) * B * H, st));
  for (int t = 0; t < L; ++t) {
    const long long o = (long long)t * B * H;
    g = GemmArgs{};
    g.A = w.h_all + o; g.lda = H; g.W = dense + lay.w_hh; g.ldw = H; g.C = w.gh; g.ldc = 3 * H; g.M = B; g.N = 3 * H; g.K = H;
    g.bias = dense + lay.b_hh;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
    ProfScope ps(PC_GRU, st, 0);
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(cdiv((long long)B * H, 256)), dim3(256), 0, st, w.gi + (long long)t * B * 3 * H, w.gh,
                       w.h_all + o, B, H, w.h_all + o + (long long)B * H, w.r + o, w.z + o, w.n + o, w.hn + o);
    UR_LAUNCH_CHECK();
  }
  g = GemmArgs{};
  g.A = w.h_all + (long long)L * B * H; g.lda = H; g.W = dense + lay.w_d; g.ldw = H; g.C = user_emb; g.ldc = d; g.M = B; g.N = d; g.K = H;
  g.bias = dense + lay.b_d;
  return gemm_nt(g, PRO_NONE, EPI_BIAS, st);
}

extern "C" int ur_gru_bwd(const UrGruCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                          const int32_t* item_seq, const float* d_user_emb, void* ws, float* dense_grad, float* d_emb_rows,
                          void* stream) {
  int rc = gru_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(dense && d_user_emb && ws && dense_grad && d_emb_rows, UR_ERR_ARG, "ur_gru_bwd: null pointer");
  (void)item_table; (void)n_items; (void)item_seq;
  const UrGruCfg& c = *cfg;
  hipStream_t st = as_stream(stream);
  const GruLayout lay = gru_layout(c);
  GruWs w = gru_carve(c, (float*)ws);
  const int B = c.B, L = c.L, d = c.d, H = c.H, M = B * L;
  if ((rc = transpose(dense + lay.w_ih, 3 * H, d, w.w_ihT, st))) return rc;   // [d, 3H]
  if ((rc = transpose(dense + lay.w_hh, 3 * H, H, w.w_hhT, st))) return rc;   // [H, 3H]
  if ((rc = transpose(dense + lay.w_d, d, H, w.w_dT, st))) return rc;         // [H, d]
  // dense head: dW_d = d_out^T h_L ; db_d ; dh_L = d_out W_d
  const float* hL = w.h_all + (long long)L * B * H;
  if ((rc = gemm_tn(d_user_emb, d, hL, H, B, d, H, 0, 0, dense_grad + lay.w_d, H, dense_grad + lay.b_d, w.tn_ws, st))) return rc;
  GemmArgs g{};
  g.A = d_user_emb; g.lda = d; g.W = w.w_dT; g.ldw = d; g.C = w.dh; g.ldc = H; g.M = B; g.N = H; g.K = d;
  if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
  for (int t = L - 1; t >= 0; --t) {
    const long long o = (long long)t * B * H, o3 = (long long)t * B * 3 * H;
    {
      ProfScope ps(PC_GRU, st, 0);
      hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(cdiv((long long)B * H, 256)), dim3(256), 0, st, w.dh, w.r + o, w.z + o, w.n + o,
                         w.hn + o, w.h_all + o, B, H, w.dgi + o3, w.dgh + o3, w.dh_carry);
      UR_LAUNCH_CHECK();
    }
    g = GemmArgs{};   // dh_{t-1} = dgh_t W_hh + dh_t * z
    g.A = w.dgh + o3; g.lda = 3 * H; g.W = w.w_hhT; g.ldw = 3 * H; g.C = w.dh; g.ldc = H; g.M = B; g.N = H; g.K = 3 * H;
    g.aux = w.dh_carry; g.ldaux = H;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD, st))) return rc;
  }
  // weight gradients over all (t, b) tokens at once (time-major rows on both operands)
  if ((rc = gemm_tn(w.dgh, 3 * H, w.h_all, H, M, 3 * H, H, 0, 0, dense_grad + lay.w_hh, H, dense_grad + lay.b_hh, w.tn_ws, st))) return rc;
  if ((rc = gemm_tn(w.dgi, 3 * H, w.x, d, M, 3 * H, d, 0, 0, dense_grad + lay.w_ih, d, dense_grad + lay.b_ih, w.tn_ws, st))) return rc;
  g = GemmArgs{};   // dx = dgi W_ih
  g.A = w.dgi; g.lda = 3 * H; g.W = w.w_ihT; g.ldw = 3 * H; g.C = w.dx_tm; g.ldc = d; g.M = M; g.N = d; g.K = 3 * H;
  if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
  if (c.p_drop > 0.f && (rc = drop_rows(w.dx_tm, M, d, drop_spec(c.p_drop, c.drop_seed, c.drop_step, 0), w.dx_tm, st))) return rc;
  ProfScope ps(PC_GRU, st, 0);
  hipLaunchKernelGGL(rows_batch_major_kernel, dim3(cdiv((long long)M * (d / 4), 256)), dim3(256), 0, st, (const float4*)w.dx_tm, B, L, d / 4,
                     (float4*)d_emb_rows);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
