// ConvFormer / FASTConvFormer user encoders.
// Reference: unirec/model/sequential/convformer.py:16-129, fastconvformer.py:20-81 (https://arxiv.org/abs/2308.02925).
//   x0 = LN(E[item_seq] + P[0..L-1])                                           (no attention mask anywhere: padding takes part)
//   per layer:  t = mix(x) + x ; y1 = LN(t) ; y = LN(act(y1 W1^T + b1) W2^T + b2 + y1)
//   ConvFormer  mix(x)[l,c] = bias[c] + sum_k w[c,k] xpad[l+k,c],  xpad = [prefix of K-1 rows ; x]   (depth-wise Conv1d)
//               prefix: circular = the last K-1 rows, reflect = the last K-1 rows reversed, constant = zeros
//   FAST        mix(x)[l,c] = (1/sqrt(L)) sum_k w[k,c] x[(l-k) mod L, c]   -- what rfft * rfft -> irfft with norm='ortho'
//               computes (a circular convolution with the zero-padded kernel), evaluated here in the time domain
//   output      x_L[:, L-1, :]   or (seq_merge)   sum_l x_L[:, l, :] 10^(decay (1 - l/(L-1))) / sqrt(item_seq_len + 1)
// The feed-forward half runs on the shared MFMA GEMM kernels exactly as in sasrec.hip; the mixer and its backward are
// row-group kernels (one 32-lane group per (sequence, position), lanes over channels).
#include "common.h"
#include "kernels.h"

namespace ur {

constexpr int CF_MAXV = 4;
static inline int cf_tpr(int d) {
  int d4 = d / 4, t = 4;
  while (t < d4 && t < 32) t <<= 1;
  return t;
}

struct CfLayout {
  long long off[3 + UR_MAX_LAYERS * 10];
  long long total;
};
// global: [0] position_embedding.weight [L,d]  [1] LayerNorm.weight  [2] LayerNorm.bias
// layer : [0] conv weight ([d,K] ConvFormer = nn.Conv1d [d,1,K];  [K,d] FAST = conv_weight [1,K,d])  [1] conv bias [d] (FAST: unused, 0)
//         [2] filterlayer.LayerNorm.weight [3] .bias  [4] dense_1.weight [I,d] [5] .bias  [6] dense_2.weight [d,I] [7] .bias
//         [8] intermediate.LayerNorm.weight [9] .bias
static CfLayout cf_layout(const UrConvFormerCfg& c) {
  CfLayout l;
  long long o = 0;
  int k = 0;
  const long long d = c.d, I = c.inner;
  auto put = [&](long long n) { l.off[k++] = o; o += n; };
  put((long long)c.L * d); put(d); put(d);
  for (int i = 0; i < c.n_layers; ++i) {
    put(d * c.conv_size); put(d); put(d); put(d); put(I * d); put(I); put(d * I); put(d); put(d); put(d);
  }
  l.total = o;
  return l;
}

// source position of tap k for output position l (-1: the tap reads a zero)
__device__ __forceinline__ int cf_src(int fast, int mode, int l, int k, int L, int K) {
  if (fast) {
    int j = l - k;
    return j < 0 ? j + L : j;
  }
  const int P = K - 1, i = l + k;
  if (i >= P) return i - P;
  return mode == 0 ? L - P + i : (mode == 1 ? L - 1 - i : -1);
}
__device__ __forceinline__ float4 cf_weight4(const float* __restrict__ w, int fast, int c4, int k, int K, int d) {
  if (fast) return *(const float4*)(w + (long long)k * d + c4 * 4);
  const float* p = w + (long long)c4 * 4 * K + k;
  return make_float4(p[0], p[K], p[2 * K], p[3 * K]);
}

// y1 = LN(mix(x) + x): one lane group per row (b,l)
template <int TPR>
__global__ __launch_bounds__(256) void cf_mix_ln_fwd_kernel(const float4* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float4* __restrict__ gamma,
                                                            const float4* __restrict__ beta, float eps, int B, int L, int K, int d4, int fast,
                                                            int mode, float scale, float4* __restrict__ y, float4* __restrict__ xhat,
                                                            float* __restrict__ rstd_out, DropSpec drop) {
  constexpr int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int M = B * L, d = d4 * 4;
  const float inv_d = 1.0f / (float)d;
  for (int row = blockIdx.x * groups + g; row < M; row += gridDim.x * groups) {
    const int b = row / L, l = row % L;
    float4 v[CF_MAXV];
#pragma unroll
    for (int q = 0; q < CF_MAXV; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {   // fixed tap order
      const int j = cf_src(fast, mode, l, k, L, K);
      if (j < 0) continue;
#pragma unroll
      for (int q = 0; q < CF_MAXV; ++q) {
        const int c = t + q * TPR;
        if (c < d4) {
          const float4 a = x[((long long)b * L + j) * d4 + c], ww = cf_weight4(w, fast, c, k, K, d);
          v[q].x = fmaf(ww.x, a.x, v[q].x); v[q].y = fmaf(ww.y, a.y, v[q].y); v[q].z = fmaf(ww.z, a.z, v[q].z); v[q].w = fmaf(ww.w, a.w, v[q].w);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < CF_MAXV; ++q) {
      const int c = t + q * TPR;
      if (c < d4) {
        const float4 r = x[(long long)row * d4 + c];
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!fast) bb = *(const float4*)(bias + c * 4);
        if (drop.thresh) {   // t = dropout(mix(x)) + x   (convformer.py:97, fastconvformer.py:58)
          float4 h = make_float4(v[q].x * scale + bb.x, v[q].y * scale + bb.y, v[q].z * scale + bb.z, v[q].w * scale + bb.w);
          h = drop4(h, mix32((unsigned)row ^ drop.key), (unsigned)(c * 4), drop);
          v[q] = make_float4(h.x + r.x, h.y + r.y, h.z + r.z, h.w + r.w);
        } else {
          v[q].x = v[q].x * scale + bb.x + r.x; v[q].y = v[q].y * scale + bb.y + r.y;
          v[q].z = v[q].z * scale + bb.z + r.z; v[q].w = v[q].w * scale + bb.w + r.w;
        }
        s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
      }
    }
    const float mean = group_sum<TPR>(s) * inv_d;
    float qq = 0.f;
#pragma unroll
    for (int q = 0; q < CF_MAXV; ++q) {
      const int c = t + q * TPR;
      if (c < d4) {
        v[q].x -= mean; v[q].y -= mean; v[q].z -= mean; v[q].w -= mean;
        qq += (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
      }
    }
    const float rstd = 1.0f / sqrtf(group_sum<TPR>(qq) * inv_d + eps);
#pragma unroll
    for (int q = 0; q < CF_MAXV; ++q) {
      const int c = t + q * TPR;
      if (c < d4) {
        const float4 gm = gamma[c], bt = beta[c];
        const float4 h = make_float4(v[q].x * rstd, v[q].y * rstd, v[q].z * rstd, v[q].w * rstd);
        xhat[(long long)row * d4 + c] = h;
        y[(long long)row * d4 + c] = make_float4(h.x * gm.x + bt.x, h.y * gm.y + bt.y, h.z * gm.z + bt.z, h.w * gm.w + bt.w);
      }
    }
    if (t == 0) rstd_out[row] = rstd;
  }
}

// dx[b,j] = dres[b,j] (residual) + scale * sum_{(l,k): src(l,k) = j} w[.,k] dt[b,l]        one lane group per row (b,j)
// (dt = gradient of the mixer output: the dropout-masked LayerNorm-input gradient; dres = the unmasked one; same buffer without dropout)
template <int TPR>
__global__ __launch_bounds__(256) void cf_mix_bwd_x_kernel(const float4* __restrict__ dt, const float4* __restrict__ dres,
                                                           const float* __restrict__ w, int B, int L, int K,
                                                           int d4, int fast, int mode, float scale, float4* __restrict__ dx) {
  constexpr int groups = 256 / TPR;
  const int g = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int M = B * L, d = d4 * 4, P = K - 1;
  for (int row = blockIdx.x * groups + g; row < M; row += gridDim.x * groups) {
    const int b = row / L, j = row % L;
    float4 acc[CF_MAXV];
#pragma unroll
    for (int q = 0; q < CF_MAXV; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the positions of row j inside the padded sequence: the body, and (circular / reflect) possibly the prefix
    const int i2 = fast ? -1 : (j >= L - P ? (mode == 0 ? j - (L - P) : (mode == 1 ? L - 1 - j : -1)) : -1);
    for (int k = 0; k < K; ++k) {
      int ls[2];
      if (fast) {
        ls[0] = j + k >= L ? j + k - L : j + k;
        ls[1] = -1;
      } else {
        ls[0] = j + P - k < L ? j + P - k : -1;      // j + P - k >= 0 always (k <= P)
        ls[1] = i2 >= k ? i2 - k : -1;               // i2 - k < L always
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int l = ls[u];
        if (l < 0) continue;
#pragma unroll
        for (int q = 0; q < CF_MAXV; ++q) {
          const int c = t + q * TPR;
          if (c < d4) {
            const float4 a = dt[((long long)b * L + l) * d4 + c], ww = cf_weight4(w, fast, c, k, K, d);
            acc[q].x = fmaf(ww.x, a.x, acc[q].x); acc[q].y = fmaf(ww.y, a.y, acc[q].y);
            acc[q].z = fmaf(ww.z, a.z, acc[q].z); acc[q].w = fmaf(ww.w, a.w, acc[q].w);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CF_MAXV; ++q) {
      const int c = t + q * TPR;
      if (c < d4) {
        const float4 r = dres[(long long)row * d4 + c];
        dx[(long long)row * d4 + c] = make_float4(acc[q].x * scale + r.x, acc[q].y * scale + r.y, acc[q].z * scale + r.z, acc[q].w * scale + r.w);
      }
    }
  }
}

// partial weight / bias gradients: block (k, s) sums dt[b,l,c] * x[b,src(l,k),c] over the sequences of split s (fixed order);
// thread = channel.  part_w[s][...] in the parameter's own layout ([d,K] or [K,d]), part_b[s][d] from the k == 0 blocks.
__global__ __launch_bounds__(256) void cf_mix_bwd_w_kernel(const float* __restrict__ dt, const float* __restrict__ x, int B, int L, int K, int d,
                                                           int fast, int mode, float scale, int seq_per_split, float* __restrict__ part_w,
                                                           float* __restrict__ part_b) {
  const int k = blockIdx.x, s = blockIdx.y;
  const int b0 = s * seq_per_split, b1 = min(B, b0 + seq_per_split);
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f, accb = 0.f;
    for (int b = b0; b < b1; ++b)
      for (int l = 0; l < L; ++l) {
        const float g = dt[((long long)b * L + l) * d + c];
        const int j = cf_src(fast, mode, l, k, L, K);
        if (j >= 0) acc = fmaf(g, x[((long long)b * L + j) * d + c], acc);
        accb += g;
      }
    part_w[(long long)s * d * K + (fast ? (long long)k * d + c : (long long)c * K + k)] = acc * scale;
    if (k == 0 && part_b) part_b[(long long)s * d + c] = accb;
  }
}

// user_emb: last position, or the decayed merge (convformer.py:66-71)
__global__ void cf_out_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ seq_len, int B, int L, int d, int merge,
                                  float decay, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * d) return;
  const long long b = i / d, c = i % d;
  if (!merge) {
    out[i] = x[(b * L + (L - 1)) * d + c];
    return;
  }
  float acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const float wl = L > 1 ? exp10f(decay * (1.0f - (float)l / (float)(L - 1))) : exp10f(decay);
    acc = fmaf(x[(b * L + l) * d + c], wl, acc);
  }
  out[i] = acc / sqrtf((float)(seq_len[b] + 1));
}
__global__ void cf_out_bwd_kernel(const float* __restrict__ g, const long long* __restrict__ seq_len, int B, int L, int d, int merge,
                                  float decay, float* __restrict__ gx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * L * d) return;
  const long long c = i % d, row = i / d, l = row % L, b = row / L;
  if (!merge) {
    gx[i] = (l == L - 1) ? g[b * d + c] : 0.f;
    return;
  }
  const float wl = L > 1 ? exp10f(decay * (1.0f - (float)l / (float)(L - 1))) : exp10f(decay);
  gx[i] = g[b * d + c] * wl / sqrtf((float)(seq_len[b] + 1));
}

struct CfLayerWs {
  float *y1, *y1hat, *rstd1, *h1, *y, *yhat, *rstd2, *w1T, *w2T;
};
struct CfWs {
  float *x0, *x0hat, *rstd0;
  CfLayerWs layer[UR_MAX_LAYERS];
  float *g_y, *g_t, *g_td, *g_a, *g_h1, *tn_ws, *ln_part, *mix_part;   // g_td: dropout-masked g_t (hidden dropout on)
  long long tn_floats, ln_floats, mix_floats, total;
};
constexpr int CF_W_SPLITS = 64;
static CfWs cf_carve(const UrConvFormerCfg& c, float* base) {
  CfWs w;
  long long o = 0;
  auto take = [&](long long n) {
    float* p = base ? base + o : nullptr;
    o += (n + 63) & ~63LL;
    return p;
  };
  const long long M = (long long)c.B * c.L, d = c.d, I = c.inner;
  w.x0 = take(M * d); w.x0hat = take(M * d); w.rstd0 = take(M);
  for (int i = 0; i < c.n_layers; ++i) {
    CfLayerWs& lw = w.layer[i];
    lw.y1 = take(M * d); lw.y1hat = take(M * d); lw.rstd1 = take(M); lw.h1 = take(M * I);
    lw.y = take(M * d); lw.yhat = take(M * d); lw.rstd2 = take(M); lw.w1T = take(I * d); lw.w2T = take(I * d);
  }
  w.g_y = take(M * d); w.g_t = take(M * d); w.g_a = take(M * d); w.g_h1 = take(M * I);
  w.g_td = c.p_hidden > 0.f ? take(M * d) : w.g_t;
  auto r64 = [](long long n) { return (n + 63) & ~63LL; };
  w.tn_floats = c.n_layers * (r64(gemm_tn_ws_floats((int)M, (int)d, (int)I)) + r64(gemm_tn_ws_floats((int)M, (int)I, (int)d)));
  w.tn_ws = take(w.tn_floats);
  w.ln_floats = (2LL * c.n_layers + 1) * LN_BWD_MAX_BLOCKS * 2 * d;
  w.ln_part = take(w.ln_floats);
  w.mix_floats = (long long)c.n_layers * CF_W_SPLITS * (d * c.conv_size + d);
  w.mix_part = take(w.mix_floats);
  w.total = o;
  return w;
}

// dropout sites (row id = token b*L + l): 0 = embedded input; layer i: 4(i+1)+2 mixer output, 4(i+1)+3 feed-forward output
static DropSpec cf_site(const UrConvFormerCfg& c, int layer, int site) {
  return drop_spec(c.p_hidden, c.drop_seed, c.drop_step, (unsigned)(site == 0 ? 0 : 4 * (layer + 1) + site));
}

static int cf_check(const UrConvFormerCfg* c) {
  UR_REQUIRE(c != nullptr, UR_ERR_ARG, "convformer: null cfg");
  UR_REQUIRE(c->B > 0 && c->L > 0 && (long long)c->B * c->L < (1LL << 31), UR_ERR_ARG, "convformer: B=%d L=%d", c->B, c->L);
  UR_REQUIRE(c->d > 0 && c->d % 4 == 0 && c->d <= 256, UR_ERR_UNSUPPORTED, "convformer: hidden size d=%d must be a multiple of 4 and <= 256", c->d);
  UR_REQUIRE(c->inner > 0 && c->inner % 4 == 0, UR_ERR_ARG, "convformer: inner_size=%d", c->inner);
  UR_REQUIRE(c->n_layers >= 1 && c->n_layers <= UR_MAX_LAYERS, UR_ERR_ARG, "convformer: n_layers=%d", c->n_layers);
  UR_REQUIRE(c->conv_size >= 1 && c->conv_size <= c->L, UR_ERR_ARG, "convformer: conv_size=%d must be in [1, max_seq_len]", c->conv_size);
  UR_REQUIRE(c->padding_mode >= 0 && c->padding_mode <= 2, UR_ERR_ARG, "convformer: padding_mode=%d", c->padding_mode);
  UR_REQUIRE(c->act >= UR_ACT_GELU && c->act <= UR_ACT_SIGMOID, UR_ERR_ARG, "convformer: act=%d", c->act);
  UR_REQUIRE(c->p_hidden >= 0.f && c->p_hidden < 1.f, UR_ERR_ARG, "convformer: hidden dropout %g not in [0, 1)", (double)c->p_hidden);
  return UR_OK;
}

}  // namespace ur

using namespace ur;

extern "C" int64_t ur_convformer_param_layout(const UrConvFormerCfg* cfg, int64_t* offsets_out) {
  int rc = cf_check(cfg);
  if (rc) return rc;
  const CfLayout l = cf_layout(*cfg);
  if (offsets_out)
    for (int i = 0; i < 3 + cfg->n_layers * 10; ++i) offsets_out[i] = l.off[i];
  return l.total;
}

extern "C" int64_t ur_convformer_workspace_bytes(const UrConvFormerCfg* cfg) {
  int rc = cf_check(cfg);
  if (rc) return rc;
  return cf_carve(*cfg, nullptr).total * (int64_t)sizeof(float);
}

#define CF_TPR_SWITCH(tpr, GO) \
  switch (tpr) {               \
    case 4: GO(4); break;      \
    case 8: GO(8); break;      \
    case 16: GO(16); break;    \
    default: GO(32); break;    \
  }

extern "C" int ur_convformer_fwd(const UrConvFormerCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                                 const int32_t* item_seq, const int64_t* seq_len, float* user_emb, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  int rc = cf_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(item_table && dense && item_seq && user_emb && ws && n_items > 0, UR_ERR_ARG, "ur_convformer_fwd: null pointer");
  UR_REQUIRE(!cfg->seq_merge || seq_len, UR_ERR_ARG, "ur_convformer_fwd: seq_merge needs item_seq_len");
  const UrConvFormerCfg& c = *cfg;
  hipStream_t st = as_stream(stream);
  const CfLayout lay = cf_layout(c);
  CfWs w = cf_carve(c, (float*)ws);
  const int M = c.B * c.L, d = c.d, I = c.inner, tpr = cf_tpr(d), groups = 256 / tpr;
  const float scale = c.fast ? 1.0f / sqrtf((float)c.L) : 1.0f;
  const DropSpec d_emb = cf_site(c, 0, 0);
  if ((rc = embed_ln_fwd(item_seq, item_table, dense + lay.off[0], dense + lay.off[1], dense + lay.off[2], c.eps, M, c.L, d, w.x0, w.x0hat,
                         w.rstd0, st, nullptr, nullptr, &d_emb, n_items)))
    return rc;
  const float* x = w.x0;
  for (int i = 0; i < c.n_layers; ++i) {
    const long long* o = lay.off + 3 + i * 10;
    CfLayerWs& lw = w.layer[i];
    {
      ProfScope ps(PC_ROWOPS, st, (double)M * d * 4.0 * (c.conv_size + 3));
      int blocks = cdiv(M, groups);
      if (blocks > 8192) blocks = 8192;
      const DropSpec d_mix = cf_site(c, i, 2);
#define GO(T) hipLaunchKernelGGL((cf_mix_ln_fwd_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float4*)x, dense + o[0], dense + o[1],      \
                                 (const float4*)(dense + o[2]), (const float4*)(dense + o[3]), c.eps, c.B, c.L, c.conv_size, d / 4, c.fast, \
                                 c.padding_mode, scale, (float4*)lw.y1, (float4*)lw.y1hat, lw.rstd1, d_mix)
      CF_TPR_SWITCH(tpr, GO)
#undef GO
      UR_LAUNCH_CHECK();
    }
    GemmArgs g{};
    g.A = lw.y1; g.lda = d; g.W = dense + o[4]; g.ldw = d; g.C = lw.h1; g.ldc = I; g.M = M; g.N = I; g.K = d; g.bias = dense + o[5];
    if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
    g = GemmArgs{};
    g.A = lw.h1; g.lda = I; g.W = dense + o[6]; g.ldw = I; g.C = lw.y; g.ldc = d; g.M = M; g.N = d; g.K = I; g.bias = dense + o[7]; g.act = c.act;
    g.aux = lw.y1; g.ldaux = d; g.gamma = dense + o[8]; g.beta = dense + o[9]; g.eps = c.eps; g.xhat = lw.yhat; g.rstd = lw.rstd2;
    g.drop = cf_site(c, i, 3);
    if ((rc = gemm_nt(g, PRO_ACT, EPI_BIAS_RES_LN, st))) return rc;
    x = lw.y;
  }
  hipLaunchKernelGGL(cf_out_fwd_kernel, dim3(cdiv((long long)c.B * d, 256)), dim3(256), 0, st, x, (const long long*)seq_len, c.B, c.L, d,
                     c.seq_merge, c.seq_decay, user_emb);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_convformer_bwd(const UrConvFormerCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                                 const int32_t* item_seq, const int64_t* seq_len, const float* d_user_emb, void* ws, float* dense_grad,
                                 float* d_emb_rows, void* stream) {
  UR_TRACE_SCOPE();
  int rc = cf_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(dense && item_seq && d_user_emb && ws && dense_grad && d_emb_rows, UR_ERR_ARG, "ur_convformer_bwd: null pointer");
  UR_REQUIRE(!cfg->seq_merge || seq_len, UR_ERR_ARG, "ur_convformer_bwd: seq_merge needs item_seq_len");
  (void)item_table; (void)n_items;
  const UrConvFormerCfg& c = *cfg;
  hipStream_t st = as_stream(stream);
  const CfLayout lay = cf_layout(c);
  CfWs w = cf_carve(c, (float*)ws);
  const int M = c.B * c.L, d = c.d, I = c.inner, K = c.conv_size, tpr = cf_tpr(d), groups = 256 / tpr;
  const float scale = c.fast ? 1.0f / sqrtf((float)c.L) : 1.0f;
  UR_HIP(hipMemsetAsync(dense_grad, 0, lay.total * sizeof(float), st));
  ReduceBatch rb;
  float *tn_cur = w.tn_ws, *ln_cur = w.ln_part, *mix_cur = w.mix_part;
  auto tn_take = [&](int T_, int R_, int C_) {
    float* p = tn_cur;
    tn_cur += (gemm_tn_ws_floats(T_, R_, C_) + 63) & ~63LL;
    return p;
  };
  auto ln_take = [&]() {
    float* p = ln_cur;
    ln_cur += (long long)LN_BWD_MAX_BLOCKS * 2 * d;
    return p;
  };
  hipLaunchKernelGGL(cf_out_bwd_kernel, dim3(cdiv((long long)M * d, 256)), dim3(256), 0, st, d_user_emb, (const long long*)seq_len, c.B, c.L, d,
                     c.seq_merge, c.seq_decay, w.g_y);
  UR_LAUNCH_CHECK();
  {
    TransposeBatch tb;
    for (int i = 0; i < c.n_layers; ++i) {
      const long long* o = lay.off + 3 + i * 10;
      tb.add(dense + o[4], I, d, w.layer[i].w1T);
      tb.add(dense + o[6], d, I, w.layer[i].w2T);
    }
    if ((rc = transpose_batch(tb, st))) return rc;
  }
  for (int i = c.n_layers - 1; i >= 0; --i) {
    const long long* o = lay.off + 3 + i * 10;
    CfLayerWs& lw = w.layer[i];
    float* G = dense_grad;
    const float* x_in = (i == 0) ? w.x0 : w.layer[i - 1].y;
    // ---- feed-forward block (as sasrec.hip)
    const DropSpec d_ffn = cf_site(c, i, 3), d_mix = cf_site(c, i, 2);
    if ((rc = ln_bwd(w.g_y, lw.yhat, lw.rstd2, dense + o[8], nullptr, nullptr, M, d, w.g_t, G + o[8], G + o[9], ln_take(), st, &rb, nullptr,
                     nullptr, nullptr, &d_ffn, w.g_td)))
      return rc;
    if ((rc = gemm_tn(w.g_td, d, lw.h1, I, M, d, I, 1, c.act, G + o[6], I, G + o[7], tn_take(M, d, I), st, &rb))) return rc;
    GemmArgs g{};
    g.A = w.g_td; g.lda = d; g.W = lw.w2T; g.ldw = d; g.C = w.g_h1; g.ldc = I; g.M = M; g.N = I; g.K = d; g.aux = lw.h1; g.ldaux = I; g.act = c.act;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_MUL_DACT, st))) return rc;
    if ((rc = gemm_tn(w.g_h1, I, lw.y1, d, M, I, d, 0, 0, G + o[4], d, G + o[5], tn_take(M, I, d), st, &rb))) return rc;
    g = GemmArgs{};
    g.A = w.g_h1; g.lda = I; g.W = lw.w1T; g.ldw = I; g.C = w.g_a; g.ldc = d; g.M = M; g.N = d; g.K = I; g.aux = w.g_t; g.ldaux = d;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD, st))) return rc;
    // ---- mixer block: LayerNorm backward -> d t, then the three gradients of t = mix(x) + x
    if ((rc = ln_bwd(w.g_a, lw.y1hat, lw.rstd1, dense + o[2], nullptr, nullptr, M, d, w.g_t, G + o[2], G + o[3], ln_take(), st, &rb, nullptr,
                     nullptr, nullptr, &d_mix, w.g_td)))
      return rc;
    {
      ProfScope ps(PC_ROWOPS, st, (double)M * d * 4.0 * (2 * K + 2));
      const int sps = cdiv(c.B, CF_W_SPLITS), S = cdiv(c.B, sps);
      float* part_w = mix_cur;
      float* part_b = mix_cur + (long long)S * d * K;
      mix_cur += (long long)CF_W_SPLITS * (d * K + d);
      hipLaunchKernelGGL(cf_mix_bwd_w_kernel, dim3(K, S), dim3(d < 256 ? ((d + 63) / 64) * 64 : 256), 0, st, w.g_td, x_in, c.B, c.L, K, d, c.fast,
                         c.padding_mode, scale, sps, part_w, c.fast ? nullptr : part_b);
      UR_LAUNCH_CHECK();
      if (rb.full(2) && (rc = reduce_batch(rb, st))) return rc;
      rb.add(part_w, (long long)d * K, S, (long long)d * K, d * K, G + o[0], d * K);
      if (!c.fast) rb.add(part_b, d, S, d, d, G + o[1], d);
      int blocks = cdiv(M, groups);
      if (blocks > 8192) blocks = 8192;
#define GO(T) hipLaunchKernelGGL((cf_mix_bwd_x_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float4*)w.g_td, (const float4*)w.g_t, dense + o[0], c.B, c.L, K, d / 4, \
                                 c.fast, c.padding_mode, scale, (float4*)w.g_y)
      CF_TPR_SWITCH(tpr, GO)
#undef GO
      UR_LAUNCH_CHECK();
    }
  }
  const DropSpec d_emb = cf_site(c, 0, 0);
  if ((rc = ln_bwd(w.g_y, w.x0hat, w.rstd0, dense + lay.off[1], nullptr, nullptr, M, d, d_emb_rows, dense_grad + lay.off[1],
                   dense_grad + lay.off[2], ln_take(), st, &rb, nullptr, nullptr, &d_emb)))
    return rc;
  if (rb.full(1) && (rc = reduce_batch(rb, st))) return rc;
  rb.add(d_emb_rows, (long long)c.L * d, c.B, (long long)c.L * d, c.L * d, dense_grad + lay.off[0], c.L * d);   // dP[l] = sum_b dx[b,l]
  UR_REQUIRE(tn_cur <= w.tn_ws + w.tn_floats && ln_cur <= w.ln_part + w.ln_floats && mix_cur <= w.mix_part + w.mix_floats, UR_ERR_ARG,
             "ur_convformer_bwd: partial-sum workspace overrun");
  return reduce_batch(rb, st);
}
