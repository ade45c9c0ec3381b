// SASRec user encoder: forward / backward orchestration over the kernels of rowops / gemm / attention.
// Stateless: every call receives the flat parameter buffer, the caller-owned workspace and a stream.
//
// Forward per layer (reference: unirec/model/modules.py:284-316, 347-355, 379-382):
//   qkv = x Wqkv^T + bqkv                      gemm_nt  EPI_BIAS            (3 nn.Linear fused: weights contiguous)
//   ctx = softmax(QK^T/sqrt(hd) + mask) V      attn_fwd
//   a   = LN(ctx Wo^T + bo + x)                gemm_nt  EPI_BIAS_RES_LN
//   h1  = a W1^T + b1                          gemm_nt  EPI_BIAS            (pre-activation kept for backward)
//   y   = LN(act(h1) W2^T + b2 + a)            gemm_nt  PRO_ACT + EPI_BIAS_RES_LN
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace ur {

struct Layout {
  long long off[UR_SASREC_N_GLOBAL + UR_MAX_LAYERS * UR_SASREC_N_PER_LAYER];
  long long total;
};

static Layout make_layout(const UrSasrecCfg& c) {
  Layout l;
  long long o = 0;
  const long long d = c.d, I = c.inner;
  int k = 0;
  auto put = [&](long long n) { l.off[k++] = o; o += n; };
  put((long long)(c.L + 1) * d);  // position_embedding.weight
  put(d);                          // LayerNorm.weight
  put(d);                          // LayerNorm.bias
  for (int i = 0; i < c.n_layers; ++i) {
    put(d * d); put(d * d); put(d * d);  // q,k,v weights (contiguous => Wqkv)
    put(d); put(d); put(d);              // q,k,v biases
    put(d * d); put(d);                  // dense
    put(d); put(d);                      // attn LN
    put(I * d); put(I);                  // dense_1
    put(d * I); put(d);                  // dense_2
    put(d); put(d);                      // ffn LN
  }
  l.total = o;
  return l;
}

struct LayerP {
  const float *wqkv, *bqkv, *wo, *bo, *g1, *b1ln, *w1, *b1, *w2, *b2, *g2, *b2ln;
};
static LayerP layer_ptrs(const float* base, const Layout& l, int i) {
  const long long* o = l.off + UR_SASREC_N_GLOBAL + i * UR_SASREC_N_PER_LAYER;
  LayerP p;
  p.wqkv = base + o[0]; p.bqkv = base + o[3]; p.wo = base + o[6]; p.bo = base + o[7];
  p.g1 = base + o[8]; p.b1ln = base + o[9]; p.w1 = base + o[10]; p.b1 = base + o[11];
  p.w2 = base + o[12]; p.b2 = base + o[13]; p.g2 = base + o[14]; p.b2ln = base + o[15];
  return p;
}

// ---- workspace carving (float units, every region 64-float aligned)
struct LayerWs {
  float *qkv, *lse, *ctx, *a, *ahat, *rstd1, *h1, *y, *yhat, *rstd2;
  float *wqkvT, *woT, *w1T, *w2T;
  // split-bf16 copies streamed by the row-chain kernels (kernels.h: TransposeBatch::add_split; 3/2 floats per weight): of the K-major
  // (transposed) weights for the forward chains, of the weights as stored for the backward chains.  nullptr when the shape has no chains
  float *s_wqkvT, *s_woT, *s_w1T, *s_w2T, *s_wqkv, *s_wo, *s_w1, *s_w2;
  float *g_tf, *g_ta, *g_h1, *g_qkv;   // backward: LN-backward outputs (FFN / attention block), d h1, d qkv
  float *g_tfd, *g_tad;                // hidden dropout on: dropout-masked g_tf / g_ta (what the block's GEMMs consume)
};
struct Ws {
  float *x0, *x0hat, *rstd0;
  LayerWs layer[UR_MAX_LAYERS];
  float *g_y, *g_a, *g_ctx, *tn_ws, *ln_part, *attn_ws;
  float* split_part;                    // chain_ffn_fwd_split / _bwd_split: per-(row block, inner chunk) partial tiles
  float* chain_part;                    // row-chain kernels: per-workgroup LayerNorm-affine partial sums (see chain_part_of)
  long long chain_blocks;               // upper bound of their workgroup count: cdiv(B*L, 32)
  float *q_last, *dq_last, *lse_last;   // last-row specialisation of the final layer ([B,d] each)
  float *x_last, *t_last;               // compact mode: gathered last rows of the layer input / their gradient
  int *tok_full, *seq_base, *seq_pad, *last_row, *m_valid;   // compact mode: row maps (see compact_plan_kernel)
  long long total_floats, tn_floats, ln_floats;
};

static Ws carve(const UrSasrecCfg& c, float* base) {
  Ws w;
  long long o = 0;
  auto take = [&](long long n) {
    float* p = base ? base + o : nullptr;
    o += (n + 63) & ~63LL;
    return p;
  };
  const long long M = (long long)c.B * c.L, d = c.d, I = c.inner;
  w.x0 = take(M * d); w.x0hat = take(M * d); w.rstd0 = take(M);
  for (int i = 0; i < c.n_layers; ++i) {
    LayerWs& lw = w.layer[i];
    lw.qkv = take(M * 3 * d); lw.lse = take(attn_lse_floats(c.B, c.n_heads, c.L)); lw.ctx = take(M * d);
    lw.a = take(M * d); lw.ahat = take(M * d); lw.rstd1 = take(M); lw.h1 = take(M * I);
    lw.y = take(M * d); lw.yhat = take(M * d); lw.rstd2 = take(M);
    lw.wqkvT = take(3 * d * d); lw.woT = take(d * d); lw.w1T = take(I * d); lw.w2T = take(I * d);
    lw.s_wqkvT = lw.s_woT = lw.s_w1T = lw.s_w2T = lw.s_wqkv = lw.s_wo = lw.s_w1 = lw.s_w2 = nullptr;
    if (chain_shape_ok(c.d, c.inner)) {
      lw.s_wqkvT = take(3 * d * d * 3 / 2); lw.s_woT = take(d * d * 3 / 2); lw.s_w1T = take(I * d * 3 / 2); lw.s_w2T = take(I * d * 3 / 2);
      lw.s_wqkv = take(3 * d * d * 3 / 2); lw.s_wo = take(d * d * 3 / 2); lw.s_w1 = take(I * d * 3 / 2); lw.s_w2 = take(I * d * 3 / 2);
    }
    // backward scratch that the weight-gradient GEMMs read: per layer, written once per backward pass, so those GEMMs
    // can run on the side stream without write-after-read hazards against the activation-gradient chain
    lw.g_tf = take(M * d); lw.g_ta = take(M * d); lw.g_h1 = take(M * I); lw.g_qkv = take(M * 3 * d);
    lw.g_tfd = lw.g_tf; lw.g_tad = lw.g_ta;
    if (c.p_hidden > 0.f) { lw.g_tfd = take(M * d); lw.g_tad = take(M * d); }
  }
  w.g_y = take(M * d); w.g_a = take(M * d); w.g_ctx = take(M * d);
  // split-reduction partials: every weight-gradient GEMM / LayerNorm backward of one backward pass keeps its own region
  // (they are all reduced by ONE launch at the end of ur_sasrec_bwd)
  const int T = (int)M;
  auto r64 = [](long long n) { return (n + 63) & ~63LL; };
  w.tn_floats = c.n_layers * (r64(gemm_tn_ws_floats(T, c.d, c.inner)) + r64(gemm_tn_ws_floats(T, c.inner, c.d)) +
                              r64(gemm_tn_ws_floats(T, 3 * c.d, c.d)) + 2 * r64(gemm_tn_ws_floats(T, c.d, c.d)));
  w.tn_ws = take(w.tn_floats);
  w.ln_floats = (2LL * c.n_layers + 1) * LN_BWD_MAX_BLOCKS * 2 * d;
  w.ln_part = take(w.ln_floats);
  w.attn_ws = take(attn_bwd_ws_floats(c.B, c.n_heads, c.L));
  w.chain_blocks = (M + 31) / 32;
  w.chain_part = take((4LL * c.n_layers + 2) * w.chain_blocks * d);
  w.split_part = take(2 * chain_split_part_floats(c.B, d, c.inner));   // last-row chain, inner split over workgroups: forward + backward partials
  w.q_last = take((long long)c.B * d); w.dq_last = take((long long)c.B * d);
  w.lse_last = take((long long)c.B * c.n_heads);
  w.x_last = take((long long)c.B * d); w.t_last = take((long long)c.B * d);
  w.tok_full = (int*)take(M); w.seq_base = (int*)take(c.B); w.seq_pad = (int*)take(c.B); w.last_row = (int*)take(c.B);
  w.m_valid = (int*)take(64);
  w.total_floats = o;
  return w;
}

// The row-chain kernels run their products in the cfg's arithmetic: split bf16 (six piece products, fp32-equivalent) whenever
// mfma_arith names a split form, the exact fp32-input MFMA at mfma_arith = 0 (test hook chain_split=0: exact whatever the cfg says).
static bool chains_split(const UrSasrecCfg& c) {
  const int base = c.mfma_arith & 0xFF;
  return (base == 6 || base == 9) && chain_shape_ok(c.d, c.inner) && ur_test_hook("chain_split", 1) != 0;
}

static int check_cfg(const UrSasrecCfg* c) {
  UR_REQUIRE(c != nullptr, UR_ERR_ARG, "sasrec: null cfg");
  UR_REQUIRE(c->B > 0 && c->L > 0, UR_ERR_ARG, "sasrec: B=%d L=%d", c->B, c->L);
  UR_REQUIRE(c->d > 0 && c->d % 4 == 0 && c->d <= 256, UR_ERR_UNSUPPORTED, "sasrec: hidden size d=%d must be a multiple of 4 and <= 256", c->d);
  UR_REQUIRE(c->inner > 0 && c->inner % 4 == 0, UR_ERR_ARG, "sasrec: inner_size=%d must be a multiple of 4", c->inner);
  UR_REQUIRE(c->n_heads > 0 && c->d % c->n_heads == 0, UR_ERR_ARG, "sasrec: d=%d not divisible by n_heads=%d", c->d, c->n_heads);
  UR_REQUIRE(c->n_layers >= 1 && c->n_layers <= UR_MAX_LAYERS, UR_ERR_ARG, "sasrec: n_layers=%d", c->n_layers);
  UR_REQUIRE(c->act >= UR_ACT_GELU && c->act <= UR_ACT_SIGMOID, UR_ERR_ARG, "sasrec: act=%d", c->act);
  UR_REQUIRE((long long)c->B * c->L < (1LL << 31), UR_ERR_ARG, "sasrec: B*L too large");
  UR_REQUIRE(c->p_hidden >= 0.f && c->p_hidden < 1.f && c->p_attn >= 0.f && c->p_attn < 1.f, UR_ERR_ARG,
             "sasrec: dropout probabilities must be in [0, 1): hidden %g attn %g", (double)c->p_hidden, (double)c->p_attn);
  UR_REQUIRE(c->p_attn == 0.f || (long long)c->B * c->n_heads * c->L < (1LL << 31), UR_ERR_ARG, "sasrec: B*n_heads*L too large for attention dropout");
  return UR_OK;
}

// dropout sites of one pass.  Row ids are TOKEN ids b*L + l whatever the row layout (padded, compact, last rows only), so
// the mask does not depend on skip_padding / last_only.
enum { DROP_SITE_EMBED = 0, DROP_SITE_ATTN = 1, DROP_SITE_OUT = 2, DROP_SITE_FFN = 3 };
static DropSpec site_spec(const UrSasrecCfg& c, int layer, int site, const int* rows, int mul = 1, int add = 0) {
  const float p = site == DROP_SITE_ATTN ? c.p_attn : c.p_hidden;
  DropSpec s = drop_spec(p, c.drop_seed, c.drop_step, (unsigned)(site == DROP_SITE_EMBED ? 0 : 4 * (layer + 1) + site));
  s.rows = rows; s.mul = mul; s.add = add;
  return s;
}

// dst[b,:] = src[b, L-1, :]
__global__ void take_last_kernel(const float* __restrict__ src, int B, int L, int d, float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * d) return;
  const long long b = i / d, c = i % d;
  dst[i] = src[(b * L + (L - 1)) * d + c];
}
// dst[b,l,:] = (l == L-1) ? src[b,:] : 0
__global__ void put_last_kernel(const float* __restrict__ src, int B, int L, int d, float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * L * d) return;
  const long long c = i % d, row = i / d, l = row % L, b = row / L;
  dst[i] = (l == L - 1) ? src[b * d + c] : 0.f;
}

// ---- padding skipped ("compact" token rows).  Sequences are left-padded, so the tokens of sequence b that can reach
// the loss are positions pad_b .. L-1 (pad_b = first non-zero position; an all-padding sequence keeps all L positions: its
// uniform attention over the padding rows IS what the reference computes).  Every token-parallel kernel of the layer
// stack then runs over M' = sum_b (L - pad_b) compact rows instead of B*L:
//   tok_full[r]  = b*L + l of compact row r          seq_base[b] = (first compact row of b) - pad_b  (row of (b,l) = seq_base[b] + l)
//   seq_pad[b]   = pad_b                             last_row[b] = compact row of (b, L-1)          m_valid[0] = M'
// M' stays on the device: grids are sized for B*L and surplus workgroups exit.
// Workgroup k writes the maps of sequences [k * CP_SEQS, (k + 1) * CP_SEQS); the compact row of its first sequence is the sum of the
// lengths of every sequence before it, which it computes ITSELF from their rows (the whole id matrix is ~100 KB and sits in L2:
// reading it eight times costs less than a second launch or a spin on another workgroup's result).  First item of a sequence: a
// wave per sequence, lane = position (coalesced row reads, ballot + count-trailing-zeros), sixteen sequences in flight per wave.
// (Round 1-2a: ONE workgroup walked all B sequences and then wrote the token map, 32 sequences per wave in turn: 14 us at the
// head of every step.)
constexpr int CP_SEQS = 64;
__global__ __launch_bounds__(1024) void compact_plan_kernel(const int* __restrict__ seq, int B, int L, int* __restrict__ tok_full,
                                                            int* __restrict__ seq_base, int* __restrict__ seq_pad,
                                                            int* __restrict__ last_row, int* __restrict__ m_valid) {
  __shared__ int s_pad[CP_SEQS];   // first non-zero position of this workgroup's sequences (0 if none: an all-padding sequence keeps every position)
  __shared__ int s_len[CP_SEQS];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b0 = blockIdx.x * CP_SEQS, b1 = min(B, b0 + CP_SEQS);
  const bool tail = b1 == B;       // the last workgroup also knows the total
  // ---- lengths of the sequences [0, b1): summed for b < b0 (per wave, then over the waves), kept for b0 <= b < b1
  int before = 0;
  constexpr int U = 16;
  for (int s0 = wv * U; s0 < b1; s0 += 16 * U) {
    int first[U];
#pragma unroll
    for (int u = 0; u < U; ++u) first[u] = L;
    for (int l0 = 0; l0 < L; l0 += 64) {
      int v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = seq[(long long)min(s0 + u, b1 - 1) * L + min(l0 + lane, L - 1)];   // clamped, unconditional: the U loads issue back to back
      const bool lin = l0 + lane < L;
      bool all_found = true;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned long long m = __ballot(lin && v[u] > 0);
        if (first[u] == L && m) first[u] = l0 + (int)__builtin_ctzll(m);
        all_found &= first[u] < L;
      }
      if (all_found) break;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sb = s0 + u;
      const int pad = first[u] == L ? 0 : first[u];
      if (sb < b0) before += L - pad;                                   // (wave-uniform values: every lane holds the same sum)
      else if (sb < b1 && lane == 0) { s_pad[sb - b0] = pad; s_len[sb - b0] = L - pad; }
    }
  }
  if (lane == 0) wsum[wv] = before;
  __syncthreads();
  int base0 = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) base0 += wsum[k];
  // ---- this workgroup's sequences: exclusive scan of their lengths (wave 0), then the maps
  const int nb = b1 - b0;
  __shared__ int s_base[CP_SEQS];
  if (wv == 0) {
    const int len = lane < nb ? s_len[lane] : 0;
    int inc = len;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(inc, off, 64);
      if (lane >= off) inc += v;
    }
    const int base = base0 + inc - len;
    s_base[lane] = base;
    if (lane < nb) {
      const int pad = s_pad[lane];
      seq_base[b0 + lane] = base - pad;
      seq_pad[b0 + lane] = pad;
      last_row[b0 + lane] = base + len - 1;
    }
    if (tail && lane == 63) m_valid[0] = base0 + inc;
  }
  __syncthreads();
  for (int sb = wv; sb < nb; sb += 16) {   // token map: a wave per sequence, coalesced stores
    const int pd = s_pad[sb], bs = s_base[sb] - pd;
    for (int l = pd + lane; l < L; l += 64) tok_full[bs + l] = (b0 + sb) * L + l;
  }
}
// dst[b,:] = src[idx[b],:]
__global__ void gather_rows_idx_kernel(const float* __restrict__ src, const int* __restrict__ idx, int B, int d, float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * d) return;
  const long long b = i / d, c = i % d;
  dst[i] = src[(long long)idx[b] * d + c];
}
// dst[idx[b],:] += src[b,:]     (idx injective)
__global__ void scatter_add_rows_idx_kernel(const float* __restrict__ src, const int* __restrict__ idx, int B, int d, float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * d) return;
  const long long b = i / d, c = i % d;
  dst[(long long)idx[b] * d + c] += src[i];
}
// dst[idx[b],:] = src[b,:]
__global__ void put_rows_idx_kernel(const float* __restrict__ src, const int* __restrict__ idx, int B, int d, float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * d) return;
  const long long b = i / d, c = i % d;
  dst[(long long)idx[b] * d + c] = src[i];
}


}  // namespace ur

using namespace ur;

extern "C" int64_t ur_sasrec_param_layout(const UrSasrecCfg* cfg, int64_t* offsets_out) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  Layout l = make_layout(*cfg);
  if (offsets_out)
    for (int i = 0; i < UR_SASREC_N_GLOBAL + cfg->n_layers * UR_SASREC_N_PER_LAYER; ++i) offsets_out[i] = l.off[i];
  return l.total;
}

extern "C" int64_t ur_sasrec_workspace_bytes(const UrSasrecCfg* cfg) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  return carve(*cfg, nullptr).total_floats * (int64_t)sizeof(float);
}


// ---- side stream for the weight-gradient GEMMs.  dW / db are consumed only by the optimizer, so they do not belong
// on the critical path dY -> dX: each gemm_tn is forked onto a second HIP stream as soon as its input gradient exists
// (event on the main stream), and joined before the final reduce_batch.  Fills the CUs that the latency-bound small
// kernels of the chain (last-row layer, LayerNorm backward, attention) leave idle.  UR_SASREC_SIDE=0 disables it.
namespace {
struct SideCtx {
  hipStream_t stream = nullptr;
  hipEvent_t ev[24];
  hipEvent_t done = nullptr, main_done = nullptr;
  bool join_pending = false;   // ur_sasrec_bwd_deferred left reductions running: `done` marks their end
  bool late_join = false;      // ur_sasrec_side_publish: the next ur_sasrec_fwd joins `done` itself, after its first launch
  bool ok = false;
};
// test aid (UR_TEST hook side_delay_us): a kernel that spins for that long on the side stream -- it widens every window in which the main stream
// could touch what the side stream has not finished with (tools/race_runs.sh, tests/test_fallback_paths_gpu.py)
__global__ void side_delay_kernel(long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}
int g_side_enabled = 1;   // runtime switch (ur_sasrec_set_side_stream)
SideCtx* side_ctx(bool even_if_disabled = false) {
  if (!g_side_enabled && !even_if_disabled) return nullptr;
  // one side stream (and set of events) per CONTEXT of the calling thread (common.h: g_ctx_id): the rank threads of the in-process loopback
  // transport each have their own; every other caller is context 0, as before
  static SideCtx* ctxs[UR_MAX_CTX] = {};
  static bool made[UR_MAX_CTX] = {};
  const int id = (g_ctx_id >= 0 && g_ctx_id < UR_MAX_CTX) ? g_ctx_id : 0;
  if (made[id]) return ctxs[id];
  made[id] = true;
  ctxs[id] = []() -> SideCtx* {
    const char* e = getenv("UR_SASREC_SIDE");
    if (e && atoi(e) == 0) return nullptr;
    SideCtx* c = new SideCtx();
    // (test hook side_prio = -1 / 1: the side stream's queue at high / low priority -- measured +- 0, profiles/HISTORY.md)
    const int prio = ur_test_hook("side_prio", 0);
    if ((prio ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return nullptr;
    // Events that order two streams of ONE device need no system-scope fence: a plain event makes the recording stream write back
    // and invalidate its caches for the host and for other devices, a 6-7 us bubble in front of the next kernel of the main stream at
    // every fork (measured: five per backward pass).
    const unsigned evf = hipEventDisableTiming | (unsigned)hipEventDisableSystemFence;
    for (auto& ev : c->ev)
      if (hipEventCreateWithFlags(&ev, evf) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c->done, evf) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c->main_done, evf) != hipSuccess) return nullptr;
    c->ok = true;
    return c;
  }();
  return ctxs[id];
}
}  // namespace

// Which forward pass a workspace holds (host side, per thread): ur_sasrec_bwd reads the K-major weight copies, row maps and activations the
// LAST ur_sasrec_fwd on that workspace left there.  A backward pass on a workspace whose last forward pass saw other weights, another
// id matrix or another shape would compute with stale or foreign state without any error: it is refused instead.
namespace {
struct FwdStamp { const void* ws; const void* dense; const void* seq; int B, L, d, n_layers; bool wsp; };   // wsp: the split-bf16 weight copies were made (chains_split)
thread_local FwdStamp g_fwd_stamps[8] = {};
thread_local int g_fwd_next = 0;
void stamp_forward(const void* ws, const void* dense, const void* seq, const UrSasrecCfg& c) {
  for (auto& e : g_fwd_stamps)
    if (e.ws == ws) { e = FwdStamp{ws, dense, seq, c.B, c.L, c.d, c.n_layers, chains_split(c)}; return; }
  g_fwd_stamps[g_fwd_next] = FwdStamp{ws, dense, seq, c.B, c.L, c.d, c.n_layers, chains_split(c)};
  g_fwd_next = (g_fwd_next + 1) % 8;
}
// -> 0 ok, 1 mismatch, -1 unknown workspace (stamped by another thread, or more than 8 workspaces ago: not checked)
int check_forward(const void* ws, const void* dense, const void* seq, const UrSasrecCfg& c) {
  for (const auto& e : g_fwd_stamps)
    if (e.ws == ws) return (e.dense == dense && e.seq == seq && e.B == c.B && e.L == c.L && e.d == c.d && e.n_layers == c.n_layers && e.wsp == chains_split(c)) ? 0 : 1;
  return -1;
}
}  // namespace

extern "C" int ur_sasrec_fwd(const UrSasrecCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                             const int32_t* item_seq, float* user_emb, void* ws, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_cfg(cfg);
  if (rc) return rc;
  UR_REQUIRE(item_table && dense && item_seq && user_emb && ws, UR_ERR_ARG, "ur_sasrec_fwd: null pointer");
  UR_REQUIRE(n_items > 0, UR_ERR_ARG, "ur_sasrec_fwd: n_items=%lld", (long long)n_items);
  const UrSasrecCfg& c = *cfg;
  stamp_forward(ws, dense, item_seq, c);
  const ArithScope arith_scope(c.mfma_arith);   // (the one-product-per-launch GEMMs of this pass: gemm.hip dispatch_tile)
  hipStream_t st = as_stream(stream);
  const Layout lay = make_layout(c);
  Ws w = carve(c, (float*)ws);
  const int M = c.B * c.L, d = c.d, I = c.inner;
  const float* pos = c.use_pos ? dense + lay.off[0] : nullptr;
  // compact mode: the padded prefix of every sequence has no rows at all (see compact_plan_kernel)
  const bool compact = c.skip_padding && attn_compact_supported(c.L, d, c.n_heads);
  const int* mv = compact ? w.m_valid : nullptr;
  const int* sbase = compact ? w.seq_base : nullptr;
  const int* spad = compact ? w.seq_pad : nullptr;
  if (compact) {
    hipLaunchKernelGGL(compact_plan_kernel, dim3(cdiv(c.B, CP_SEQS)), dim3(1024), 0, st, item_seq, c.B, c.L, w.tok_full, w.seq_base, w.seq_pad,
                       w.last_row, w.m_valid);
    UR_LAUNCH_CHECK();
  }
  if (SideCtx* sc = side_ctx(true); sc && sc->join_pending && sc->late_join) {   // (ur_sasrec_side_publish: the weights below are being updated on the side stream)
    UR_HIP(hipStreamWaitEvent(st, sc->done, 0));
    sc->join_pending = false;
    sc->late_join = false;
  }
  const bool wsp = chains_split(c);
  {   // K-major copies of every layer's weights, one launch: the forward chain kernels stream THEM (coalesced B operand, rowchain.hip);
      // the backward's unfused GEMMs (gemm_nt: C = A W^T) read the same copies -- the weights do not change between the two passes
    TransposeBatch tb;
    for (int i = 0; i < c.n_layers; ++i) {
      const LayerP p = layer_ptrs(dense, lay, i);
      LayerWs& lw = w.layer[i];
      if (tb.n + 12 > TransposeBatch::MAX) {
        if ((rc = transpose_batch(tb, st))) return rc;
        tb.n = 0;
      }
      tb.add(p.wqkv, 3 * d, d, lw.wqkvT); tb.add(p.wo, d, d, lw.woT); tb.add(p.w1, I, d, lw.w1T); tb.add(p.w2, d, I, lw.w2T);
      if (wsp) {   // ... and the split-bf16 copies both passes' chain kernels stream
        tb.add_split(p.wqkv, 3 * d, d, lw.s_wqkvT, false); tb.add_split(p.wo, d, d, lw.s_woT, false);
        tb.add_split(p.w1, I, d, lw.s_w1T, false); tb.add_split(p.w2, d, I, lw.s_w2T, false);
        tb.add_split(p.wqkv, 3 * d, d, lw.s_wqkv, true); tb.add_split(p.wo, d, d, lw.s_wo, true);
        tb.add_split(p.w1, I, d, lw.s_w1, true); tb.add_split(p.w2, d, I, lw.s_w2, true);
      }
    }
    if ((rc = transpose_batch(tb, st))) return rc;
  }
  const int* tokmap = compact ? w.tok_full : nullptr;   // buffer row -> token id (identity when not compact)
  const DropSpec d_emb = site_spec(c, 0, DROP_SITE_EMBED, nullptr);
  const bool chain = chain_supported(d, I, CHAIN_FWD);   // out-projection + LN + feed-forward + LN (+ next projection) as ONE launch per layer
  const bool chain_last = chain_supported(d, I, CHAIN_LAST);   // ... and for the B last rows of the last-row layer
  bool proj_done = false;                     // this layer's K,V (Q,K,V) rows were written by the previous layer's chain kernel (or the input block's)
  if (chain_supported(d, I, CHAIN_EMBED)) {   // lookup + position + LayerNorm + the first layer's projection as one launch
    const LayerP p0 = layer_ptrs(dense, lay, 0);
    const int skip_q = (c.last_only && c.n_layers == 1) ? 1 : 0;   // a last-row first layer projects K, V only here
    ChainEmbedArgs ce{};
    ce.seq = item_seq; ce.n_rows = n_items; ce.table = item_table; ce.pos = pos; ce.g0 = dense + lay.off[1]; ce.b0ln = dense + lay.off[2]; ce.eps = c.eps;
    ce.L = c.L; ce.tok = tokmap; ce.drop = d_emb;
    ce.x0 = w.x0; ce.x0hat = w.x0hat; ce.rstd0 = w.rstd0;
    ce.wnT = w.layer[0].wqkvT + skip_q * d; ce.ldwn = 3 * d; ce.bn = p0.bqkv + skip_q * d;
    if (wsp) { ce.wnT = w.layer[0].s_wqkvT + 4 * skip_q * d; ce.wsplit = true; }   // (a 16-byte cell per column of a slice plane)
    ce.outn = w.layer[0].qkv + skip_q * d; ce.ldn = 3 * d; ce.Nn = (3 - skip_q) * d;
    ce.M = M; ce.m_dev = mv;
    if ((rc = chain_embed_proj(ce, d, st))) return rc;
    proj_done = true;
  } else {
    rc = embed_ln_fwd(item_seq, item_table, pos, dense + lay.off[1], dense + lay.off[2], c.eps, M, c.L, d, w.x0, w.x0hat, w.rstd0, st,
                      tokmap, mv, &d_emb, n_items);
    if (rc) return rc;
  }
  const float* x = w.x0;
  for (int i = 0; i < c.n_layers; ++i) {
    const LayerP p = layer_ptrs(dense, lay, i);
    LayerWs& lw = w.layer[i];
    GemmArgs g{};
    const DropSpec d_attn = site_spec(c, i, DROP_SITE_ATTN, nullptr);
    if (c.last_only && i == c.n_layers - 1) {
      // Final layer, exact last-row specialisation: K,V for every position, everything else for row L-1 only.
      const int B = c.B;
      const float* x_last = x + (long long)(c.L - 1) * d;   // rows (b, L-1): a strided view, leading dimension L*d
      int ld_last = c.L * d;
      // the one-query attention projects its own queries (and, compact rows, gathers the last rows on the way): no gather launch, no
      // B x d x d GEMM launch in front of it
      if (compact) {   // the last rows are not equally spaced any more: the attention kernel gathers them into x_last
        x_last = w.x_last;
        ld_last = d;
      }
      g.A = x; g.lda = d; g.W = p.wqkv + (long long)d * d; g.ldw = d; g.C = lw.qkv + d; g.ldc = 3 * d; g.M = M; g.N = 2 * d; g.K = d;
      g.bias = p.bqkv + d; g.m_dev = mv;
      if (!proj_done && (rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
      if (lastrow_supported(B, c.L, d, c.n_heads, I)) {
        // ONE launch: query projection -> one-query attention -> out-projection + LN -> feed-forward + LN for the B last rows (lastrow.hip)
        LastRowFwdArgs la{};
        la.x = x; la.xrow = compact ? w.last_row : nullptr; la.xstride = c.L; la.xoff = c.L - 1;
        la.qkv = lw.qkv; la.seq = item_seq; la.seq_base = sbase; la.seq_pad = spad;
        la.wqT = lw.wqkvT; la.ldq = 3 * d; la.bq = p.bqkv; la.woT = lw.woT; la.bo = p.bo; la.g1 = p.g1; la.b1ln = p.b1ln;
        la.w1T = lw.w1T; la.b1 = p.b1; la.w2T = lw.w2T; la.b2 = p.b2; la.g2 = p.g2; la.b2ln = p.b2ln;
        la.q_out = w.q_last; la.x_out = compact ? w.x_last : nullptr; la.ctx = lw.ctx; la.lse = w.lse_last;
        la.a = lw.a; la.ahat = lw.ahat; la.rstd1 = lw.rstd1; la.h1 = lw.h1; la.y = user_emb; la.yhat = lw.yhat; la.rstd2 = lw.rstd2;
        la.B = B; la.L = c.L; la.I = I; la.act = c.act; la.eps = c.eps;
        la.sqrt_hd = sqrtf((float)(d / c.n_heads)); la.scale = 1.0f / la.sqrt_hd;
        la.drop_out = site_spec(c, i, DROP_SITE_OUT, nullptr, c.L, c.L - 1);   // row b of these [B, .] tiles is token (b, L-1)
        la.drop_ffn = site_spec(c, i, DROP_SITE_FFN, nullptr, c.L, c.L - 1);
        la.dkey = d_attn.key; la.dthresh = d_attn.thresh; la.dscale = d_attn.scale;
        return lastrow_fwd(la, d, c.n_heads, st);
      }
      {
        AttnQProj qp{};
        qp.x = x; qp.xrow = compact ? w.last_row : nullptr; qp.xstride = c.L; qp.xoff = c.L - 1;
        qp.wq = p.wqkv; qp.bq = p.bqkv; qp.q_out = w.q_last; qp.x_out = compact ? w.x_last : nullptr;
        if ((rc = attn_last_fwd(nullptr, lw.qkv, item_seq, B, c.L, d, c.n_heads, lw.ctx, w.lse_last, st, sbase, spad, &d_attn, &qp))) return rc;
      }
      if (chain_last) {
        // the B last rows through the same row-chain kernel as the full layers: out-projection + LN + feed-forward + LN in one launch
        // (16 workgroups at B = 512: three latency-bound launches of 10 + 7 + 25 us become one)
        ChainFwdArgs ca{};
        ca.ctx = lw.ctx; ca.ldctx = d; ca.res = x_last; ca.ldres = ld_last;
        ca.woT = lw.woT; ca.bo = p.bo; ca.g1 = p.g1; ca.b1ln = p.b1ln; ca.w1T = lw.w1T; ca.b1 = p.b1; ca.w2T = lw.w2T; ca.b2 = p.b2;
        ca.g2 = p.g2; ca.b2ln = p.b2ln;
        ca.a = lw.a; ca.ahat = lw.ahat; ca.rstd1 = lw.rstd1; ca.h1 = lw.h1; ca.y = user_emb; ca.yhat = lw.yhat; ca.rstd2 = lw.rstd2;
        ca.M = B; ca.I = I; ca.act = c.act; ca.eps = c.eps;
        ca.drop_out = site_spec(c, i, DROP_SITE_OUT, nullptr, c.L, c.L - 1);   // row b of these [B, .] tiles is token (b, L-1)
        ca.drop_ffn = site_spec(c, i, DROP_SITE_FFN, nullptr, c.L, c.L - 1);
        if (I / d >= 2 && cdiv(B, chain_rows_per_block(d)) <= CHAIN_SPLIT_MAX_BLOCKS) {
          ca.split_part = w.split_part;
          return chain_ffn_fwd_split(ca, d, st);
        }
        return chain_ffn_fwd(ca, d, st);
      }
      g = GemmArgs{};
      g.A = lw.ctx; g.lda = d; g.W = p.wo; g.ldw = d; g.C = lw.a; g.ldc = d; g.M = B; g.N = d; g.K = d; g.bias = p.bo;
      g.aux = x_last; g.ldaux = ld_last; g.gamma = p.g1; g.beta = p.b1ln; g.eps = c.eps; g.xhat = lw.ahat; g.rstd = lw.rstd1;
      g.drop = site_spec(c, i, DROP_SITE_OUT, nullptr, c.L, c.L - 1);   // row b of these [B, .] GEMMs is token (b, L-1)
      if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS_RES_LN, st))) return rc;
      g = GemmArgs{};
      g.A = lw.a; g.lda = d; g.W = p.w1; g.ldw = d; g.C = lw.h1; g.ldc = I; g.M = B; g.N = I; g.K = d; g.bias = p.b1;
      if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
      g = GemmArgs{};
      g.A = lw.h1; g.lda = I; g.W = p.w2; g.ldw = I; g.C = user_emb; g.ldc = d; g.M = B; g.N = d; g.K = I; g.bias = p.b2; g.act = c.act;
      g.aux = lw.a; g.ldaux = d; g.gamma = p.g2; g.beta = p.b2ln; g.eps = c.eps; g.xhat = lw.yhat; g.rstd = lw.rstd2;
      g.drop = site_spec(c, i, DROP_SITE_FFN, nullptr, c.L, c.L - 1);
      return gemm_nt(g, PRO_ACT, EPI_BIAS_RES_LN, st);
    }
    g.A = x; g.lda = d; g.W = p.wqkv; g.ldw = d; g.C = lw.qkv; g.ldc = 3 * d; g.M = M; g.N = 3 * d; g.K = d; g.bias = p.bqkv;
    g.m_dev = mv;
    if (!proj_done && (rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
    if ((rc = attn_fwd(lw.qkv, item_seq, c.B, c.L, d, c.n_heads, c.use_pos, lw.ctx, lw.lse, 0, st, sbase, spad, &d_attn))) return rc;
    proj_done = false;
    if (chain) {
      ChainFwdArgs ca{};
      ca.ctx = lw.ctx; ca.ldctx = d; ca.res = x; ca.ldres = d;
      ca.woT = lw.woT; ca.bo = p.bo; ca.g1 = p.g1; ca.b1ln = p.b1ln; ca.w1T = lw.w1T; ca.b1 = p.b1; ca.w2T = lw.w2T; ca.b2 = p.b2;
      ca.g2 = p.g2; ca.b2ln = p.b2ln;
      ca.a = lw.a; ca.ahat = lw.ahat; ca.rstd1 = lw.rstd1; ca.h1 = lw.h1; ca.y = lw.y; ca.yhat = lw.yhat; ca.rstd2 = lw.rstd2;
      ca.M = M; ca.m_dev = mv; ca.I = I; ca.act = c.act; ca.eps = c.eps;
      ca.drop_out = site_spec(c, i, DROP_SITE_OUT, tokmap);
      ca.drop_ffn = site_spec(c, i, DROP_SITE_FFN, tokmap);
      if (i + 1 < c.n_layers) {   // the next layer's input projection rides along: K,V only when that layer is the last-row one
        const LayerP pn = layer_ptrs(dense, lay, i + 1);
        const int skip_q = (c.last_only && i + 1 == c.n_layers - 1) ? 1 : 0;
        ca.wnT = (wsp ? w.layer[i + 1].s_wqkvT + 4 * skip_q * d : w.layer[i + 1].wqkvT + skip_q * d); ca.ldwn = 3 * d; ca.bn = pn.bqkv + skip_q * d;
        ca.outn = w.layer[i + 1].qkv + skip_q * d; ca.ldn = 3 * d; ca.Nn = (3 - skip_q) * d;
        proj_done = true;
      }
      if (wsp) { ca.woT = lw.s_woT; ca.w1T = lw.s_w1T; ca.w2T = lw.s_w2T; ca.wsplit = true; }
      if ((rc = chain_ffn_fwd(ca, d, st))) return rc;
      x = lw.y;
      continue;
    }
    g = GemmArgs{};
    g.A = lw.ctx; g.lda = d; g.W = p.wo; g.ldw = d; g.C = lw.a; g.ldc = d; g.M = M; g.N = d; g.K = d; g.bias = p.bo;
    g.aux = x; g.ldaux = d; g.gamma = p.g1; g.beta = p.b1ln; g.eps = c.eps; g.xhat = lw.ahat; g.rstd = lw.rstd1; g.m_dev = mv;
    g.drop = site_spec(c, i, DROP_SITE_OUT, tokmap);
    if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS_RES_LN, st))) return rc;
    g = GemmArgs{};
    g.A = lw.a; g.lda = d; g.W = p.w1; g.ldw = d; g.C = lw.h1; g.ldc = I; g.M = M; g.N = I; g.K = d; g.bias = p.b1; g.m_dev = mv;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
    g = GemmArgs{};
    g.A = lw.h1; g.lda = I; g.W = p.w2; g.ldw = I; g.C = lw.y; g.ldc = d; g.M = M; g.N = d; g.K = I; g.bias = p.b2; g.act = c.act;
    g.aux = lw.a; g.ldaux = d; g.gamma = p.g2; g.beta = p.b2ln; g.eps = c.eps; g.xhat = lw.yhat; g.rstd = lw.rstd2; g.m_dev = mv;
    g.drop = site_spec(c, i, DROP_SITE_FFN, tokmap);
    if ((rc = gemm_nt(g, PRO_ACT, EPI_BIAS_RES_LN, st))) return rc;
    x = lw.y;
  }
  if (compact) hipLaunchKernelGGL(gather_rows_idx_kernel, dim3(cdiv((long long)c.B * d, 256)), dim3(256), 0, st, x, w.last_row, c.B, d, user_emb);
  else hipLaunchKernelGGL(take_last_kernel, dim3(cdiv((long long)c.B * d, 256)), dim3(256), 0, st, x, c.B, c.L, d, user_emb);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_sasrec_bwd_join(void* stream);

static int sasrec_bwd_impl(const UrSasrecCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                           const int32_t* item_seq, const float* d_user_emb, void* ws, float* dense_grad,
                           float* d_emb_rows, void* stream, bool defer_join) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  UR_REQUIRE(dense && item_seq && d_user_emb && ws && dense_grad && d_emb_rows, UR_ERR_ARG, "ur_sasrec_bwd: null pointer");
  (void)item_table; (void)n_items;
  const UrSasrecCfg& c = *cfg;
  UR_REQUIRE(check_forward(ws, dense, item_seq, c) != 1, UR_ERR_ARG,
             "ur_sasrec_bwd: the last ur_sasrec_fwd on this workspace saw other weights / ids / shapes / mfma_arith (the backward pass reads what it left there)");
  hipStream_t st = as_stream(stream);
  const ArithScope arith_scope(c.mfma_arith | 0x200);   // (the weight-gradient products of this pass: gemm.hip; 0x200: small groups take the split kernel too)
  if ((rc = ur_sasrec_bwd_join(stream))) return rc;   // (a deferred pass nobody joined)
  const Layout lay = make_layout(c);
  Ws w = carve(c, (float*)ws);
  const int M = c.B * c.L, d = c.d, I = c.inner;
  const bool compact = c.skip_padding && attn_compact_supported(c.L, d, c.n_heads);   // row maps were built by ur_sasrec_fwd
  const int* mv = compact ? w.m_valid : nullptr;
  const int* sbase = compact ? w.seq_base : nullptr;
  const int* spad = compact ? w.seq_pad : nullptr;
  // row-chain kernels (rowchain.hip) for the full-sequence layers; hidden dropout keeps the unfused path (the chain backward
  // does not carry the second, dropout-masked copy of the LayerNorm-backward outputs)
  const bool chain_bwd = chain_supported(d, I, CHAIN_BWD);   // (hidden dropout: the masks are re-evaluated in the kernel)
  const bool chain_proj = chain_supported(d, I, CHAIN_PROJ);
  const bool wsp = chains_split(c);   // (the split copies of the weights were made by ur_sasrec_fwd, like the K-major ones)
  const bool chain_last_bwd = chain_supported(d, I, CHAIN_LAST_BWD);
  bool ln0_done = false;                // the embedding LayerNorm backward already ran in the epilogue of the bottom layer's last GEMM
  // LayerNorm backward in the epilogue of the GEMM that produces its input gradient (EPI_ADD_LNBWD): the attention block's
  // LayerNorm behind the d FFN-1 GEMM, the embedding LayerNorm behind the bottom layer's projection-gradient GEMM.  Two launches and
  // two [M, d] round trips less per full layer.  Not with hidden dropout (a second, masked copy of the result would be needed).
  const bool lnfuse = c.p_hidden == 0.f && d <= 128;
  auto lnfuse_part = [&](int slot) { return w.chain_part + (long long)slot * 4 * w.chain_blocks * d; };   // slot n_layers: LN0
  ReduceBatch rb, rb_more[3];           // second stages of all split reductions: launched at the end (4 x 48 items: ~10 layers)
  rb.next = &rb_more[0]; rb_more[0].next = &rb_more[1]; rb_more[1].next = &rb_more[2];
  float* tn_cur = w.tn_ws;
  float* ln_cur = w.ln_part;
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
  // The side stream's GEMMs read the valid-row count from the backward's OWN copy (made by the transpose launch below): with a late join
  // (ur_sasrec_side_publish) the next forward pass's row-compaction plan rewrites m_valid[0] while the side stream may not have started
  // this pass's last GEMM yet -- seen as a loss that differed in the 6th digit in 3 of 14 runs.
  const int* mv_side = mv ? w.m_valid + 16 : nullptr;
  // weight-gradient GEMM, forked onto the side stream (its inputs are complete at this point of the main stream)
  // (the queue of deferred reductions must not flush mid-pass: it chains batches instead; up to 24 forks have an event of their own)
  SideCtx* sc = (c.n_layers <= 6) ? side_ctx() : nullptr;
  int n_fork = 0;
  struct PendingTn { const float *P, *Q; int ldp, ldq, T, R, C, pro_act, act, ldo; float *out, *bias_out, *ws; const int* t_dev; };
  PendingTn pend[12];
  int n_pend = 0;
  // launch the queued weight-gradient GEMMs: on the side stream behind ONE event of the main stream (every queued GEMM's
  // inputs are complete at the point of the main stream where fork() is called), or in line when there is no side stream
  // arm(): the next gemm_nt / attention launch of the main stream carries the next fork's event as its own completion event
  // (UR_LAUNCH_EV) -- the fork behind it then needs no hipEventRecord (a marker packet = ~5 us of idle main stream).  Only in front of a
  // launch that is followed by fork() with queued GEMMs and nothing else on the main stream in between.
  hipEvent_t armed = nullptr;
  const bool timing_producers = prof_brackets(PC_GEMM_NT) || prof_brackets(PC_ATTN_BWD) || prof_brackets(PC_CHAIN_SMALL);   // (their brackets would include the event: see prof_brackets)
  auto arm = [&]() {
    if (!timing_producers && sc && n_fork < 24) { armed = sc->ev[n_fork]; g_stop_event = armed; }
  };
  bool main_done_armed = false, main_done_carried = false;
  auto fork = [&]() -> int {
    const bool carried = armed != nullptr && g_stop_event == nullptr;   // the armed launch took the event
    if (armed) { armed = nullptr; g_stop_event = nullptr; }
    if (n_pend == 0) {
      if (carried) ++n_fork;   // (the event is in flight: its slot is used up)
      return UR_OK;
    }
    hipStream_t s2 = st;
    if (sc && n_fork < 24) {
      if (!carried) UR_HIP(hipEventRecord(sc->ev[n_fork], st));
      UR_HIP(hipStreamWaitEvent(sc->stream, sc->ev[n_fork], 0));
      ++n_fork;
      s2 = sc->stream;
      // test aid (hook side_delay_us, see ur_sasrec_side_stream): the side stream starts this pass's work that much late, i.e. the
      // main stream runs that far ahead of everything the side stream still has to read
      static const int delay_us = ur_test_hook("side_delay_us");
      if (delay_us > 0 && n_fork == 1) hipLaunchKernelGGL(side_delay_kernel, dim3(1), dim3(64), 0, s2, (long long)delay_us * 100);
    }
    {   // every queued product in ONE launch (gemm_tn_group_kernel): few token splits each, small partial tiles
      TnReq rq[12];
      for (int i = 0; i < n_pend; ++i) {
        const PendingTn& t = pend[i];
        rq[i] = TnReq{t.P, t.ldp, t.Q, t.ldq, t.T, t.R, t.C, t.pro_act, t.act, t.out, t.ldo, t.bias_out, t.ws, t.t_dev};
      }
      int rc2 = gemm_tn_group(rq, n_pend, s2, &rb);
      if (rc2) return rc2;
    }
    n_pend = 0;
    // everything queued for the final reduction so far is complete in the side stream's order once these GEMMs are (their own partials;
    // partial sums written by main-stream launches in front of this fork's event): reduced HERE, behind the first fork's GEMMs, it runs
    // in the gap the side stream has before the next fork instead of at the tail of the pass (UR_SASREC_EARLY_REDUCE=0: all at the end)
    if (s2 != st && n_fork == 1 && defer_join) {   // (behind EVERY fork: +25 us -- the later flushes run beside the attention backward)
      int rc2 = reduce_batch(rb, s2);
      if (rc2) return rc2;
    }
    return UR_OK;
  };
  auto tn = [&](const float* P, int ldp, const float* Q, int ldq, int T_, int R_, int C_, int pro_act, int act, float* out, int ldo,
                float* bias_out) -> int {
    if (n_pend == 12) {
      int rc2 = fork();
      if (rc2) return rc2;
    }
    pend[n_pend++] = PendingTn{P, Q, ldp, ldq, T_, R_, C_, pro_act, act, ldo, out, bias_out, tn_take(T_, R_, C_), T_ == M ? mv_side : nullptr};
    return sc ? UR_OK : fork();
  };
  if (!c.last_only) {
    if (compact) {
      UR_HIP(hipMemsetAsync(w.g_y, 0, (size_t)M * d * sizeof(float), st));
      hipLaunchKernelGGL(put_rows_idx_kernel, dim3(cdiv((long long)c.B * d, 256)), dim3(256), 0, st, d_user_emb, w.last_row, c.B, d, w.g_y);
    } else {
      hipLaunchKernelGGL(put_last_kernel, dim3(cdiv((long long)M * d, 256)), dim3(256), 0, st, d_user_emb, c.B, c.L, d, w.g_y);
    }
    UR_LAUNCH_CHECK();
  }

  // zero-fills of the pass (dense_grad: slots nobody writes -- unused position rows, absent parameters; the padded positions' gradient
  // rows) + the backward's own copy of the valid-row count: riders of the last-row layer's launch when that kernel runs (it is the first
  // launch of the pass and nothing reads any of this before it is done), else a launch of their own
  TransposeBatch riders;
  if (lay.total % 4 == 0) { riders.zero_ptr = dense_grad; riders.zero_n = lay.total; }
  else UR_HIP(hipMemsetAsync(dense_grad, 0, lay.total * sizeof(float), st));
  if (compact) {   // padded positions: zero gradient rows (the valid rows are written whole by the bottom layer's projection-gradient launch)
    riders.zero2_ptr = d_emb_rows; riders.zero2_n = (long long)M * d;
    riders.zero2_pad = w.seq_pad; riders.zero2_L = c.L; riders.zero2_d = d;
  }
  if (mv) { riders.copy_src = mv; riders.copy_dst = w.m_valid + 16; }
  const bool riders_ride = c.last_only && lastrow_supported(c.B, c.L, d, c.n_heads, I);
  if (!riders_ride && (rc = transpose_batch(riders, st))) return rc;
  // g_x = g_qkv Wqkv + g_ta as a row-chain launch (CHAIN_PROJ); for the bottom layer the backward of the embedding LayerNorm rides in
  // the epilogue and the rows go straight to their (padded-layout) places in d_emb_rows
  auto proj_chain = [&](int i, LayerWs& lw) -> int {
    const int nblk = cdiv(M, chain_rows_per_block(d));
    ChainProjBwdArgs cp{};
    cp.g = lw.g_qkv; cp.ldg = 3 * d; cp.K = 3 * d; cp.w = layer_ptrs(dense, lay, i).wqkv; cp.ldw = d; cp.res = lw.g_ta;
    cp.out = w.g_y; cp.M = M; cp.m_dev = mv;
    if (wsp) { cp.w = lw.s_wqkv; cp.wsplit = true; }
    if (i == 0) {
      float* part0 = w.chain_part + 4LL * c.n_layers * w.chain_blocks * d;
      cp.xhat = w.x0hat; cp.rstd = w.rstd0; cp.gamma = dense + lay.off[1]; cp.out = d_emb_rows; cp.out_rows = compact ? w.tok_full : nullptr;
      cp.part = part0;
      cp.drop = site_spec(c, 0, DROP_SITE_EMBED, compact ? w.tok_full : nullptr);   // (x0 = dropout(LN0(.)): masked before the LayerNorm backward)
      if (rb.full(2)) {
        int rc2 = reduce_batch(rb, st);
        if (rc2) return rc2;
      }
      rb.add(part0, 2 * d, nblk, d, d, dense_grad + lay.off[1], d);
      rb.add(part0 + d, 2 * d, nblk, d, d, dense_grad + lay.off[2], d);
      ln0_done = true;
    }
    return chain_proj_bwd(cp, d, st);
  };
  // the projection-gradient step of a full layer: chain launch or GEMM (+ LN0 backward in its epilogue for the bottom layer)
  auto proj_step = [&](int i, LayerWs& lw) -> int {
    if (chain_proj) return proj_chain(i, lw);
    int rc2;
    GemmArgs g{};
    g.A = lw.g_qkv; g.lda = 3 * d; g.W = lw.wqkvT; g.ldw = 3 * d; g.C = w.g_y; g.ldc = d; g.M = M; g.m_dev = mv; g.N = d; g.K = 3 * d;
    g.aux = lw.g_ta; g.ldaux = d;
    if (lnfuse && i == 0) {
      // bottom layer: the embedding LayerNorm's backward rides in the epilogue, rows go straight to their (padded-layout) places
      g.C = d_emb_rows; g.xhat = w.x0hat; g.rstd = w.rstd0; g.gamma = dense + lay.off[1]; g.out_rows = compact ? w.tok_full : nullptr;
      g.ln_part = lnfuse_part(c.n_layers);
      // the LAST launch of the pass on the main stream: it carries `main_done` (what the side stream's reductions wait for) itself
      if (!timing_producers && sc && defer_join && n_fork > 0) { g_stop_event = sc->main_done; main_done_armed = true; }
      if ((rc2 = gemm_nt(g, PRO_NONE, EPI_ADD_LNBWD, st))) return rc2;
      main_done_carried = main_done_armed && g_stop_event == nullptr;
      g_stop_event = nullptr;
      if (rb.full(2) && (rc2 = reduce_batch(rb, st))) return rc2;
      rb.add(g.ln_part, 2 * d, gemm_nt_lnbwd_tiles(M), d, d, dense_grad + lay.off[1], d);
      rb.add(g.ln_part + d, 2 * d, gemm_nt_lnbwd_tiles(M), d, d, dense_grad + lay.off[2], d);
      ln0_done = true;
      return UR_OK;
    }
    return gemm_nt(g, PRO_NONE, EPI_ADD, st);
  };
  for (int i = c.n_layers - 1; i >= 0; --i) {
    const LayerP p = layer_ptrs(dense, lay, i);
    const long long* o = lay.off + UR_SASREC_N_GLOBAL + i * UR_SASREC_N_PER_LAYER;
    float* G = dense_grad;
    LayerWs& lw = w.layer[i];
    const float* x_in = (i == 0) ? w.x0 : w.layer[i - 1].y;
    const int* tokmap = compact ? w.tok_full : nullptr;
    const DropSpec d_attn = site_spec(c, i, DROP_SITE_ATTN, nullptr);
    const bool last_rows = c.last_only && i == c.n_layers - 1;
    const DropSpec d_out = last_rows ? site_spec(c, i, DROP_SITE_OUT, nullptr, c.L, c.L - 1) : site_spec(c, i, DROP_SITE_OUT, tokmap);
    const DropSpec d_ffn = last_rows ? site_spec(c, i, DROP_SITE_FFN, nullptr, c.L, c.L - 1) : site_spec(c, i, DROP_SITE_FFN, tokmap);
    if (c.last_only && i == c.n_layers - 1) {
      // final layer: only row L-1 carries gradient (d_user_emb); K,V gradients still cover every position
      const int B = c.B;
      GemmArgs g{};
      if (lastrow_supported(B, c.L, d, c.n_heads, I)) {
        // ONE launch for the B last rows' chain, the one-query attention backward AND the layer's whole input gradient (lastrow.hip)
        float* part = w.chain_part + (long long)i * 4 * w.chain_blocks * d;
        const int nblk = cdiv(B, lastrow_rows_per_block());
        LastRowBwdArgs lb{};
        lb.gy = d_user_emb; lb.yhat = lw.yhat; lb.rstd2 = lw.rstd2; lb.g2 = p.g2; lb.h1 = lw.h1;
        lb.w2 = p.w2; lb.w1 = p.w1; lb.wo = p.wo; lb.wqkv = p.wqkv; lb.ahat = lw.ahat; lb.rstd1 = lw.rstd1; lb.g1 = p.g1;
        lb.q = w.q_last; lb.ctx = lw.ctx; lb.lse = w.lse_last; lb.qkv = lw.qkv; lb.seq = item_seq; lb.seq_base = sbase; lb.seq_pad = spad;
        lb.g_tf = lw.g_tf; lb.g_tfd = lw.g_tfd; lb.g_h1 = lw.g_h1; lb.g_ta = lw.g_ta; lb.g_tad = lw.g_tad; lb.dq = w.dq_last;
        lb.g_qkv = lw.g_qkv; lb.g_x = w.g_y; lb.part = part;
        lb.B = B; lb.L = c.L; lb.I = I; lb.act = c.act;
        lb.sqrt_hd = sqrtf((float)(d / c.n_heads)); lb.scale = 1.0f / lb.sqrt_hd;
        lb.drop_ffn = d_ffn; lb.drop_out = d_out; lb.dkey = d_attn.key; lb.dthresh = d_attn.thresh; lb.dscale = d_attn.scale;
        lb.zero_ptr = riders.zero_ptr; lb.zero_n = riders.zero_n; lb.zero2_ptr = riders.zero2_ptr; lb.zero2_n = riders.zero2_n;
        lb.zero2_pad = riders.zero2_pad; lb.zero2_L = riders.zero2_L; lb.zero2_d = riders.zero2_d;
        lb.copy_src = riders.copy_src; lb.copy_dst = riders.copy_dst;
        arm();
        if ((rc = lastrow_bwd(lb, d, c.n_heads, st))) return rc;
        if (rb.full(4) && (rc = reduce_batch(rb, st))) return rc;
        rb.add(part, 4 * d, nblk, d, d, G + o[14], d);
        rb.add(part + d, 4 * d, nblk, d, d, G + o[15], d);
        rb.add(part + 2 * d, 4 * d, nblk, d, d, G + o[8], d);
        rb.add(part + 3 * d, 4 * d, nblk, d, d, G + o[9], d);
        if ((rc = tn(lw.g_tfd, d, lw.h1, I, B, d, I, 1, c.act, G + o[12], I, G + o[13]))) return rc;
        if ((rc = tn(lw.g_h1, I, lw.a, d, B, I, d, 0, 0, G + o[10], d, G + o[11]))) return rc;
        if ((rc = tn(lw.g_tad, d, lw.ctx, d, B, d, d, 0, 0, G + o[6], d, G + o[7]))) return rc;
        if ((rc = tn(w.dq_last, d, compact ? w.x_last : x_in + (long long)(c.L - 1) * d, compact ? d : c.L * d, B, d, d, 0, 0, G + o[0], d, G + o[3]))) return rc;
        if ((rc = tn(lw.g_qkv + d, 3 * d, x_in, d, M, 2 * d, d, 0, 0, G + o[1], d, G + o[4]))) return rc;
        if ((rc = fork())) return rc;
        continue;
      }
      if (chain_last_bwd) {
        // LN backward -> d act GEMM -> d FFN-1 GEMM + residual -> LN backward -> out-projection GEMM of the B last rows as ONE row-chain
        // launch (the full layers' chain_ffn_bwd; four latency-bound launches of 8 + 13 + 19 + 7 us become one)
        float* part = w.chain_part + (long long)i * 4 * w.chain_blocks * d;
        const int nblk = cdiv(B, chain_rows_per_block(d));
        ChainBwdArgs cb{};
        cb.gy = d_user_emb; cb.yhat = lw.yhat; cb.rstd2 = lw.rstd2; cb.g2 = p.g2; cb.h1 = lw.h1; cb.w2 = p.w2; cb.w1 = p.w1;
        cb.ahat = lw.ahat; cb.rstd1 = lw.rstd1; cb.g1 = p.g1; cb.wo = p.wo;
        cb.g_tf = lw.g_tf; cb.g_h1 = lw.g_h1; cb.g_ta = lw.g_ta; cb.g_ctx = w.g_ctx; cb.part = part;
        cb.M = B; cb.I = I; cb.act = c.act;
        cb.drop_ffn = d_ffn; cb.drop_out = d_out; cb.g_tfd = lw.g_tfd; cb.g_tad = lw.g_tad;
        if (I / d >= 2 && nblk <= CHAIN_SPLIT_MAX_BLOCKS) {   // inner split over workgroups (see chain_ffn_fwd_split)
          cb.split_part = w.split_part + chain_split_part_floats(B, d, I);
          if ((rc = chain_ffn_bwd_split(cb, d, st))) return rc;
        } else if ((rc = chain_ffn_bwd(cb, d, st))) return rc;
        if (rb.full(4) && (rc = reduce_batch(rb, st))) return rc;
        rb.add(part, 4 * d, nblk, d, d, G + o[14], d);
        rb.add(part + d, 4 * d, nblk, d, d, G + o[15], d);
        rb.add(part + 2 * d, 4 * d, nblk, d, d, G + o[8], d);
        rb.add(part + 3 * d, 4 * d, nblk, d, d, G + o[9], d);
        if ((rc = tn(lw.g_tfd, d, lw.h1, I, B, d, I, 1, c.act, G + o[12], I, G + o[13]))) return rc;
        if ((rc = tn(lw.g_h1, I, lw.a, d, B, I, d, 0, 0, G + o[10], d, G + o[11]))) return rc;
        if ((rc = tn(lw.g_tad, d, lw.ctx, d, B, d, d, 0, 0, G + o[6], d, G + o[7]))) return rc;
      } else {
      if ((rc = ln_bwd(d_user_emb, lw.yhat, lw.rstd2, p.g2, nullptr, nullptr, B, d, lw.g_tf, G + o[14], G + o[15], ln_take(), st, &rb,
                       nullptr, nullptr, nullptr, &d_ffn, lw.g_tfd)))
        return rc;
      if ((rc = tn(lw.g_tfd, d, lw.h1, I, B, d, I, 1, c.act, G + o[12], I, G + o[13]))) return rc;
      g.A = lw.g_tfd; g.lda = d; g.W = lw.w2T; g.ldw = d; g.C = lw.g_h1; g.ldc = I; g.M = B; g.N = I; g.K = d; g.aux = lw.h1; g.ldaux = I; g.act = c.act;
      if ((rc = gemm_nt(g, PRO_NONE, EPI_MUL_DACT, st))) return rc;
      if ((rc = tn(lw.g_h1, I, lw.a, d, B, I, d, 0, 0, G + o[10], d, G + o[11]))) return rc;
      g = GemmArgs{};   // (the small weight-gradient GEMMs of this layer are forked together, once, below)
      g.A = lw.g_h1; g.lda = I; g.W = lw.w1T; g.ldw = I; g.C = w.g_a; g.ldc = d; g.M = B; g.N = d; g.K = I; g.aux = lw.g_tf; g.ldaux = d;
      if (lnfuse) {   // the attention block's LayerNorm backward in this GEMM's epilogue (as in the full layers below)
        g.C = lw.g_ta; g.xhat = lw.ahat; g.rstd = lw.rstd1; g.gamma = p.g1; g.ln_part = lnfuse_part(i);
        if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD_LNBWD, st))) return rc;
        if (rb.full(2) && (rc = reduce_batch(rb, st))) return rc;
        rb.add(g.ln_part, 2 * d, gemm_nt_lnbwd_tiles(B), d, d, G + o[8], d);
        rb.add(g.ln_part + d, 2 * d, gemm_nt_lnbwd_tiles(B), d, d, G + o[9], d);
      } else {
        if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD, st))) return rc;
        if ((rc = ln_bwd(w.g_a, lw.ahat, lw.rstd1, p.g1, nullptr, nullptr, B, d, lw.g_ta, G + o[8], G + o[9], ln_take(), st, &rb,
                         nullptr, nullptr, nullptr, &d_out, lw.g_tad)))
          return rc;
      }
      if ((rc = tn(lw.g_tad, d, lw.ctx, d, B, d, d, 0, 0, G + o[6], d, G + o[7]))) return rc;
      g = GemmArgs{};
      g.A = lw.g_tad; g.lda = d; g.W = lw.woT; g.ldw = d; g.C = w.g_ctx; g.ldc = d; g.M = B; g.N = d; g.K = d;
      if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
      }
      arm();
      if ((rc = attn_last_bwd(w.q_last, lw.qkv, item_seq, lw.ctx, w.g_ctx, w.lse_last, B, c.L, d, c.n_heads, w.dq_last, lw.g_qkv, st, sbase, spad, &d_attn))) return rc;
      // dWq from the B last rows, dWk/dWv from all rows
      if ((rc = tn(w.dq_last, d, compact ? w.x_last : x_in + (long long)(c.L - 1) * d, compact ? d : c.L * d, B, d, d, 0, 0, G + o[0], d, G + o[3]))) return rc;
      if ((rc = tn(lw.g_qkv + d, 3 * d, x_in, d, M, 2 * d, d, 0, 0, G + o[1], d, G + o[4]))) return rc;
      if ((rc = fork())) return rc;
      g = GemmArgs{};   // g_x = [dK dV] Wkv  for every row
      g.A = lw.g_qkv + d; g.lda = 3 * d; g.W = lw.wqkvT + d; g.ldw = 3 * d; g.C = w.g_y; g.ldc = d; g.M = M; g.m_dev = mv; g.N = d; g.K = 2 * d;
      if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
      g = GemmArgs{};   // rows L-1 additionally get dq Wq + the residual branch of the attention LayerNorm
      if (compact) {   // last rows are irregularly spaced: dq Wq + g_ta is scatter-accumulated onto them by the epilogue
        g.A = w.dq_last; g.lda = d; g.W = lw.wqkvT; g.ldw = 3 * d; g.C = w.g_y; g.ldc = d; g.M = B; g.N = d; g.K = d;
        g.aux = lw.g_ta; g.ldaux = d; g.out_rows = w.last_row;
        if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD, st))) return rc;
        continue;
      }
      float* gy_last = w.g_y + (long long)(c.L - 1) * d;   // in place on the strided last rows (each element: one thread reads then writes it)
      g.A = w.dq_last; g.lda = d; g.W = lw.wqkvT; g.ldw = 3 * d; g.C = gy_last; g.ldc = c.L * d; g.M = B; g.N = d; g.K = d;
      g.aux = lw.g_ta; g.ldaux = d; g.aux2 = gy_last; g.ldaux2 = c.L * d;
      if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD, st))) return rc;
      continue;
    }
    if (chain_bwd) {
      // ---- the whole block behind the attention as ONE launch: LN backward -> d act GEMM -> d FFN-1 GEMM + residual -> LN backward ->
      // out-projection GEMM; g_tf, g_h1, g_ta are written for the weight-gradient GEMMs, everything else stays in LDS
      float* part = w.chain_part + (long long)i * 4 * w.chain_blocks * d;
      const int nblk = cdiv(M, chain_rows_per_block(d));
      ChainBwdArgs cb{};
      cb.gy = w.g_y; cb.yhat = lw.yhat; cb.rstd2 = lw.rstd2; cb.g2 = p.g2; cb.h1 = lw.h1; cb.w2 = p.w2; cb.w1 = p.w1;
      cb.ahat = lw.ahat; cb.rstd1 = lw.rstd1; cb.g1 = p.g1; cb.wo = p.wo;
      cb.g_tf = lw.g_tf; cb.g_h1 = lw.g_h1; cb.g_ta = lw.g_ta; cb.g_ctx = w.g_ctx; cb.part = part;
      cb.M = M; cb.m_dev = mv; cb.I = I; cb.act = c.act;
      cb.drop_ffn = d_ffn; cb.drop_out = d_out; cb.g_tfd = lw.g_tfd; cb.g_tad = lw.g_tad;
      if (wsp) { cb.w2 = lw.s_w2; cb.w1 = lw.s_w1; cb.wo = lw.s_wo; cb.wsplit = true; }
      if ((rc = chain_ffn_bwd(cb, d, st))) return rc;
      if (rb.full(4) && (rc = reduce_batch(rb, st))) return rc;
      rb.add(part, 4 * d, nblk, d, d, G + o[14], d);
      rb.add(part + d, 4 * d, nblk, d, d, G + o[15], d);
      rb.add(part + 2 * d, 4 * d, nblk, d, d, G + o[8], d);
      rb.add(part + 3 * d, 4 * d, nblk, d, d, G + o[9], d);
      if ((rc = tn(lw.g_tfd, d, lw.h1, I, M, d, I, 1, c.act, G + o[12], I, G + o[13]))) return rc;   // (the activation is applied to h1 on the operand)
      if ((rc = tn(lw.g_h1, I, lw.a, d, M, I, d, 0, 0, G + o[10], d, G + o[11]))) return rc;
      if ((rc = tn(lw.g_tad, d, lw.ctx, d, M, d, d, 0, 0, G + o[6], d, G + o[7]))) return rc;
      // (dW_2, dW_1, dW_o wait for dW_qkv: one launch BEHIND the attention backward, see the unfused path below)
      if ((rc = attn_bwd(lw.qkv, item_seq, lw.ctx, w.g_ctx, lw.lse, c.B, c.L, d, c.n_heads, c.use_pos, lw.g_qkv, w.attn_ws, 0, st, sbase, spad, &d_attn))) return rc;
      if ((rc = tn(lw.g_qkv, 3 * d, x_in, d, M, 3 * d, d, 0, 0, G + o[0], d, G + o[3]))) return rc;
      if ((rc = fork())) return rc;
      if ((rc = proj_step(i, lw))) return rc;
      continue;
    }
    // ---- feed-forward block
    if ((rc = ln_bwd(w.g_y, lw.yhat, lw.rstd2, p.g2, nullptr, nullptr, M, d, lw.g_tf, G + o[14], G + o[15], ln_take(), st, &rb, mv,
                     nullptr, nullptr, &d_ffn, lw.g_tfd)))
      return rc;
    if ((rc = tn(lw.g_tfd, d, lw.h1, I, M, d, I, 1, c.act, G + o[12], I, G + o[13]))) return rc;
    GemmArgs g{};
    g.A = lw.g_tfd; g.lda = d; g.W = lw.w2T; g.ldw = d; g.C = lw.g_h1; g.ldc = I; g.M = M; g.m_dev = mv; g.N = I; g.K = d;
    g.aux = lw.h1; g.ldaux = I; g.act = c.act;
    // dW_2 and dW_1 are forked right behind this GEMM (round 3): they run beside the d FFN-1 and out-projection GEMMs and are done before
    // the attention backward starts; dW_o waits for dW_qkv: the attention backward, which loses most beside a weight-gradient launch
    // (116 us against 49 alone), has the chip to itself (profiles/r03_a_dw_schedule.txt)
    arm();
    if ((rc = gemm_nt(g, PRO_NONE, EPI_MUL_DACT, st))) return rc;
    if ((rc = tn(lw.g_h1, I, lw.a, d, M, I, d, 0, 0, G + o[10], d, G + o[11]))) return rc;
    if ((rc = fork())) return rc;
    g = GemmArgs{};
    g.A = lw.g_h1; g.lda = I; g.W = lw.w1T; g.ldw = I; g.C = w.g_a; g.ldc = d; g.M = M; g.m_dev = mv; g.N = d; g.K = I; g.aux = lw.g_tf; g.ldaux = d;
    if (lnfuse) {
      // ---- d FFN-1 GEMM + residual + the attention block's LayerNorm backward in its epilogue: g_ta directly
      g.C = lw.g_ta; g.xhat = lw.ahat; g.rstd = lw.rstd1; g.gamma = p.g1; g.ln_part = lnfuse_part(i);
      if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD_LNBWD, st))) return rc;
      if (rb.full(2) && (rc = reduce_batch(rb, st))) return rc;
      rb.add(g.ln_part, 2 * d, gemm_nt_lnbwd_tiles(M), d, d, G + o[8], d);
      rb.add(g.ln_part + d, 2 * d, gemm_nt_lnbwd_tiles(M), d, d, G + o[9], d);
    } else {
      if ((rc = gemm_nt(g, PRO_NONE, EPI_ADD, st))) return rc;
      // ---- attention block
      if ((rc = ln_bwd(w.g_a, lw.ahat, lw.rstd1, p.g1, nullptr, nullptr, M, d, lw.g_ta, G + o[8], G + o[9], ln_take(), st, &rb, mv,
                       nullptr, nullptr, &d_out, lw.g_tad)))
        return rc;
    }
    if ((rc = tn(lw.g_tad, d, lw.ctx, d, M, d, d, 0, 0, G + o[6], d, G + o[7]))) return rc;   // (queued: launched with dW_qkv)
    g = GemmArgs{};
    g.A = lw.g_tad; g.lda = d; g.W = lw.woT; g.ldw = d; g.C = w.g_ctx; g.ldc = d; g.M = M; g.m_dev = mv; g.N = d; g.K = d;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
    arm();
    if ((rc = attn_bwd(lw.qkv, item_seq, lw.ctx, w.g_ctx, lw.lse, c.B, c.L, d, c.n_heads, c.use_pos, lw.g_qkv, w.attn_ws, 0, st, sbase, spad, &d_attn))) return rc;
    if ((rc = tn(lw.g_qkv, 3 * d, x_in, d, M, 3 * d, d, 0, 0, G + o[0], d, G + o[3]))) return rc;
    if ((rc = fork())) return rc;
    if ((rc = proj_step(i, lw))) return rc;
  }
  // ---- input block: LN0 backward -> row gradients of E[item_seq] and of the position table  (x0 = dropout(LN0(.)): g_y is
  // masked on read)
  DropSpec d_emb = site_spec(c, 0, DROP_SITE_EMBED, compact ? w.tok_full : nullptr);
  if (!ln0_done && (rc = ln_bwd(w.g_y, w.x0hat, w.rstd0, dense + lay.off[1], nullptr, nullptr, M, d, d_emb_rows, dense_grad + lay.off[1],
                   dense_grad + lay.off[2], ln_take(), st, &rb, mv, compact ? w.tok_full : nullptr, &d_emb)))
    return rc;
  // position-table gradient dP[l,:] = sum_b dx[b,l,:] (no padding index: sasrec.py:25): a split reduction over b with
  // partial stride L*d, queued with the others.  Row L of the table is never looked up (dense_grad was zeroed).
  if (c.use_pos) {
    if (rb.full(1) && (rc = reduce_batch(rb, st))) return rc;
    rb.add(d_emb_rows, (long long)c.L * d, c.B, (long long)c.L * d, c.L * d, dense_grad + lay.off[0], c.L * d);
  }
  UR_REQUIRE(tn_cur <= w.tn_ws + w.tn_floats && ln_cur <= w.ln_part + w.ln_floats, UR_ERR_ARG, "ur_sasrec_bwd: partial-sum workspace overrun");
  if ((rc = fork())) return rc;
  if (n_fork > 0 && defer_join) {
    // deferred join: the reductions into dense_grad run on the SIDE stream, behind the weight-gradient GEMMs there and behind
    // the main stream's last producer of partial sums; the caller's stream goes on (row-gradient reduce, sparse update) and
    // picks dense_grad up with ur_sasrec_bwd_join
    if (!main_done_carried) UR_HIP(hipEventRecord(sc->main_done, st));
    UR_HIP(hipStreamWaitEvent(sc->stream, sc->main_done, 0));
    if ((rc = reduce_batch(rb, sc->stream))) return rc;
    UR_HIP(hipEventRecord(sc->done, sc->stream));
    sc->join_pending = true;
    return UR_OK;
  }
  if (n_fork > 0) {   // join: the partial sums written on the side stream are read by the reduction below
    UR_HIP(hipEventRecord(sc->done, sc->stream));
    UR_HIP(hipStreamWaitEvent(st, sc->done, 0));
  }
  return reduce_batch(rb, st);
}

extern "C" int ur_sasrec_bwd(const UrSasrecCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                             const int32_t* item_seq, const float* d_user_emb, void* ws, float* dense_grad,
                             float* d_emb_rows, void* stream) {
  UR_TRACE_SCOPE();
  return sasrec_bwd_impl(cfg, item_table, n_items, dense, item_seq, d_user_emb, ws, dense_grad, d_emb_rows, stream, false);
}

extern "C" int ur_sasrec_bwd_deferred(const UrSasrecCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                                      const int32_t* item_seq, const float* d_user_emb, void* ws, float* dense_grad,
                                      float* d_emb_rows, void* stream) {
  UR_TRACE_SCOPE();
  return sasrec_bwd_impl(cfg, item_table, n_items, dense, item_seq, d_user_emb, ws, dense_grad, d_emb_rows, stream, true);
}

extern "C" int ur_sasrec_bwd_join(void* stream) {
  UR_TRACE_SCOPE();
  SideCtx* sc = side_ctx(true);   // (a pass deferred before ur_sasrec_set_side_stream(0) still has to be joined)
  if (sc && sc->join_pending) {
    UR_HIP(hipStreamWaitEvent(as_stream(stream), sc->done, 0));
    sc->join_pending = false;
    sc->late_join = false;
  }
  return UR_OK;
}

// The side stream while a deferred pass is pending (else NULL): what the caller enqueues there runs behind the pass's dense-gradient
// reductions with no cross-stream wait in between (the dense half of the optimizer step: the main stream's wait for `done` + the launch
// + the wait's latency were ~30 us at the end of every step during which the main stream did 5 us of work).
// (hook side_delay_us: the spin kernel goes in front of whatever the caller enqueues on the side stream)
extern "C" void* ur_sasrec_side_stream(void) {
  UR_TRACE_SCOPE();
  SideCtx* sc = side_ctx(true);
  if (!(sc && sc->join_pending)) return nullptr;
  static const int delay_us = ur_test_hook("side_delay_us");
  if (delay_us > 0) hipLaunchKernelGGL(side_delay_kernel, dim3(1), dim3(64), 0, sc->stream, (long long)delay_us * 100);   // wall_clock64: 100 MHz
  return (void*)sc->stream;
}

// Marks the end of what the caller added to the side stream: `done` is recorded again.  late != 0: the next ur_sasrec_fwd of this
// process joins it on ITS stream right after its first launch (the row-compaction plan reads ids only, no weights), so the join's
// latency hides under that launch; ur_sasrec_bwd_join(stream) joins at once, as before, whichever was asked for.
extern "C" int ur_sasrec_side_publish(int late) {
  SideCtx* sc = side_ctx(true);
  if (!sc || !sc->join_pending) return UR_OK;
  UR_HIP(hipEventRecord(sc->done, sc->stream));
  sc->late_join = late != 0;
  return UR_OK;
}

// `waiter` waits for everything enqueued on `waited` so far (both streams of the current device), through an event WITHOUT the
// system-scope fence (see side_ctx): what torch's Stream.wait_stream does with a default event, ~1.5 us cheaper on the recording stream.
extern "C" int ur_stream_wait_stream(void* waiter, void* waited) {
  UR_TRACE_SCOPE();
  static thread_local hipEvent_t ring[16];      // (per thread: rank threads of the loopback transport call this concurrently)
  static thread_local int made = 0, next = 0;
  if (!made) {
    for (auto& e : ring) UR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    made = 1;
  }
  hipEvent_t e = ring[next++ & 15];
  UR_HIP(hipEventRecord(e, as_stream(waited)));
  UR_HIP(hipStreamWaitEvent(as_stream(waiter), e, 0));
  return UR_OK;
}

// test aid: a kernel that spins for `us` microseconds on `stream` -- skews one stream against the others (a rank's plan stream running
// late: tests/test_loopback_gpu.py)
extern "C" int ur_debug_delay(int32_t us, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(us >= 0 && us <= 1000000, UR_ERR_ARG, "ur_debug_delay: us=%d", us);
  if (us > 0) hipLaunchKernelGGL(side_delay_kernel, dim3(1), dim3(64), 0, as_stream(stream), (long long)us * 100);   // wall_clock64: 100 MHz
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_sasrec_set_side_stream(int on) {
  UR_TRACE_SCOPE();
  const int prev = g_side_enabled;
  g_side_enabled = on ? 1 : 0;
  return prev;
}
