/* unirec_amd -- C ABI of the MI355X (gfx950) hot-path library for microsoft/UniRec.
 *
 * Drop-in boundary: the reference is pure Python on torch; the arithmetic of its
 * training hot path lives in torch ops called from the files cited next to each entry point below
 * (paths relative to the reference tree, tag 2024_08_07).  A maintainer binds these symbols with
 * ctypes (INTEGRATION.md shows the stub); unirec_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; tensors are dense, row-major, fp32
 *     unless stated; `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - all work is enqueued on `stream`, no hidden synchronisation, no allocation: scratch memory is
 *     provided by the caller (sizes from the *_workspace_bytes queries);
 *   - inputs are borrowed, outputs are written in place; functions return UR_OK (0) or a negative
 *     UR_ERR_* code, and ur_last_error() then returns a thread-local message;
 *   - ids: 0 is the padding id (row 0 of every table never receives a gradient).
 */
#ifndef UNIREC_AMD_H
#define UNIREC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UR_OK 0
#define UR_ERR_ARG (-1)         /* bad argument (null pointer, size, unsupported shape) */
#define UR_ERR_HIP (-2)         /* a HIP runtime call or kernel launch failed */
#define UR_ERR_UNSUPPORTED (-3) /* valid request that this build does not implement */

/* hidden_act ids -- unirec/model/modules.py:337-345 (ACT2FN) */
enum { UR_ACT_GELU = 0, UR_ACT_RELU = 1, UR_ACT_SWISH = 2, UR_ACT_TANH = 3, UR_ACT_SIGMOID = 4, UR_ACT_NONE = 5 };
/* loss_type ids -- unirec/constants/loss_funcs.py:6-11 */
enum { UR_LOSS_NONE = -1 /* scores only */, UR_LOSS_BCE = 0, UR_LOSS_BPR = 1, UR_LOSS_SOFTMAX = 2, UR_LOSS_CCL = 3, UR_LOSS_FULLSOFTMAX = 4 };

const char* ur_last_error(void);
int ur_version(void);

/* ---------------------------------------------------------------------------------------------
 * Id guard -- what nn.Embedding's index check does for the reference (table created at
 * unirec/model/base/reco_abc.py:168-170: an id < 0 or >= n_rows raises IndexError in its forward).
 * Every id of a TRAINING batch passes through ur_rows_plan / ur_rows_plan_sharded once; an id outside
 * [0, n_rows) raises this device's guard there (first offender recorded) and is treated as the
 * padding id 0 by everything that WRITES through the plan (row reduce, sparse update, row exchange):
 * no table byte outside or inside is touched on its behalf.  While the guard is raised, every update
 * entry (ur_sparse_adam_rows*, ur_dense_adam) skips its step exactly like a NaN step, and a sharded
 * rank reports "NaN" in its step flags so that EVERY rank skips -- no extra launch, no host
 * synchronisation.  The host polls ur_id_guard_state (a plain load from a host-mapped mirror the plan
 * kernel writes) at the head of each step and raises IndexError one or two steps after the bad batch.
 * The gathers themselves (lookup kernels, scorer: every entry point that is given the table's row
 * count) read the padding row 0 for an index outside the table in the release build -- a clamp, not
 * a report: forward-only calls (evaluation) have no plan and raise nothing; the bounds-checked build
 * (`python -m unirec_amd.build --debug-bounds`, loaded when UR_DEBUG_BOUNDS=1) prints the site of
 * the first bad index and traps instead.
 * ur_id_guard_state: 0 = clear; 1 = raised, out3 (nullable) = {offending id, rows of the table it
 * was aimed at (saturated to 2^31 - 1), 0}.  ur_id_guard_reset clears it (synchronises `stream`). */
int ur_id_guard_state(int64_t* host_out3);
int ur_id_guard_reset(void* stream);

/* Tracing (SURVEY.md 5 "roctx ranges per op"; the reference has none: torch.profiler ranges are its closest facility).  With UR_ROCTX=1 in
 * the environment every entry point that enqueues device work pushes / pops a roctx range named after itself (libroctx64 resolved at run
 * time; rocprofv3 --marker-trace shows them).  -> the number of ranges pushed so far (0 with the switch off). */
int64_t ur_trace_ranges_pushed(void);

/* ---------------------------------------------------------------------------------------------
 * Embedding lookup: out[i,:] = table[idx[i],:]      (bit-exact copy)
 * replaces nn.Embedding.forward as called by unirec/model/base/recommender.py:67 (forward_item_emb)
 * and :137 (item_embedding_for_user); table created at unirec/model/base/reco_abc.py:168,170.
 * idx_bytes = 4 (int32, the dtype of item_seq) or 8 (int64, the dtype of item_id). d % 4 == 0. */
int ur_embedding_gather_f32(const float* table, int64_t n_rows, int d, const void* idx, int idx_bytes,
                            int64_t n, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SASRec user encoder (unirec/model/sequential/sasrec.py:59-76 + unirec/model/modules.py:284-433):
 *   x0 = LN(E[item_seq] + P[0..L-1]); n_layers x { MHA(+mask) -> FFN }; user_emb = x[:, L-1, :].
 * Dropout: see p_hidden / p_attn below. */
#define UR_MAX_LAYERS 8
typedef struct UrSasrecCfg {
  int32_t B;        /* sequences in this batch */
  int32_t L;        /* max_seq_len */
  int32_t d;        /* hidden_size == embedding_size (sasrec.py:25,60-66), d % 4 == 0, d <= 256 */
  int32_t n_heads;  /* d % n_heads == 0 */
  int32_t inner;    /* inner_size, % 4 == 0 */
  int32_t n_layers; /* 1..UR_MAX_LAYERS */
  int32_t act;      /* UR_ACT_* */
  int32_t use_pos;  /* use_position_emb: adds P and enables the causal part of the mask (sasrec.py:43-52) */
  float eps;        /* layer_norm_eps */
  int32_t last_only; /* 1: exact last-position specialisation of the final layer */
  int32_t skip_padding; /* 1: the left-padded prefix of every sequence gets no rows at all (exact: those positions cannot
                           reach the loss); used when the head dim is 4/8/16 and the sequence fits the MFMA attention kernels
                           (L <= 400 at head dim 8, 249 at 16), ignored otherwise */
  /* Training-time dropout (sasrec.py:69 on the embedded input; modules.py:307 on the attention probabilities; modules.py:313,
   * 352 on the two block outputs before the residual): probabilities in [0,1), 0 = off (evaluation).  The keep mask is a
   * counter-based hash of (drop_seed, drop_step, site, element) -- see DropSpec in csrc/common.h and oracle/dropout_ref.py --
   * never stored: ur_sasrec_bwd must be called with the SAME (drop_seed, drop_step) as the forward it differentiates.  The
   * caller advances drop_step every training step. */
  float p_hidden;       /* hidden_dropout_prob */
  float p_attn;         /* attn_dropout_prob */
  int64_t drop_seed;
  int64_t drop_step;
  int32_t mfma_arith;   /* arithmetic of the encoder's dense contractions: 0 = exact fp32-input MFMA everywhere; 6 / 9 = the fp32 operands
                         * split exactly into three bf16 pieces, six / nine piece products accumulated in fp32 on the bf16 pipes
                         * (fp32-equivalent: measured error vs fp64 <= the exact kernels', profiles/r06_a_stage_a*, r06_k_*); see
                         * ur_set_mfma_arith.  Runs in the split form: (a) the weight-gradient products (autograd of modules.py:285-287,
                         * 312,347-355) -- 128-wide products the wide form (128 x 128 tiles), 64-wide ones the narrow form; products that
                         * fill < 70 % of even 64 x 64 tiles keep the exact kernel unless 0x100 is added (unit tests); (b) the row-chain
                         * kernels of the full-sequence layers, forward and backward (d = 32 / 64 / 128; always six terms; the weights are
                         * pre-split by ur_sasrec_fwd, so ur_sasrec_bwd must be given the SAME mfma_arith).  Attention, the last-row
                         * layer and one-product-per-launch GEMMs stay on the fp32-input MFMA. */
  int32_t reserved_;
} UrSasrecCfg;

/* Layout of the flat dense-parameter buffer (and of its gradient buffer). offsets_out receives
 * UR_SASREC_N_GLOBAL + n_layers * UR_SASREC_N_PER_LAYER element offsets; returns the total number of
 * floats, or a negative error code.
 *   global   : [0] position_embedding.weight [(L+1),d]   [1] LayerNorm.weight [d]   [2] LayerNorm.bias [d]
 *   per layer: [0] query.weight [1] key.weight [2] value.weight (contiguous => Wqkv [3d,d])
 *              [3] query.bias   [4] key.bias   [5] value.bias   (contiguous => bqkv [3d])
 *              [6] dense.weight [d,d] [7] dense.bias [8] attn LayerNorm.weight [9] attn LayerNorm.bias
 *              [10] dense_1.weight [inner,d] [11] dense_1.bias [12] dense_2.weight [d,inner] [13] dense_2.bias
 *              [14] ffn LayerNorm.weight [15] ffn LayerNorm.bias
 * (state_dict names: SURVEY.md section 8b) */
#define UR_SASREC_N_GLOBAL 3
#define UR_SASREC_N_PER_LAYER 16
int64_t ur_sasrec_param_layout(const UrSasrecCfg* cfg, int64_t* offsets_out);
int64_t ur_sasrec_workspace_bytes(const UrSasrecCfg* cfg);

/* forward: writes user_emb [B,d]; activations needed by the backward stay in `ws` -- and so do the K-major copies of every layer's
 * weights the forward makes at its head (one transpose launch).  CONTRACT of the pair: ur_sasrec_bwd must be given the SAME ws and the
 * SAME `dense` values as the ur_sasrec_fwd it differentiates, with no other forward on that ws and no update of `dense` in between
 * (what autograd guarantees for a torch module; an evaluation forward in between needs a workspace of its own).  The library remembers,
 * per calling thread, which (dense, item_seq, shape) the last forward on each of the last 8 workspaces saw, and ur_sasrec_bwd returns
 * UR_ERR_ARG when it is handed a workspace whose last forward saw something else. */
int ur_sasrec_fwd(const UrSasrecCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                  const int32_t* item_seq, float* user_emb, void* ws, void* stream);
/* backward (autograd of the above; the reference has no hand-written backward):
 *   dense_grad  flat buffer, same layout as `dense`, OVERWRITTEN with d loss / d dense-params;
 *   d_emb_rows  [B*L, d]: gradient w.r.t. the gathered rows E[item_seq[b,l]] (row-sparse form of
 *               embedding_dense_backward; rows whose id is 0 hold the gradient w.r.t. the zero padding vector
 *               and are discarded by ur_rows_reduce: padding_idx=0). */
int ur_sasrec_bwd(const UrSasrecCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                  const int32_t* item_seq, const float* d_user_emb, void* ws, float* dense_grad,
                  float* d_emb_rows, void* stream);
/* the same pass with a DEFERRED join (no reference counterpart: loss.backward() is one call there).  On return `stream`
 * holds everything but dense_grad: d_emb_rows is complete in stream order, while the weight-gradient GEMMs and the
 * reductions into dense_grad may still be running on the library's side stream.  The caller does the work that does not
 * need dense_grad (ur_rows_reduce, ur_sparse_adam_rows: the row-sparse half of the optimizer step) and then makes its
 * stream wait with ur_sasrec_bwd_join before the first read of dense_grad.  dense_grad, ws must stay alive until then.
 * Without a side stream (ur_sasrec_set_side_stream(0), > 2 layers) it behaves like ur_sasrec_bwd; ur_sasrec_bwd_join is
 * then a no-op.  A pass nobody joined is joined by the next ur_sasrec_bwd* call. */
int ur_sasrec_bwd_deferred(const UrSasrecCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                           const int32_t* item_seq, const float* d_user_emb, void* ws, float* dense_grad,
                           float* d_emb_rows, void* stream);
int ur_sasrec_bwd_join(void* stream);
/* The side stream (a hipStream_t) while a deferred pass is pending, else NULL: work enqueued there runs behind the pass's
 * dense-gradient reductions without a cross-stream wait (the dense half of the optimizer step).  ur_sasrec_side_publish marks the
 * end of that work (`done` is recorded again); late != 0: the next ur_sasrec_fwd joins it on its own stream after its first launch
 * (which reads ids only), late == 0 / ur_sasrec_bwd_join: the caller joins explicitly.  Until the join nothing else may read or
 * write what the side-stream work touches. */
/* Stream `waiter` waits for what has been enqueued on stream `waited` so far (same device); the event in between carries no system-scope
 * fence.  Plumbing for callers that fork their own side streams (the optimizer's id-plan stream). */
int ur_stream_wait_stream(void* waiter, void* waited);
void* ur_sasrec_side_stream(void);
int ur_sasrec_side_publish(int late);

/* ---------------------------------------------------------------------------------------------
 * ConvFormer / FASTConvFormer user encoders (unirec/model/sequential/convformer.py:16-129,
 * fastconvformer.py:20-81).  x0 = LN(E[item_seq] + P[0..L-1]); per layer y1 = LN(mix(x) + x), y = LN(act(y1 W1^T + b1) W2^T
 * + b2 + y1); no attention mask.  mix: depth-wise Conv1d over the sequence with a (K-1)-row prefix (padding_mode 0
 * circular / 1 reflect / 2 constant), or, fast = 1, the FFT layer's circular convolution (1/sqrt(L)) sum_k w[k,c] x[(l-k) mod L,c]
 * evaluated in the time domain.  Output: position L-1, or (seq_merge) sum_l x[:,l,:] 10^(seq_decay (1 - l/(L-1))) /
 * sqrt(item_seq_len + 1).
 * Flat dense buffer (ur_convformer_param_layout fills 3 + 10*n_layers offsets, returns the total):
 *   global [0] position_embedding.weight [L,d]  [1] LayerNorm.weight  [2] LayerNorm.bias
 *   layer  [0] mixer weight ([d,K] = nn.Conv1d [d,1,K]; fast: [K,d] = conv_weight [1,K,d])  [1] mixer bias [d] (fast: unused)
 *          [2],[3] filterlayer.LayerNorm  [4],[5] intermediate.dense_1 [I,d],[I]  [6],[7] dense_2 [d,I],[d]  [8],[9] intermediate.LayerNorm */
typedef struct UrConvFormerCfg {
  int32_t B, L, d, inner, n_layers;
  int32_t act;          /* UR_ACT_* */
  int32_t conv_size;    /* K <= L */
  int32_t padding_mode; /* 0 circular, 1 reflect, 2 constant */
  int32_t fast;         /* 1: FASTConvFormer's spectral layer */
  int32_t seq_merge;
  float eps, seq_decay;
  float p_hidden;       /* hidden_dropout_prob (training; convformer.py:59,97,115): masks as in UrSasrecCfg, 0 = off */
  int64_t drop_seed, drop_step;
} UrConvFormerCfg;
int64_t ur_convformer_param_layout(const UrConvFormerCfg* cfg, int64_t* offsets_out);
int64_t ur_convformer_workspace_bytes(const UrConvFormerCfg* cfg);
/* seq_len int64[B] = item_seq_len (only read when seq_merge) */
int ur_convformer_fwd(const UrConvFormerCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                      const int32_t* item_seq, const int64_t* seq_len, float* user_emb, void* ws, void* stream);
/* dense_grad: every element written; d_emb_rows [B*L, d] in item_seq.reshape(-1) order (rows of id 0 are discarded by
 * ur_rows_reduce: padding_idx=0). */
int ur_convformer_bwd(const UrConvFormerCfg* cfg, const float* item_table, int64_t n_items, const float* dense,
                      const int32_t* item_seq, const int64_t* seq_len, const float* d_user_emb, void* ws, float* dense_grad,
                      float* d_emb_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * AttHist user encoder (unirec/model/sequential/atthist.py:9-23 + AttentionMergeLayer,
 * unirec/model/modules.py:226-244): z = E[item_seq] W^T + b; p = softmax_l(z . h) over ALL L positions (no mask);
 * user_emb = sum_l p_l z_l.  Flat dense buffer: [0] attention.dense.weight [d,d]  [1] attention.dense.bias [d]
 * [2] attention.h [d] (the reference's [d,1]).  ur_atthist_param_layout fills 3 offsets and returns the total. */
typedef struct UrAttHistCfg {
  int32_t B, L;
  int32_t d; /* embedding_size, % 4 == 0, <= 512 */
  /* training-time dropout on the pooled output (modules.py:231,242; config dropout_prob), 0 = off; masks as in UrSasrecCfg,
   * row id = b */
  float p_drop;
  int64_t drop_seed, drop_step;
} UrAttHistCfg;
int64_t ur_atthist_param_layout(const UrAttHistCfg* cfg, int64_t* offsets_out);
int64_t ur_atthist_workspace_bytes(const UrAttHistCfg* cfg);
int ur_atthist_fwd(const UrAttHistCfg* cfg, const float* item_table, int64_t n_items, const float* dense, const int32_t* item_seq,
                   float* user_emb, void* ws, void* stream);
/* dense_grad: every element written; d_emb_rows [B*L, d] in item_seq.reshape(-1) order (rows of id 0 are discarded by
 * ur_rows_reduce: padding_idx=0). */
int ur_atthist_bwd(const UrAttHistCfg* cfg, const float* item_table, int64_t n_items, const float* dense, const int32_t* item_seq,
                   const float* d_user_emb, void* ws, float* dense_grad, float* d_emb_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pooled-history user encoders: AvgHist (unirec/model/sequential/avghist.py:35-42) and SVD++
 * (unirec/model/sequential/svdplusplus.py:32-40):
 *   user_emb[b,:] = base[b,:] + (seq_len[b] + 1)^(-alpha) * sum_l E[item_seq[b,l],:]      (base nullable; E[0] = 0)
 * backward: d_rows[b*L + l,:] = (seq_len[b] + 1)^(-alpha) * d_user_emb[b,:]  (row-sparse gradient of E in item_seq order;
 * d base = d_user_emb).  item_seq int32[B,L], seq_len int64[B]. */
int ur_pool_rows_fwd(const float* table, int64_t n_rows, int32_t d, const int32_t* item_seq, const int64_t* seq_len,
                     const float* base, float alpha, int32_t B, int32_t L, float* user_emb, void* stream);
int ur_pool_rows_bwd(const float* d_user_emb, const int64_t* seq_len, float alpha, int32_t B, int32_t L, int32_t d,
                     float* d_rows, void* stream);

/* ur_sasrec_bwd forks its weight-gradient GEMMs onto a second HIP stream (joined before it returns, so the call stays
 * stream-ordered for the caller).  0 keeps everything on the caller's stream (measurement of isolated kernel durations,
 * debugging); returns the previous setting.  Default 1 (also: environment UR_SASREC_SIDE=0). */
int ur_sasrec_set_side_stream(int on);

/* Row-chain kernels (csrc/rowchain.hip), alternative schedules of the same arithmetic for d in {32, 64, 128} with inner_size % d == 0.
 * `mask` bit 1: ur_sasrec_fwd runs everything behind the attention of a full-sequence layer -- out-projection + residual + LayerNorm
 * (unirec/model/modules.py:312-316), dense_1, activation, dense_2 + residual + LayerNorm (:347-355) and the NEXT layer's Q/K/V projection
 * (:285-287) -- as one launch with the intermediate tiles in LDS; bit 2: ur_sasrec_bwd runs the mirror image (two LayerNorm backwards +
 * three activation-gradient GEMMs) as one launch; bit 4: the input-gradient GEMM of the projection with the embedding LayerNorm's backward
 * in its epilogue as a chain launch; bits 8 / 16: the B last rows of the last-row layer (last_only) go through the same forward /
 * backward chain kernels instead of three / four small GEMM launches; bit 32: the embedding lookup + position + LayerNorm
 * (unirec/model/sequential/sasrec.py:60-69) and the first layer's Q/K/V projection as one launch; bit 64 (csrc/lastrow.hip; wins over
 * bits 8 / 16; d in {64, 128}, head dim 4 / 8 / 16, 8 <= L <= 64, inner_size 256 / 512): the WHOLE last-row layer as two launches --
 * query projection, one-query attention, out-projection, feed-forward for the B last rows (unirec/model/modules.py:284-316, 347-355,
 * row L-1 only: unirec/model/sequential/sasrec.py:75), and their mirror image including the layer's full input gradient.  Returns the previous mask.  The default mask and the
 * measurements behind it: DESIGN.md section 6d; test hook UR_TEST=chain_mask=<mask>. */
int ur_sasrec_set_chain(int mask);

/* ---------------------------------------------------------------------------------------------
 * GRU user encoder (unirec/model/sequential/gru.py:13-35; arithmetic of torch.nn.GRU, 1 layer, batch_first,
 * h0 = 0, gate order r,z,n; all L steps run including the left padding; user_emb = dense(h_{L-1})).
 * Flat dense buffer layout (ur_gru_param_layout fills 6 offsets, returns the total float count):
 *   [0] gru_layers.weight_ih_l0 [3H,d]  [1] gru_layers.weight_hh_l0 [3H,H]  [2] gru_layers.bias_ih_l0 [3H]
 *   [3] gru_layers.bias_hh_l0 [3H]      [4] dense.weight [d,H]              [5] dense.bias [d] */
typedef struct UrGruCfg {
  int32_t B, L;
  int32_t d; /* embedding_size, % 4 == 0 */
  int32_t H; /* hidden_size, % 4 == 0 */
  /* training-time dropout on the gathered embeddings (gru.py:17,29; config dropout_prob), 0 = off.  Masks as in UrSasrecCfg
   * (counter-based hash, same (drop_seed, drop_step) for the backward); row id of element (b, t, :) is t*B + b. */
  float p_drop;
  int64_t drop_seed, drop_step;
  int32_t mfma_arith;   /* as UrSasrecCfg.mfma_arith: the dW_ih / dW_hh products */
  int32_t reserved_;
} UrGruCfg;
int64_t ur_gru_param_layout(const UrGruCfg* cfg, int64_t* offsets_out);
int64_t ur_gru_workspace_bytes(const UrGruCfg* cfg);
int ur_gru_fwd(const UrGruCfg* cfg, const float* item_table, int64_t n_items, const float* dense, const int32_t* item_seq,
               float* user_emb, void* ws, void* stream);
/* dense_grad: every element written; d_emb_rows [B*L, d] in item_seq.reshape(-1) order (rows of id 0 hold the
 * gradient w.r.t. the zero padding vector and are discarded by ur_rows_reduce: padding_idx=0). */
int ur_gru_bwd(const UrGruCfg* cfg, const float* item_table, int64_t n_items, const float* dense, const int32_t* item_seq,
               const float* d_user_emb, void* ws, float* dense_grad, float* d_emb_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused candidate gather + dot-product scorer + loss
 *   scores[b,g] = ( <E[item_id[b,g]], user_emb[b]> + user_bias[user_id[b]] + item_bias[item_id[b,g]] ) / tau,
 *   clamped to +-score_clip when score_clip > 0        (unirec/model/modules.py:49-67,
 *                                                       unirec/model/base/recommender.py:76-96)
 *   loss = _cal_loss(scores, label)                     (unirec/model/base/reco_abc.py:220-272,
 *                                                       unirec/model/modules.py:15-35)
 * The [B,G,d] candidate tensor is never materialised. */
typedef struct UrLossCfg {
  int32_t B, G, d;
  int32_t loss_type;   /* UR_LOSS_BCE | BPR | SOFTMAX | CCL */
  float tau;
  float score_clip;    /* <= 0: off */
  float ccl_w, ccl_m;
  int32_t group_size;  /* 0: off.  > 0: the user-item-label row format (unirec/model/base/reco_abc.py:233-236): G == 1, one (user, item,
                        * label) triple per row, every group_size consecutive rows are ONE score row of the loss; loss_rows then holds
                        * B / group_size row losses.  ur_gather_dot_loss_fused_supported answers 0: call _fwd and _bwd. */
} UrLossCfg;
/* user_bias / item_bias / user_id may be NULL (bias off). label (int32 [B,G]) is required for BCE and
 * SOFTMAX. Outputs: scores [B,G]; loss_out[4] ([3] unused): [2] = update guard (1, or -1 when the loss is NaN), [0] = reduced loss (reduction=True), [1] = the mean's
 * denominator (rows, elements or positives); loss_rows [2*B]: per-row numerators (for BPR/CCL the
 * reduction=False row losses) followed by per-row denominators. */
int ur_gather_dot_loss_fwd(const UrLossCfg* cfg, const float* user_emb, const float* item_table, int64_t n_items,
                           const int64_t* item_id, const int32_t* label, const float* user_bias,
                           const float* item_bias, const int64_t* user_id, float* scores, float* loss_rows,
                           float* loss_out, void* stream);
/* backward for upstream gradient d_loss (device scalar, NULL = 1.0); loss_out is the forward's output:
 *   coef[b,g]   = d loss / d <E,u>            (so  dE[item_id[b,g]] += coef[b,g] * user_emb[b], implicit)
 *   d_user[b,:] = sum_g coef[b,g] * E[item_id[b,g],:]
 *   d_user_bias_rows[b] = sum_g coef[b,g]  (NULL to skip);  the item-bias row gradient equals coef. */
int ur_gather_dot_loss_bwd(const UrLossCfg* cfg, const float* user_emb, const float* item_table, int64_t n_items,
                           const int64_t* item_id, const int32_t* label, const float* scores,
                           const float* loss_out, const float* d_loss, float* coef, float* d_user,
                           float* d_user_bias_rows, void* stream);
/* The training step's loss section as ONE launch: scores, per-row loss, d loss / d score (coef), d_user and the batch loss +
 * update guard (loss_out, as ur_gather_dot_loss_fwd leaves it) -- what ur_gather_dot_loss_fwd followed by
 * ur_gather_dot_loss_bwd (d_loss = NULL) compute in three launches.  Available for bpr / bce / ccl (the mean's denominator is
 * B or B * G, known before the scores are) with G * d * 4 <= 32 KB: ask ur_gather_dot_loss_fused_supported, otherwise call the
 * two entry points above.  Reference: unirec/model/base/recommender.py:199-241 (scorer + loss) under loss.backward().
 * Launches on one device must not overlap each other (a per-device completion counter). */
int ur_gather_dot_loss_fused_supported(const UrLossCfg* cfg);
int ur_gather_dot_loss_fwd_bwd(const UrLossCfg* cfg, const float* user_emb, const float* item_table, int64_t n_items,
                               const int64_t* item_id, const int32_t* label, const float* user_bias,
                               const float* item_bias, const int64_t* user_id, float* scores, float* loss_rows,
                               float* loss_out, float* coef, float* d_user, float* d_user_bias_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row-sparse gradient of an embedding table (replaces embedding_dense_backward + the dense [N,d]
 * gradient of nn.Embedding(padding_idx=0), unirec/model/base/reco_abc.py:170 / SURVEY.md K2).
 * ur_rows_plan sorts the looked-up ids (stable LSD radix sort, deterministic), and produces
 *   uniq_idx[u]  distinct ids in ascending order (id 0 included if looked up),
 *   seg_start[u] start of id u's run in sorted_pos,     seg_start[n_uniq] = n,
 *   sorted_pos[] lookup positions grouped by id (position p < n_a refers to ids_a, else to ids_b[p-n_a]),
 *   n_uniq_dev[0] number of distinct ids (device int32; nothing is copied to the host).
 * ids_a: int32 [n_a] (item_seq), ids_b: int64 [n_b] (item_id); either may be empty. */
int64_t ur_rows_plan_workspace_bytes(int64_t n);
int ur_rows_plan(const int32_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b, int64_t n_rows,
                 int32_t* uniq_idx, int32_t* seg_start, int32_t* sorted_pos, int32_t* n_uniq_dev, void* ws,
                 void* stream);
/* The same plan for ids that arrive as n_runs concatenated runs, each ascending and unique (the owner side of the row-sharded
 * step: one run per sending rank): a W-way merge by binary searches instead of a sort.  host_run_start[n_runs + 1] (HOST
 * array, start[0] = 0, start[n_runs] = n).  Output identical to ur_rows_plan(ids, n, NULL, 0, ...).  ws: ur_rows_plan_workspace_bytes(n). */
int ur_rows_plan_merge(const int32_t* ids, int64_t n, const int32_t* host_run_start, int32_t n_runs, int32_t* uniq_idx,
                       int32_t* seg_start, int32_t* sorted_pos, int32_t* n_uniq_dev, void* ws, void* stream);
/* Row-sharded table (row i lives on rank i % world at local row i / world; SURVEY.md 8e -- not in the reference,
 * whose only strategy is DDP over a replicated dense table: unirec/facility/trainer.py:67).  Same as ur_rows_plan,
 * but the sort key is owner * ceil(n_rows/world) + local_row, so uniq_key[] is grouped by owner rank and
 * owner_counts_dev[r] (device int32[world], nullable) = number of distinct rows requested from rank r. */
int ur_rows_plan_sharded(const int32_t* ids_a, int64_t n_a, const int64_t* ids_b, int64_t n_b, int64_t n_rows,
                         int32_t world, int32_t* uniq_key, int32_t* seg_start, int32_t* sorted_pos, int32_t* n_uniq_dev,
                         int32_t* owner_counts_dev, void* ws, void* stream);
/* ---- row exchange of the row-sharded tables: SURVEY.md 8b `a2a_embedding_exchange(...)` / 8e.  Replaces what DDP does for the embedding
 * table in the reference (unirec/facility/trainer.py:67,261: accelerator.prepare -> DDP; :346 accelerator.backward all-reduces the dense
 * [N, d] gradient): rows travel instead of tables.  FIXED CAPACITY: every (source, owner) pair carries exactly `cap` slots in all three
 * exchanges of a step, so no count exchange and no host synchronisation exist; slot q = owner * cap + p of a rank's block holds its
 * requests for that owner right-aligned behind padding slots that ask for local row 0.  transport != 0: the packed block is moved by
 * the library's RCCL communicator (ur_comm_init) on `stream`; transport == 0: pack / unpack only, the caller moves the block itself
 * (torch.distributed over gloo in the CPU-staged tests).
 *   ur_shard_exchange_ids : plan keys (ur_rows_plan_sharded: uniq_key, n_uniq_dev; the per-owner counts are found by bisection of the
 *                           sorted keys and written to counts_dev when non-NULL) -> send_ids[world*cap] (+ the maps
 *                           slot_of_uniq[u] / u_of_slot[q], -1 for padding) -> recv_ids[world*cap]; flags_dev[0] |= 1 if a count > cap
 *   ur_shard_exchange_rows: rows_ws[q,:] = table[req_ids[q],:] (this rank's shard) -> compact[world*cap, d] on the requesters
 *   ur_shard_exchange_grads: uniq_grad[n_uniq,d] -> slot layout (padding: zeros) in send_ws -> grads_in[world*cap, d] on the owners;
 *                           uniq_grad NULL: the rows are in their slots of send_ws already (ur_rows_reduce with out_rows = slot_of_uniq;
 *                           the padding slots then hold stale bytes, which no owner reads: they ask for the padding row) and only the
 *                           flag rows are written */
int ur_shard_exchange_ids(const int32_t* uniq_key, const int32_t* n_uniq_dev, int32_t* counts_dev, int64_t n_local,
                          int32_t world, int32_t cap, int32_t* send_ids, int32_t* slot_of_uniq, int32_t* u_of_slot,
                          int32_t* flags_dev, int32_t* recv_ids, int32_t transport, void* stream);
int ur_shard_exchange_rows(const float* table, const int32_t* req_ids, int32_t world, int32_t cap, int32_t d, float* rows_ws,
                           float* compact, int32_t transport, void* stream);
int ur_shard_exchange_grads(const float* uniq_grad, const int32_t* u_of_slot, int32_t world, int32_t cap, int32_t d,
                            const float* loss_out, const int32_t* flags_dev, float* send_ws, float* grads_in, int32_t transport,
                            void* stream);
/* The row exchange of batch t + 1 made a step AHEAD ("rows travel"; no reference counterpart: DDP replicates the table).
 * ur_shard_exchange_rows for the next batch runs on the plan stream under step t's forward / backward (ur_comm_all_to_all with ahead = 1:
 * the second communicator); the rows step t's own update still changes -- the owner-side unique rows of step t, `prev_uniq` -- are re-sent
 * behind that update in a small exchange of cap2 <= cap slots per (owner, requester) pair:
 *   ur_shard_fixup_plan : (ids only) for every source block of the NEXT batch's request list recv_ids[world*cap]: req2[s*cap2 + i] = the
 *                         rows it asks for that are in prev_uniq (ascending unique int32, *prev_n_uniq_dev of them; NULL = none), in
 *                         arbitrary order, slot2[..] = the slot of each inside the block; padding (row 0, slot -1) behind them;
 *                         flags_dev[0] |= 1 when a block has more than cap2
 *   ur_shard_fixup_apply: compact[(q / cap2) * cap + slot2[q], :] = rows2[q, :] for the received slots with slot2[q] >= 0
 * ur_rows_split_hot splits a plan's unique rows against prev_uniq: hot = in both, cold = the others that have optimizer history
 * (last_step[row] != 0; last_step NULL: all others) -- the rows a lazy catch-up made a step ahead may touch. */
int ur_comm_all_to_all(const void* send, void* recv, int64_t bytes_per_peer, int32_t ahead, int32_t kind /* 0 ids, 1 rows, 2 row gradients: the timer class */,
                       void* stream);
int ur_shard_fixup_plan(const int32_t* recv_ids, int32_t world, int32_t cap, const int32_t* prev_uniq, const int32_t* prev_n_uniq_dev,
                        int64_t prev_n_max, int32_t cap2, int32_t* req2, int32_t* slot2, int32_t* counts_ws /* world ints */,
                        int32_t* flags_dev, void* stream);
int ur_shard_fixup_apply(float* compact, const float* rows2, const int32_t* slot2, int32_t world, int32_t cap, int32_t cap2, int32_t d,
                         void* stream);
int ur_rows_split_hot(const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const int32_t* last_step,
                      const int32_t* excl_sorted, const int32_t* excl_n_dev, int64_t excl_max, int32_t* cold_idx, int32_t* cold_n_dev,
                      int32_t* hot_idx, int32_t* hot_n_dev, int32_t* hot_u /* nullable: index of each hot row in excl_sorted */,
                      int32_t* excl_mark /* nullable, excl_max ints: 1 for the entries of excl_sorted that are hot */, void* stream);
/* Slot 0 of every block is reserved padding (a block holds cap - 1 keys); in the gradient exchange it carries the sender's step flags
 * [loss is NaN (loss_out[2] < 0, as the loss kernels publish it), capacity overflow (flags_dev[0] & 1), loss_out[0], 1] to every owner.
 * ur_shard_step_flags sums them in source-rank order: out4 = [gradient scale for the update kernels: 1 / world (DDP's mean,
 * trainer.py:346) or -1 = skip the step on every rank, mean loss over the ranks (trainer.py:353 gather_for_metrics(loss).mean()),
 * ranks with a NaN loss, ranks with an overflow]. */
int ur_shard_step_flags(const float* grads_in, int32_t world, int32_t cap, int32_t d, float* out4, void* stream);
/* The library's RCCL communicator (one per process, one process per GPU; RCCL is resolved at run time from the librccl.so.1 the process has
 * loaded).  ur_comm_world: -1 no RCCL library, 0 not initialised, else the communicator's size.  ur_comm_unique_id: 256 bytes (two
 * ids: one communicator for the row exchanges, one for the dense all-reduce, so that neither queues behind the other) made on
 * rank 0 and handed to every rank's ur_comm_init by the host (a broadcast over its process group).  ur_comm_all_reduce_sum: in place,
 * fp32 -- the flat dense-gradient all-reduce of the step (what DDP's bucketed all-reduce does, trainer.py:346), on `stream`. */
int ur_comm_world(void);
/* RCCL's own view of the two communicators (ncclCommCount / ncclCommUserRank): out4 (nullable) = {ranks of the row communicator, ranks
 * of the ahead communicator, this rank in each}; returns the rank count (0: not initialised), < 0 if they disagree with each other or
 * with ur_comm_init's arguments.  What `bench.py --gpus N` prints as rccl_ranks (trainer.py:67,261: Accelerate's num_processes). */
int ur_comm_count(int32_t* out4);
/* ---- in-process loopback transport (test / measurement infrastructure of the multi-GPU step on a 1-GPU box; csrc/exchange.hip has the
 * protocol): W rank contexts in ONE process on one device.  A rank is the THREAD that called ur_loop_attach(group, rank): from then on its
 * per-context library state (the encoder's side stream and events, hand-off counters) is its own.  A collective is ur_loop_post ->
 * [host rendezvous of the rank threads] -> ur_loop_all_to_all_pull | ur_loop_all_reduce_pull -> [host rendezvous] -> ur_loop_finish, all
 * non-blocking and stream-ordered (cross-rank events; no device-host synchronisation).  comm = 0 / 1: the two communicators of the real
 * transport -- operations of one index run in issue order, as RCCL's do.  ur_loop_finish(all_reduce_out != NULL, n): the summed buffer is
 * copied in place.  The reference's counterpart: a 2-process NCCL run (tests/test_model/run_ddp_test.sh:28-86). */
void* ur_loop_create(int32_t world);
int ur_loop_destroy(void* group);
int ur_loop_attach(void* group, int32_t rank);
int ur_loop_detach(void);
int ur_loop_world(void);
int ur_loop_post(const void* send, int32_t comm, void* stream);
int ur_loop_all_to_all_pull(void* recv, int64_t bytes_per_peer, int32_t kind, int32_t comm, void* stream);
int ur_loop_all_reduce_pull(int64_t n, int32_t comm, void* stream);
int ur_loop_finish(int32_t comm, float* all_reduce_out, int64_t n, void* stream);
/* test aid: a kernel that spins for `us` microseconds on `stream` (skews one stream of a schedule against the others) */
int ur_debug_delay(int32_t us, void* stream);
int ur_comm_unique_id(void* id_out256);
int ur_comm_init(const void* id256, int32_t rank, int32_t world);
int ur_comm_destroy(void);
int ur_comm_all_reduce_sum(float* buf, int64_t n, void* stream);
/* idx_a[p] (p < n_a) / idx_b[p - n_a] = u for every lookup position p in the run of unique key u: the batch's
 * lookups re-expressed as indices into the compact [n_uniq, d] table of fetched rows; slot_of_uniq (nullable, from
 * ur_shard_exchange_ids): the index is slot_of_uniq[u] instead -- the row of the fixed-capacity [world * cap, d] table. */
int ur_compact_index(const int32_t* seg_start, const int32_t* sorted_pos, const int32_t* n_uniq_dev, int64_t n, int64_t n_a,
                     const int32_t* slot_of_uniq, int32_t* idx_a, int64_t* idx_b, void* stream);
/* uniq_grad[u,:] = sum over the run of uniq_idx[u] (in sorted, i.e. position, order) of
 *   rows_a[p,:]                      for p <  n_a
 *   coef_b[p-n_a] * vec_b[(p-n_a)/G,:] for p >= n_a     (the scorer's implicit candidate-row gradient)
 * rows of id 0 are written as zeros. sumsq_dev non-NULL: the rows n_uniq .. n-1 of uniq_grad are zeroed as well (a caller that
 * sums squares over all n rows).  out_rows (nullable, not with sumsq_dev): row u is written to uniq_grad[out_rows[u],:] instead of
 * uniq_grad[u,:] -- e.g. slot_of_uniq of ur_shard_exchange_ids: the sums land in their exchange slots, no scatter pass. */
int ur_rows_reduce(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos,
                   const int32_t* n_uniq_dev, int64_t n, const float* rows_a, int64_t n_a, const float* coef_b,
                   const float* vec_b, int32_t G, int32_t d, float* uniq_grad, float* sumsq_dev, const int32_t* out_rows,
                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer (torch.optim.Adam as built at unirec/facility/trainer.py:134-136 and stepped at :349;
 * weight_decay is L2 added to the gradient; bias-corrected; denom = sqrt(v)/sqrt(1-b2^t) + eps).
 * grad_scale_dev: optional device float multiplied into every gradient (global-norm clipping,
 * trainer.py:347-348); NULL = 1. */
/* algo: which torch.optim rule (the reference passes only lr and weight_decay, trainer.py:134-152, so every other
 * hyper-parameter is torch's default and is what the host fills in):
 *   UR_OPT_ADAM     Adam(betas 0.9/0.999, eps 1e-8), weight_decay = L2 added to the gradient
 *   UR_OPT_ADAMW    same moments, decoupled decay w *= 1 - lr*wd
 *   UR_OPT_SGD      plain SGD (momentum 0); m, v unused
 *   UR_OPT_ADAGRAD  state_sum in v (eps 1e-10, lr_decay 0); m unused
 *   UR_OPT_RMSPROP  square_avg in v, alpha in beta2 (0.99), eps 1e-8, momentum 0, not centered; m unused
 * ('sparse_adam' has no meaning for the reference's dense nn.Embedding gradients -- torch raises; here it selects Adam
 *  with the row-wise table mode.) */
enum { UR_OPT_ADAM = 0, UR_OPT_ADAMW = 1, UR_OPT_SGD = 2, UR_OPT_ADAGRAD = 3, UR_OPT_RMSPROP = 4 };
typedef struct UrAdamCfg {
  float lr, beta1, beta2, eps, weight_decay;
  int32_t step; /* 1-based step count t of THIS update */
  int32_t algo; /* UR_OPT_* */
} UrAdamCfg;
int ur_dense_adam(const UrAdamCfg* cfg, float* param, const float* grad, float* m, float* v, int64_t n,
                  const float* grad_scale_dev, void* stream);
/* Row-wise Adam on the rows listed in uniq_idx (id 0 skipped).
 * last_step == NULL : "rowwise" semantics, untouched rows do not move (torch SparseAdam-like, but with
 *                     the global step in the bias correction).
 * last_step != NULL : "lazy dense" semantics = the reference's dense Adam, evaluated lazily: int32
 *                     last_step[N] remembers the step at which a row's (w,m,v) are valid; before the row
 *                     is used, ur_lazy_adam_catchup applies the zero-gradient Adam steps it missed
 *                     (exact for weight_decay == 0). */
int ur_sparse_adam_rows(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step,
                        const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const float* uniq_grad,
                        int32_t d, const float* grad_scale_dev, void* stream);
/* ur_rows_reduce + ur_sparse_adam_rows in ONE launch: the per-row gradient sums (ur_rows_reduce's arguments, same summation order) go
 * straight into the row update (ur_sparse_adam_rows's arguments, same arithmetic) and never reach HBM -- bit-identical tables, two row-array
 * passes and one dependent launch fewer.  No uniq_grad comes out: a caller that clips by the global norm (the reference's
 * clip_grad_norm_, unirec/facility/trainer.py:347-348) or exchanges row gradients between ranks uses the two calls.
 * next_cold_idx / next_hot_idx (optional, lazy-dense tables; ur_rows_split_hot's two outputs for the NEXT batch's plan against this one,
 * capacity next_list_max each): the next batch's lazy replay rides in the same launch -- extra workgroups bring its `cold` rows (not in
 * this step's plan: disjoint from every row the update writes) to "after step cfg->step"; its `hot` rows (in both plans) are made current
 * by the update itself, and replayed like the cold ones when the step is skipped (grad_scale < 0).  Replaces the ur_lazy_adam_catchup
 * launch that would follow. */
int ur_rows_reduce_update(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos, const int32_t* n_uniq_dev, int64_t n,
                          const float* rows_a, int64_t n_a, const float* coef_b, const float* vec_b, int32_t G, int32_t d,
                          const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step, const float* grad_scale_dev,
                          const int32_t* next_cold_idx, const int32_t* next_cold_n_dev, const int32_t* next_hot_idx,
                          const int32_t* next_hot_n_dev, int64_t next_list_max, void* stream);
/* ur_rows_reduce_update on the OWNER side of the row-sharded step (= ur_rows_reduce_riders with step_flags_out4, + ur_sparse_adam_rows):
 * recv_rows is the received gradient block [world * cap, d] (n = world * cap plan entries, every position an explicit row); the gradient
 * scale is the step's flags found in slot 0 of its source blocks (1 / world, or skip the step: a NaN loss, an id out of range or a capacity
 * overflow on ANY rank), published to step_flags_out4 as ur_shard_step_flags does. */
int ur_rows_reduce_update_owner(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos, const int32_t* n_uniq_dev,
                                int64_t n, const float* recv_rows, int32_t d, int32_t world, int32_t cap, float* step_flags_out4,
                                const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step, void* stream);
/* brings rows uniq_idx[0..n_uniq) to the state "after step (cfg->step - 1)" */
int ur_lazy_adam_catchup(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step,
                         const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, int32_t d,
                         void* stream);
/* the same replay for a launch issued under a step's compute (a side stream beside forward / backward kernels): a grid of at most one
 * workgroup per CU, default wave priority.  Same values. */
int ur_lazy_adam_catchup_background(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step,
                                    const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, int32_t d, void* stream);
/* The row update of a step in two launches, so that most of it can run beside the NEXT forward pass (facility/optimizer.py; no
 * reference counterpart: torch.optim steps every parameter before the next forward, unirec/facility/trainer.py:349):
 *   ur_rows_reduce_subset    : ur_rows_reduce for the unique ids u_list[0 .. *n_list_dev) only, entry i -> out[i, :] (same sums, same order)
 *   ur_sparse_adam_rows_split: hot != 0: ur_sparse_adam_rows over a short row list (the rows the next batch reads as well); a skipped step
 *                              (grad_scale_dev < 0) is applied to them as a zero-gradient step.  hot == 0: over the whole plan except
 *                              the unique ids with skip_mark[u] != 0. */
/* ur_rows_reduce of the row-sharded step with two riders that used to be launches of their own: step_flags_out4 != NULL (owner-side
 * reduce: rows_a = the received gradient block [world * cap, d]): out4 as ur_shard_step_flags; write_flag_rows != 0 (requester-side reduce
 * with out_rows = the exchange slots): this rank's flag row into slot 0 of every block of uniq_grad, as ur_shard_exchange_grads(uniq_grad
 * = NULL) writes it (loss_out / flags_dev as there). */
int ur_rows_reduce_riders(const int32_t* uniq_idx, const int32_t* seg_start, const int32_t* sorted_pos, const int32_t* n_uniq_dev,
                          int64_t n, const float* rows_a, int64_t n_a, const float* coef_b, const float* vec_b, int32_t G, int32_t d,
                          float* uniq_grad, float* sumsq_dev, const int32_t* out_rows, int32_t world, int32_t cap,
                          float* step_flags_out4, int32_t write_flag_rows, const float* loss_out, const int32_t* flags_dev, void* stream);


/* out_idx[0 .. *out_n_dev) = the rows of uniq_idx[0 .. *n_uniq_dev) that were ever updated (last_step != 0), in arbitrary order: with
 * weight_decay == 0 the only rows ur_lazy_adam_catchup has work for.  Made next to the plan (side stream), it keeps the catch-up of a
 * batch of never-seen rows -- a chain of random last_step reads and nothing else -- off the main stream.  out_idx: n_max ints. */
int ur_rows_filter_touched(const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const int32_t* last_step,
                           int32_t* out_idx, int32_t* out_n_dev, void* stream);
/* ur_sparse_adam_rows (lazy-dense semantics, last_step != NULL) for THIS step's rows and ur_lazy_adam_catchup for the NEXT batch's
 * rows (next_uniq_idx: its plan's row list) in ONE launch: the next batch's rows -- minus the ones updated here -- are brought to
 * the state after this step (cfg->step).  Same results as the two calls in sequence; the two halves are latency-bound chains of random
 * accesses and overlap instead of queueing up at the tail of the step. */

/* same for a contiguous block of rows [row0, row0+n): flush before evaluation / checkpoint */
int ur_lazy_adam_flush(const UrAdamCfg* cfg, float* table, float* m, float* v, int32_t* last_step, int64_t row0,
                       int64_t n, int32_t d, void* stream);

/* out[0] (+)= sum(x[i]^2), deterministic two-stage reduction; accumulate != 0 adds to out[0]. */
int ur_sumsq(const float* x, int64_t n, float* out, int accumulate, void* ws_2048_floats, void* stream);
/* scale_out[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))   (torch clip_grad_norm_ coefficient) */
int ur_clip_coef(const float* sumsq, float max_norm, float* scale_out, void* stream);
/* NaN guard (Trainer.fit skips the update of a step whose loss is NaN: unirec/facility/trainer.py:164-168,343-350).  The loss
 * kernels publish loss_out[2] = 1, or -1 when the loss is NaN; a grad_scale_dev value < 0 makes ur_dense_adam / ur_sparse_adam_rows
 * return without touching anything (no host round trip).  This variant of ur_clip_coef passes the guard through:
 * scale_out[0] = guard[0] < 0 ? -1 : clip coefficient. */
int ur_clip_coef_guarded(const float* sumsq, float max_norm, const float* guard, float* scale_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Negative sampling and row construction (replaces the per-sample Python of
 * unirec/data/transform/addnegsamples.py:90-115, unirec/data/transform/adduserhistory.py:32-73,
 * unirec/data/dataset/seqrecdataset.py:38-68, unirec/utils/sampling.py:9-31).
 *
 * HOST generator (all pointers are HOST pointers): CPython-compatible MT19937; a stream created with seed s
 * produces exactly what `random.seed(s)` + the reference transforms produce on one stream (num_workers=0). */
void* ur_host_sampler_create(uint64_t seed);
void ur_host_sampler_destroy(void* sampler);
uint64_t ur_host_sampler_getrandbits(void* sampler, int k);          /* random.getrandbits(k), k <= 64 */
double ur_host_sampler_random(void* sampler);                         /* random.random() */
int64_t ur_host_sampler_randint(void* sampler, int64_t a, int64_t b); /* random.randint(a, b) */
/* popularity-biased negatives: host_weights[n] = pop^alpha / sum with weight[0] = 0 (addnegsamples.py:58-62) */
int ur_host_sampler_set_alias(void* sampler, const double* host_weights, int64_t n);
/* One batch of rows (SeqRecDataset.__getitem__ order of RNG use: negatives, then the history cut).
 * hist_ptr[n_users+1] + hist_items: histories in interaction order; hist_sorted: same ranges sorted ascending.
 * mask_mode 0 = 'unorder', 1 = 'autoregressive', 2 = other (history unchanged).  item_seq may be NULL (MF). */
int ur_host_build_rows(void* sampler, const int64_t* user_id, const int64_t* pos_item, int64_t n, int64_t n_users,
                       int64_t n_items, int32_t n_neg, const int64_t* hist_ptr, const int32_t* hist_items,
                       const int32_t* hist_sorted, int32_t reject_history, int32_t mask_mode, int32_t seq_last,
                       int32_t L, int64_t* item_id, int32_t* item_seq, int64_t* seq_len);
/* DEVICE sampler: same rule (uniform [1,N-1], reject positive + history, <= 100 tries, else 0) on Philox4x32-10
 * keyed by `seed` with counter (step, row, slot, try): order-independent, bit-exact vs oracle/philox_ref.py.
 * item_id int64[B, K+1] (column 0 = positive), label int32[B, K+1] = [1,0,..,0] (nullable).  hist_* nullable. */
int ur_sample_negatives(const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items,
                        int64_t n_users, const int64_t* hist_ptr, const int32_t* hist_sorted, uint64_t seed,
                        uint32_t step, int64_t* item_id, int32_t* label, void* stream);
/* Popularity-biased variant (AddNegSamples with item_popularity and neg_by_pop_alpha, unirec/data/transform/addnegsamples.py:
 * 58-62,75-80 + the alias method of unirec/utils/sampling.py:9-31): a try draws x = random() * n_items, i = int(x) and takes
 * alias_idx[i] if x - i > alias_odds[i] else i; random() is CPython's 53-bit construction from two Philox words.  Same
 * rejection rule / 100 tries / counter layout as ur_sample_negatives.  alias_odds double[n_items], alias_idx int64[n_items]
 * on the DEVICE, as built on the host by ur_alias_table_build. */
int ur_sample_negatives_pop(const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items,
                            int64_t n_users, const int64_t* hist_ptr, const int32_t* hist_sorted, const double* alias_odds,
                            const int64_t* alias_idx, uint64_t seed, uint32_t step, int64_t* item_id, int32_t* label, void* stream);
/* HOST: the alias table of prepare_aliased_randomizer (unirec/utils/sampling.py:9-24) for weights w[n]: odds[n], alias[n]
 * (-1 where the reference keeps (1, None)); same traversal order, so the same table. */
int ur_alias_table_build(const double* host_weights, int64_t n, double* host_odds, int64_t* host_alias);

/* DEVICE history rows: item_seq[b,:] = left_pad(AddUserHistory(history(user_b), ids_b), L) for a
 * whole batch from a CSR history resident in HBM -- adduserhistory.py:32-73 + seqrecdataset.py:60-68 without the
 * per-sample Python __getitem__.  hist_items: per-user interaction (time) order.  item_id int64[B,G]: the row's id group
 * (positive in column 0).  mask_mode 0 'unorder', 1 'autoregressive', 2 unchanged; seq_last as the reference's.
 * match_all 0: only the positive can occur in the history (negatives sampled with history rejection); 1: compare all G.
 * The autoregressive cut without seq_last picks occurrence (philox4x32_10(step,row,0xFFFFFFFF,0)[0] * count) >> 32
 * (bit-exact vs oracle/philox_ref.py; the host builder reproduces the reference's MT19937 stream instead).
 * item_seq int32[B,L]; seq_len int64[B] (nullable) = min(len(history'), L), 1 for users without history. */
int ur_device_build_seq(const int64_t* user_id, const int64_t* item_id, int32_t B, int32_t G, int64_t n_users,
                        const int64_t* hist_ptr, const int32_t* hist_items, int32_t mask_mode, int32_t seq_last,
                        int32_t match_all, int32_t L, uint64_t seed, uint32_t step, int32_t* item_seq, int64_t* seq_len,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same rows on the REFERENCE's random stream, on the device (round 6).  The reference draws negatives and history cuts from one
 * process-global CPython `random` (MT19937) stream, row after row (unirec/data/transform/addnegsamples.py:75-115: up to 100 x
 * random.randint(1, n_items - 1) per negative; adduserhistory.py:32-73: random.choice of the occurrence to cut at).  ur_mt_build_rows walks
 * THAT stream (csrc/mt_sampler.hip: block-parallel twist, acceptable words indexed by a prefix count, a speculative row chain, exact
 * replay of the rare rows with a rejected candidate): item_id / label equal ur_host_build_rows' and the reference DataLoader's bit for bit.
 *   state  uint32[626] device: mt[624], position (624 = twist first), sticky error (workspace exhausted); initialise it with
 *          ur_host_sampler_state of a host sampler seeded like random.seed(s).  Advanced by exactly the words the reference consumes.
 *   want_cut = 1: 'autoregressive' masking with seq_last = 0 -- choice[B] receives the index (history order) of the occurrence each row's
 *          history is cut before (-1: none), to be passed to ur_device_build_seq_choice.
 *   ws     ur_mt_workspace_bytes(B, K, n_items, want_cut) bytes. */
int ur_host_sampler_state(void* sampler, uint32_t* out625);      /* mt[624] + position of a host sampler (host memory) */
int64_t ur_mt_workspace_bytes(int32_t B, int32_t K, int64_t n_items, int32_t want_cut);
int ur_mt_build_rows(uint32_t* state, const int64_t* user_id, const int64_t* pos_item, int32_t B, int32_t K, int64_t n_items,
                     int64_t n_users, const int64_t* hist_ptr, const int32_t* hist_items, const int32_t* hist_sorted,
                     int32_t reject_history, int32_t want_cut, int64_t* item_id, int32_t* label, int32_t* choice, void* ws,
                     void* stream);
int ur_device_build_seq_choice(const int64_t* user_id, const int64_t* item_id, int32_t B, int32_t G, int64_t n_users,
                               const int64_t* hist_ptr, const int32_t* hist_items, int32_t mask_mode, int32_t seq_last,
                               int32_t match_all, int32_t L, const int32_t* choice, int32_t* item_seq, int64_t* seq_len, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The two fp32-MFMA GEMM kernels of the encoders, exposed for unit tests and micro-benchmarks.
 *   ur_gemm_nt: C[M,N] = epi( pro(A)[M,K] @ W[N,K]^T )  == nn.Linear (unirec/model/modules.py:285-287,312,347-350)
 *     pro: 0 none, 1 activation `act` on A;  epi: 0 none, 1 +bias[N], 2 LayerNorm(acc+bias+aux) (writes xhat, rstd; N<=256),
 *     3 * act'(aux), 4 + aux.   N, K and every leading dimension % 4 == 0.
 *   ur_gemm_tn: out[R,Cc] = P[T,R]^T @ pro(Q)[T,Cc], bias_out[R] = column sums of P (nullable): the weight / bias
 *     gradient of nn.Linear; deterministic split over T; ws: ur_gemm_tn_workspace_floats(T,R,Cc) floats. */
int ur_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, int pro, int epi,
               int act, const float* bias, const float* aux, int ldaux, const float* gamma, const float* beta, float eps,
               float* xhat, float* rstd, void* stream);
int64_t ur_gemm_tn_workspace_floats(int T, int R, int Cc);
/* n <= 12 such products in ONE launch (+ one deferred-reduction launch): how the encoders' backward passes issue the weight gradients
 * queued at one fork (every product fills a share of the chip; csrc/sasrec.hip).  Arrays of n; bias_out / pro_act_on_q nullable. */
int ur_gemm_tn_group(int n, const float* const* P, const int* ldp, const float* const* Q, const int* ldq, const int* T, const int* R,
                     const int* Cc, const int* pro_act_on_q, int act, float* const* out, const int* ldo, float* const* bias_out,
                     float* const* ws, void* stream);
/* Arithmetic of the dense fp32 contractions (process-wide; round 6).  0 = exact fp32-input MFMA (v_mfma_f32_32x32x2_f32, the default).
 * 6 / 9 = every fp32 operand written as the EXACT sum of three bf16 pieces and the product accumulated in fp32 from the six (nine)
 * piece products on v_mfma_f32_32x32x16_bf16 -- fp32-equivalent results (measured error vs fp64 in profiles/r06_*), 16x the matrix
 * rate per instruction.  3 = a three-term split, NARROWER than fp32, for error studies only.  Replaces nothing in the reference: it is
 * how torch's fp32 matmul (unirec/model/modules.py:285-287,312,347-355 and their autograd) is evaluated here. */
int ur_set_mfma_arith(int terms);
int ur_get_mfma_arith(void);
int ur_gemm_tn(const float* P, int ldp, const float* Q, int ldq, int T, int R, int Cc, int pro_act_on_q, int act, float* out,
               int ldo, float* bias_out, float* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Full-item ranking for the one_vs_all evaluation protocol.
 * Replaces Evaluator.evaluate_with_full_items (unirec/facility/evaluation/evaluator_abc.py:189-278: table copied to
 * numpy, user_emb @ item_emb.T on the CPU, a Python loop that sets scores[history] = -inf, column 0 overwritten
 * with the target's score) followed by numba get_rank (unirec/facility/evaluation/onepos.py:20-31:
 * rank = #{columns 1.. : score > score[0]}).  Same result without materialising the [B, n_items] scores:
 *   rank[b] = #{ n in [0,n_items), n != target_b, n != 0, n not in history(user_b) : s(b,n) > s(b,target_b) }
 *   s(b,n)  = (user_emb[b] . item_table[n] + user_bias[user_b] + item_bias[n]) / tau        (tau > 0)
 * target_score[b] = s(b, target_b).   history: CSR over users (hist_ptr[n_users+1], hist_sorted ascending within a
 * user, duplicates allowed); users outside [0,n_users) or hist_ptr == NULL have no history.
 * rank: int32[B]; thr_ws: B floats of scratch.  d % 4 == 0, d <= 512. */
int ur_full_rank(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                 const int64_t* target, const int64_t* user_id, const int64_t* hist_ptr, const int32_t* hist_sorted,
                 int64_t n_users, const float* user_bias, const float* item_bias, float tau, int32_t* rank,
                 float* target_score, float* thr_ws, void* stream);

/* The same count over a ROW-SHARDED catalogue ("full-item eval shards the [B,N] score matmul by the same
 * row ownership"): item i lives on rank i % W at local row i / W + 1, local row 0 is each shard's padding row.  Every rank holds
 * the user vectors of ALL ranks (all-gather) and runs, on its own shard,
 *   phase 1: thr[b] = u_b . E_local[local_target[b]] + item_bias_local[...]   (0 when local_target[b] < 0: not this rank's row)
 *            -> all-reduce(sum) of thr: every rank now holds the target scores, each produced by the MFMA sequence of the shard
 *               that owns the target, so phase 2 on that shard never counts the target itself;
 *   phase 2: rank_partial[b] = #{ local rows n >= 1, n != local_target[b], n not in local history(user_b) : s(b,n) > thr[b] }
 *            -> all-reduce(sum) of rank_partial = the rank over the whole catalogue.
 * hist_sorted_local: the users' histories restricted to this rank's items, in LOCAL row ids, ascending (CSR hist_ptr).
 * n_local: rows to scan = this rank's last valid local row + 1 (allocated rows beyond it are ignored); excl_row: one more
 * local row that is not an item and must not count (rank 0's local row 1, the slot global id 0 would take), or -1.
 * No collective inside: the two all-reduces are the caller's (torch.distributed over RCCL). */
int ur_full_rank_shard(int32_t phase, const float* user_emb, const float* shard_table, int64_t n_local, int32_t B, int32_t d,
                       const int64_t* local_target, const int64_t* user_id, const int64_t* hist_ptr,
                       const int32_t* hist_sorted_local, int64_t n_users, const float* item_bias_local, int64_t excl_row,
                       float* thr, int32_t* rank_partial, void* stream);

/* Full-item top-k retrieval -- BaseRecommender.topk with candidates=None (unirec/model/base/recommender.py:149-197:
 * scores of all items, all_scores[row, user_hist] = -inf, torch.topk) and the scoring loop of main/reco_topk.py:22-96.
 *   topk_scores[b, :], topk_ids[b, :] = the k best items of row b by s(b,n) = (u_b . E_n + user_bias + item_bias[n]) / tau,
 *   excluding item 0 (the padding row) and history(user_b) (CSR as in ur_full_rank), sorted by (score desc, id asc);
 *   rows with fewer than k admissible items are padded with (-inf, -1).  k <= 1024.  The [B, n_items] score matrix is
 * only ever materialised one 2^20-item chunk at a time (ws: ur_full_topk_workspace_bytes). */
int64_t ur_full_topk_workspace_bytes(int32_t B, int64_t n_items, int32_t k);
int ur_full_topk(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d, int32_t k,
                 const int64_t* user_id, const int64_t* hist_ptr, const int32_t* hist_sorted, int64_t n_users,
                 const float* user_bias, const float* item_bias, float tau, float* topk_scores, int64_t* topk_ids, void* ws,
                 void* stream);

/* ---------------------------------------------------------------------------------------------
 * fullsoftmax training loss: every item is a candidate.
 * Replaces BaseRecommender.forward with loss_type 'fullsoftmax' (unirec/model/base/recommender.py:46-55) + _cal_loss
 * (unirec/model/base/reco_abc.py:266-270):  loss = mean_b( logsumexp_n s(b,n) - s(b,target_b) ) over ALL n in [0,n_items),
 * s(b,n) = (u_b . E_n + user_bias[user_b] + item_bias[n]) / tau, clamped to +-score_clip when score_clip > 0.
 *   fwd: target_score[B] = s(b,target_b) (from ur_gather_dot_loss_fwd with UR_LOSS_NONE); writes lse[B], loss_out[4] (as ur_gather_dot_loss_fwd).
 *   bwd: d_user_emb [B,d]; d_item_table [n_items,d] DENSE and overwritten (row 0 = 0: padding_idx); d_item_bias [n_items]
 *        (required iff item_bias); d user_bias is identically 0.  d_loss: device scalar (nullable = 1).
 * The [B, n_items] scores exist one 2^20-item chunk at a time (ws: ur_full_softmax_workspace_bytes). */
int64_t ur_full_softmax_workspace_bytes(int32_t B, int32_t d, int64_t n_items);
int ur_full_softmax_fwd(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                        const int64_t* target, const int64_t* user_id, const float* user_bias, const float* item_bias,
                        float tau, float score_clip, const float* target_score, float* lse, float* loss_out, void* ws,
                        void* stream);
int ur_full_softmax_bwd(const float* user_emb, const float* item_table, int64_t n_items, int32_t B, int32_t d,
                        const int64_t* target, const int64_t* user_id, const float* user_bias, const float* item_bias,
                        float tau, float score_clip, const float* lse, const float* d_loss, float* d_user_emb,
                        float* d_item_table, float* d_item_bias, void* ws, void* stream);
/* fullsoftmax over a ROW-SHARDED catalogue (the loss of the reference's own multi-GPU test,
 * tests/test_model/run_ddp_test.sh:28 loss_type='fullsoftmax' under accelerate/DDP: unirec/facility/trainer.py:67,346).  Every rank
 * holds the user vectors of ALL ranks (all-gather, B = all columns) and scores them against the n_rows rows it owns:
 *   fwd_shard      : part3[3][B] = per column (running max, sum-exp, score of the positive if target_row[b] >= 0 else 0) over those rows;
 *   combine_shards : the all-gathered parts[world][3][B_all] folded in rank order -> lse[B_all] (identical on every rank) and
 *                    loss_out[4] = [mean over this rank's columns col0 .. col0 + B_own of lse - s_target, B_own, update guard, -];
 *   bwd_shard      : as ur_full_softmax_bwd on the shard's rows with the GLOBAL lse: d_shard_rows [n_rows,d] is this rank's slice of the
 *                    table gradient summed over every rank's columns (no exchange needed), d_user_emb [B,d] the partial sum over this
 *                    rank's items (all-reduce / reduce-scatter it); scale 1 / (B tau) times *d_loss; zero_row0: row 0 of shard_rows is
 *                    global row 0 (padding_idx) and gets a zero gradient. */
int ur_full_softmax_fwd_shard(const float* user_emb, const float* shard_rows, int64_t n_rows, int32_t B, int32_t d,
                              const int64_t* target_row, const int64_t* user_id, const float* user_bias,
                              const float* item_bias_rows, float tau, float score_clip, float* part3, void* ws, void* stream);
int ur_full_softmax_combine_shards(const float* parts, int32_t world, int32_t B_all, int32_t col0, int32_t B_own, float* lse,
                                   float* loss_out, void* stream);
int ur_full_softmax_bwd_shard(const float* user_emb, const float* shard_rows, int64_t n_rows, int32_t B, int32_t d,
                              const int64_t* target_row, const int64_t* user_id, const float* user_bias, const float* item_bias_rows,
                              float tau, float score_clip, const float* lse, const float* d_loss, float* d_user_emb,
                              float* d_shard_rows, float* d_item_bias_rows, int32_t zero_row0, void* ws, void* stream);
/* dense[uniq_idx[u], :] += rows[u, :] for u < *n_uniq_dev (unique rows: no conflicts) -- folds the encoder's row-sparse
 * gradient into the dense table gradient of a fullsoftmax step. */
int ur_rows_scatter_add(const int32_t* uniq_idx, const int32_t* n_uniq_dev, int64_t n_max, const float* rows, int32_t d,
                        float* dense, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Live profiler (measurement only; no reference counterpart).  While enabled, every launch group is
 * bracketed by HIP events on its stream.  ur_prof_read fills three host arrays of ur_prof_num_classes()
 * entries: summed milliseconds, number of launch groups, and summed algorithmic work (flops for the GEMM
 * classes, bytes for the HBM-bound classes; definitions in DESIGN.md section 6). */
int ur_prof_enable(int on);
int ur_prof_set_mask(uint32_t class_mask); /* bit c = bracket kernel class c (default: all); fewer events = less perturbation */
int ur_prof_reset(void);
int ur_prof_num_classes(void);
const char* ur_prof_class_name(int cls);
int ur_prof_read(double* host_ms, int64_t* host_count, double* host_work);

#ifdef __cplusplus
}
#endif
#endif /* UNIREC_AMD_H */
