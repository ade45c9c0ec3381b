// Internal launcher declarations shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace ur {

constexpr int LN_BWD_MAX_BLOCKS = 512;

// ---- rowops.hip
// n_rows (here and below; 0 = unknown): rows of the table -- what the bounds-checked build (common.h: UR_ROW) checks every index against
int gather_rows(const float* table, const void* idx, int idx_bytes, long long n, int d, float* out, hipStream_t st, long long n_rows = 0);
// tok (nullable): compact row r is token tok[r] of seq (= b*L + l); m_dev (nullable): device-side row count
int embed_ln_fwd(const int* seq, const float* table, const float* pos, const float* gamma, const float* beta, float eps,
                 int M, int L, int d, float* y, float* xhat, float* rstd, hipStream_t st, const int* tok = nullptr,
                 const int* m_dev = nullptr, const DropSpec* drop = nullptr, long long n_rows = 0);   // drop: y <- dropout(y), row id = token b*L + l
// out[r,:] = dropout(x[r,:]), row id of row r per `drop` (out may alias x; drop off: a copy, or nothing when aliased)
int drop_rows(const float* x, long long rows, int d, const DropSpec& drop, float* out, hipStream_t st);
int ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps, int M, int d, float* y,
           float* xhat, float* rstd, hipStream_t st);
// part_ws: LN_BWD_MAX_BLOCKS * 2 * d floats
// Deferred split reductions.  The weight / bias / LayerNorm-affine gradients are only consumed by the optimizer, so
// the second stage of every split reduction of a backward pass (out[i] = sum_s part[s*stride + i], fixed order) is
// queued here and executed by ONE launch at the end (reduce_batch) instead of one small launch per GEMM / LayerNorm.
struct ReduceItem {
  const float* part; float* out;
  long long stride, n;      // n elements (multiple of 4); partial s starts at part + s*stride
  int S, cols, ldo;         // element i goes to out[(i / cols) * ldo + i % cols]
  int first_block;
};
struct ReduceBatch {
  static constexpr int MAX = 48;
  ReduceItem item[MAX]; int n = 0;
  // host side only: a batch that is full goes on in `next` (one launch per link at the end).  A backward pass whose reductions are
  // deferred to the side stream must not flush in the middle of the pass -- deeper models (> 2 layers: > 48 items) chain batches.
  ReduceBatch* next = nullptr;
  bool full(int need) const {
    int room = 0;
    for (const ReduceBatch* b = this; b; b = b->next) room += MAX - b->n;
    return room < need;
  }
  void add(const float* part, long long stride, int S, long long n_el, int cols, float* out, int ldo) {
    ReduceBatch* b = this;
    while (b->n >= MAX && b->next) b = b->next;
    b->item[b->n++] = ReduceItem{part, out, stride, n_el, S, cols, ldo, 0};
  }
};
int reduce_batch(ReduceBatch& rb, hipStream_t st);   // runs and empties the queue (every link of the chain)

// defer != nullptr: part_ws must stay untouched until reduce_batch(*defer) has run
// m_dev (nullable): device-side row count (M is then the maximum); out_rows (nullable): dx row r is written to row
// out_rows[r] of dx (compact -> padded layout)
int ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, const float* add_in,
           const int* seq, int M, int d, float* dx, float* dgamma, float* dbeta, float* part_ws, hipStream_t st,
           ReduceBatch* defer = nullptr, const int* m_dev = nullptr, const int* out_rows = nullptr,
           const DropSpec* in_drop = nullptr, const DropSpec* out_drop = nullptr, float* dx_drop = nullptr);
// in_drop: dy is the gradient of dropout(LN(.)) -- it is masked on read.  out_drop + dx_drop: the LayerNorm input was
// dropout(h) + residual -- dx (gradient of the sum, for the residual branch) AND dx_drop = dropout-masked dx (gradient of h).
int pos_grad(const float* dx, int B, int L, int d, float* dpos, hipStream_t st);

// ---- gemm.hip
// C[M,N] = epi( pro(A)[M,K] @ W[N,K]^T )
enum GemmPro { PRO_NONE = 0, PRO_ACT = 1 };
enum GemmEpi {
  EPI_NONE = 0,      // C = acc
  EPI_BIAS = 1,      // C = acc + bias[n]
  EPI_BIAS_RES_LN = 2,  // t = acc + bias[n] + res[m,n]; C = LN(t)*gamma+beta; also xhat, rstd
  EPI_MUL_DACT = 3,  // C = acc * act'(aux[m,n])
  EPI_ADD = 4,       // C = acc + aux[m,n]
  EPI_COUNT_GT = 5,  // nothing stored: ((int*)C)[m] += #{n != skip[m] : acc + bias[n] > aux[m]}   (full-item ranking)
  EPI_ADD_LNBWD = 6, // t = acc (+ aux[m,n]): the gradient wrt a LayerNorm OUTPUT; C[out_rows[m]] = LayerNorm backward of t given the
                     // saved xhat / rstd and gamma (N <= 128 = one tile row); per-M-tile (d gamma | d beta) partial sums -> ln_part
};
struct GemmArgs {
  const float* A; int lda;
  const float* W; int ldw;
  float* C; int ldc;
  int M, N, K;
  int act;              // for PRO_ACT / EPI_MUL_DACT
  const float* bias;    // [N]
  const float* aux; int ldaux;   // residual / addend / pre-activation
  const float* aux2; int ldaux2; // EPI_ADD: optional second addend (nullable)
  const float* gamma; const float* beta; float eps;
  float* xhat; float* rstd;      // EPI_BIAS_RES_LN outputs
  const int* m_dev;              // nullable: device-side row count, M = min(M, *m_dev) (compacted token rows)
  DropSpec drop;                 // EPI_BIAS_RES_LN: t = dropout(acc + bias) + res  (thresh == 0: off)
  const long long* skip; long long skip_base;   // EPI_COUNT_GT: column skip[m] - skip_base of row m is left out (nullable)
  // EPI_ADD with out_rows: C[out_rows[m], :] += acc + aux[m, :] (rows of out_rows distinct).
  // EPI_ADD_LNBWD: xhat / rstd are INPUTS here (saved by the forward pass); out_rows (nullable) scatters row m of the result to row
  // out_rows[m] of C; ln_part [gemm_nt_lnbwd_tiles(M)][2 N] receives the partial sums (every slot is written); M_host = the M
  // the grid was sized for (set by gemm_nt)
  const int* out_rows; float* ln_part; int M_host;
  // ksplit > 1 (EPI_NONE only): the K dimension is cut into ksplit pieces, piece s is one grid row and writes ITS partial product to
  // C + s * split_stride; the consumer sums the pieces in order.  For few-row, long-K products (the GRU's per-step GEMMs at H = 768:
  // one workgroup per CU walking 24-72 dependent K-steps) this is what puts several workgroups on a CU.
  int ksplit; long long split_stride;
};
int gemm_nt_lnbwd_tiles(int M);   // M-tiles (= partial-sum rows) of an EPI_ADD_LNBWD launch over M rows
int gemm_nt(const GemmArgs& a, int pro, int epi, hipStream_t st);

// Out[R,Cc] = P[T,R]^T @ pro(Q)[T,Cc];  bias_out[R] = colsum(P) (nullable). Deterministic split over T.
// ws: gemm_tn_ws_floats(R, Cc) floats.
long long gemm_tn_ws_floats(int T, int R, int Cc);
int gemm_tn(const float* P, int ldp, const float* Q, int ldq, int T, int R, int Cc, int pro_act_on_q, int act,
            float* out, int ldo, float* bias_out, float* ws, hipStream_t st, ReduceBatch* defer = nullptr,
            const int* t_dev = nullptr);   // t_dev (nullable): device-side token count, T = min(T, *t_dev)
// Several such products in ONE launch (gemm_tn_group_kernel): the chip is filled across products, so each needs few token splits.
struct TnReq {
  const float* P; int ldp; const float* Q; int ldq; int T, R, Cc, pro_act, act;
  float* out; int ldo; float* bias_out; float* ws; const int* t_dev;   // ws: gemm_tn_group_ws_floats(R, Cc) floats
};
long long gemm_tn_group_ws_floats(int R, int Cc);
int gemm_tn_group(const TnReq* req, int n, hipStream_t st, ReduceBatch* defer = nullptr);
// arithmetic of the dense contractions: 0 = exact fp32 MFMA, 6 / 9 = split-bf16 terms (fp32-equivalent), 3 = error studies only
int mfma_arith();
int set_mfma_arith(int terms);
// the arithmetic of ONE encoder call (its cfg's mfma_arith field) for every product the calling thread launches inside the scope
struct ArithScope { int prev; explicit ArithScope(int terms); ~ArithScope(); ArithScope(const ArithScope&) = delete; ArithScope& operator=(const ArithScope&) = delete; };
// dst[c,r] = src[r,c]
int transpose(const float* src, int rows, int cols, float* dst, hipStream_t st);
struct TransposeItem { const float* src; float* dst; int rows, cols, first_block, mode; };   // mode 0: transpose; 1 / 2 (| 4): split copy (add_split)
struct TransposeBatch {
  static constexpr int MAX = 48;
  TransposeItem item[MAX]; int n = 0;
  float* zero_ptr = nullptr; long long zero_n = 0; int zero_first_block = 0;   // optional: also zero-fill a buffer (n % 4 == 0)
  float* zero2_ptr = nullptr; long long zero2_n = 0; int zero2_first_block = 0; // ... and a second one
  const int* zero2_pad = nullptr; int zero2_L = 0, zero2_d = 0;                  // ... of [B, L, d] rows: only rows (b, l < zero2_pad[b]) are zeroed (nullptr: all)
  const int* copy_src = nullptr; int* copy_dst = nullptr;                        // optional rider: one int copied (the backward's own copy of the valid-row count)
  bool add(const float* src, int rows, int cols, float* dst) {
    if (n >= MAX) return false;
    item[n++] = TransposeItem{src, dst, rows, cols, 0, 0};
    return true;
  }
  // the SPLIT copy the row-chain kernels stream in split-bf16 arithmetic (rowchain.hip: RcW<true>) of the K-major matrix Wt [K, N]:
  // Wt = src [rows = K, cols = N] (as_stored) or Wt = src^T with src [rows = N, cols = K].  dst: 3/2 K N floats.
  // lay = 16 (K % 16 == 0): [K/16][piece][k group of 2][n][8 bf16 = k 16 kb + 4 g + {0..3}, 16 kb + 8 + 4 g + {0..3}]: v_mfma_f32_32x32x16_bf16
  // fragments in the row chains' k order; lay = 32 (K % 32 == 0): [K/32][piece][k group of 4][n][8 consecutive k]: v_mfma_f32_16x16x32_bf16 (gru.hip)
  bool add_split(const float* src, int rows, int cols, float* dst, bool as_stored, int lay = 16) {
    if (n >= MAX) return false;
    item[n++] = TransposeItem{src, dst, rows, cols, 0, (as_stored ? 1 : 2) | (lay == 32 ? 4 : 0)};
    return true;
  }
};
int transpose_batch(TransposeBatch& tb, hipStream_t st);   // every queued transpose in one launch

// ---- rowchain.hip: row-block chain kernels (a workgroup carries BM token rows through a sequence of GEMMs, tiles in LDS)
constexpr int CHAIN_FWD = 1, CHAIN_BWD = 2, CHAIN_PROJ = 4, CHAIN_LAST = 8, CHAIN_LAST_BWD = 16, CHAIN_EMBED = 32, CHAIN_LASTROW = 64, CHAIN_ALL = 127;   // LAST*: the B last rows of the last-row layer through the row-chain kernels; EMBED: lookup + LN + first projection; LASTROW: the last-row layer as two launches (lastrow.hip; wins over LAST*)
constexpr int CHAIN_DEFAULT = CHAIN_ALL;  // which chain kernels run by default (test hook chain_mask / ur_sasrec_set_chain: bit mask); DESIGN.md 6d
bool chain_supported(int d, int inner, int which);
bool chain_shape_ok(int d, int inner);              // the kernels exist for this shape (whatever the switch says)   // d in {32, 64, 128}, inner % d == 0, and the bit(s) `which` switched on
int chain_rows_per_block(int d);
int chain_set_enabled(int mask);          // mask < 0: query only; returns the previous mask
struct ChainFwdArgs {
  const float* ctx; int ldctx;          // attention output [M, d]
  const float* res; int ldres;          // residual of the attention block = the layer input x
  // Weights come TRANSPOSED ([in, out]: the copies ur_sasrec_fwd makes at its head): a lane of the B operand owns an output column, so
  // the 32 lanes of a half-wave read 128 contiguous bytes of one k-row (lane = row of the nn.Linear layout, 16 B each, was 64 separate
  // 16-byte accesses per wave instruction: 35 GB/s per workgroup against 80-115 coalesced, tools/probe/wstream_probe.hip)
  const float *woT, *bo, *g1, *b1ln;    // out-projection^T [d,d] + its LayerNorm
  const float *w1T, *b1, *w2T, *b2;     // dense_1^T [d,I], dense_2^T [I,d]
  const float *g2, *b2ln;
  float *a, *ahat, *rstd1, *h1, *y, *yhat, *rstd2;
  float* u;                             // nullable: act(h1) [M, I], saved for the dense_2 weight-gradient GEMM (no activation recompute there)
  const float *wnT, *bn; float* outn; int ldn, Nn, ldwn;   // optional: next projection y Wn^T + bn; WnT [d, ldwn] (columns = outputs), Nn % d == 0
  int M; const int* m_dev;
  int I, act; float eps;
  DropSpec drop_out, drop_ffn;
  float* split_part = nullptr;          // chain_ffn_fwd_split: [row blocks][I/d][rows per block][d] partial dense_2 outputs
  unsigned* split_cnt = nullptr;        //   and one completion counter per row block (zero on entry, reset by the kernel)
  int arrive_mode = 0;                  //   ur_arrive_mode(): memory order of the arrival / test skew (common.h)
  bool wsplit = false;                  // the weight pointers name SPLIT copies (three bf16 pieces per weight, rowchain.hip: RcW<true>): split-bf16 arithmetic
};
struct ChainBwdArgs {
  const float* gy;                      // d loss / d y  [M, d]
  const float *yhat, *rstd2, *g2;       // feed-forward LayerNorm
  const float* h1;                      // [M, I] pre-activation
  const float* w2;                      // [d, I]  dense_2.weight as stored: g_u = g_tf W2 contracts over its rows (coalesced B operand, see ChainFwdArgs)
  const float* w1;                      // [I, d]  dense_1.weight as stored
  const float *ahat, *rstd1, *g1;       // attention LayerNorm
  const float* wo;                      // [d, d]  dense.weight as stored
  float *g_tf, *g_h1, *g_ta, *g_ctx;    // outputs (g_tf, g_h1, g_ta feed the weight-gradient GEMMs)
  float* part;                          // [workgroups][4 d]: d gamma2 | d beta2 | d gamma1 | d beta1 partial sums
  int M; const int* m_dev;
  int I, act;
  float* split_part = nullptr;          // chain_ffn_bwd_split: [row blocks][I/d][rows per block][d] partial d a tiles
  unsigned* split_cnt = nullptr;        //   and one completion counter per row block (zero on entry, reset by the kernel)
  int arrive_mode = 0;                  //   ur_arrive_mode(): memory order of the arrival / test skew (common.h)
  // hidden dropout (chain_ffn_bwd only; thresh == 0: off): the masks of the forward pass's two sites.  The residual branches take the
  // unmasked g_tf / g_ta; the GEMMs (and the weight-gradient products) take the masked copies, written to g_tfd / g_tad
  DropSpec drop_ffn = {}, drop_out = {};
  float *g_tfd = nullptr, *g_tad = nullptr;
  bool wsplit = false;                  // the weight pointers name SPLIT copies (three bf16 pieces per weight, rowchain.hip: RcW<true>): split-bf16 arithmetic
};
struct ChainProjBwdArgs {
  const float* g; int ldg; int K;       // [M, K] gradient of the projection output; K % d == 0
  const float* w; int ldw;              // [K, ldw] projection weight as stored: out[m, n] = sum_k g[m, k] w[k, n]
  const float* res;                     // nullable addend [M, d]
  const float *xhat, *rstd, *gamma;     // nullable: LayerNorm backward applied to the result (embedding LayerNorm)
  float* out; const int* out_rows;      // row m of the result goes to row out_rows[m] (nullable: m) of out [., d]
  float* part;                          // with LayerNorm: [workgroups][2 d] d gamma | d beta partial sums
  int M; const int* m_dev;
  DropSpec drop = {};                   // with LayerNorm: x0 = dropout(LN0(.)) -- the result is masked before the LayerNorm backward (thresh 0: off)
  bool wsplit = false;                  // the weight pointers name SPLIT copies (three bf16 pieces per weight, rowchain.hip: RcW<true>): split-bf16 arithmetic
};
struct ChainEmbedArgs {
  const int* seq;                       // [B*L] item ids of the padded token grid
  long long n_rows = 0;                 // rows of the item table (bounds-checked build)
  const float *table, *pos;             // item table [N, d]; position table [L, d] (nullable)
  const float *g0, *b0ln; float eps;    // the embedding LayerNorm
  int L; const int* tok;                // tok (nullable): buffer row -> token of the padded grid (compacted rows)
  DropSpec drop;                        // embedding dropout (keyed by the token)
  float *x0, *x0hat, *rstd0;            // outputs: the layer input [M, d], its normalised copy and 1/std (for the backward)
  const float *wnT, *bn; float* outn; int ldn, Nn, ldwn;   // first projection: x0 Wn^T + bn; WnT [d, ldwn] (columns = outputs), Nn % d == 0
  int M; const int* m_dev;
  bool wsplit = false;                  // the weight pointers name SPLIT copies (three bf16 pieces per weight, rowchain.hip: RcW<true>): split-bf16 arithmetic
};
int chain_embed_proj(const ChainEmbedArgs& a, int d, hipStream_t st);   // lookup + position + LayerNorm + the first layer's Q/K/V projection
int chain_ffn_fwd(const ChainFwdArgs& a, int d, hipStream_t st);
// the same block for FEW rows (the B last rows): the inner dimension is split over I/d workgroups per row block, the LayerNorm behind
// dense_2 is done by the workgroup of a row block that finishes last.  Needs cdiv(M, rows per block) <= CHAIN_SPLIT_MAX_BLOCKS.
constexpr int CHAIN_SPLIT_MAX_BLOCKS = 64;
int chain_ffn_fwd_split(const ChainFwdArgs& a, int d, hipStream_t st);
int chain_ffn_bwd_split(const ChainBwdArgs& a, int d, hipStream_t st);   // part: [cdiv(M, rows per block)][4 d]
long long chain_split_part_floats(int M, int d, int inner);
int chain_ffn_bwd(const ChainBwdArgs& a, int d, hipStream_t st);     // workgroups = cdiv(M, chain_rows_per_block(d))
int chain_proj_bwd(const ChainProjBwdArgs& a, int d, hipStream_t st);

// ---- lastrow.hip: the last-row layer (last_only: only position L-1 of the top layer reaches the loss) as two launches
struct LastRowFwdArgs {
  const float* x; const int* xrow; long long xstride, xoff;   // layer input [., d]: the row of sequence b is xrow ? xrow[b] : b * xstride + xoff
  const float* qkv;                     // [M, 3d]: this layer's K | V rows in columns d .. 3d (written by the layer below's chain kernel)
  const int* seq;                       // [B, L] item ids (the key mask)
  const int *seq_base, *seq_pad;        // compacted rows (nullable): position l of sequence b is row seq_base[b] + l, l >= seq_pad[b]
  const float* wqT; int ldq;            // K-major query weight: [d][ldq], columns 0 .. d-1 (= the layer's wqkvT)
  const float *bq, *woT, *bo, *g1, *b1ln, *w1T, *b1, *w2T, *b2, *g2, *b2ln;   // K-major copies (kernels.h: ChainFwdArgs) and the vectors
  float *q_out, *x_out;                 // [B, d] queries (the backward reads them); x_out (nullable): the gathered input rows
  float *ctx, *lse;                     // [B, d] attention output, [B, H] log-sum-exp
  float *a, *ahat, *rstd1, *h1, *y, *yhat, *rstd2;
  int B, L, I, act; float eps, scale, sqrt_hd;
  DropSpec drop_out, drop_ffn;          // hidden dropout sites (row id of row b per the spec)
  unsigned dkey, dthresh; float dscale; // dropout on the attention probabilities (dthresh == 0: off): row id (b * H + h) * L + L - 1, column j
  int part_floats;                      // (set by the launcher)
  unsigned long long* trace;            // (set by the launcher: phase stamps, ur_debug_lr_trace)
};
struct LastRowBwdArgs {
  const float* gy;                      // [B, d] d loss / d y
  const float *yhat, *rstd2, *g2, *h1;  // saved by the forward
  const float *w2, *w1, *wo, *wqkv;     // weights AS STORED: [d, I], [I, d], [d, d], [3d, d] (K-major for the gradient products)
  const float *ahat, *rstd1, *g1;
  const float *q, *ctx, *lse;           // [B, d], [B, d], [B, H]
  const float* qkv; const int* seq; const int *seq_base, *seq_pad;
  float *g_tf, *g_tfd, *g_h1, *g_ta, *g_tad, *dq;   // [B, .] operands of the weight-gradient products (g_tfd / g_tad: dropout-masked copies, = g_tf / g_ta without)
  float* g_qkv;                         // [M, 3d]: dK | dV of every row into columns d .. 3d
  float* g_x;                           // [M, d]: the layer's input gradient, EVERY row of every sequence written
  float* part;                          // [workgroups][4 d]: d gamma2 | d beta2 | d gamma1 | d beta1 partial sums
  int B, L, I, act; float scale, sqrt_hd;
  DropSpec drop_ffn, drop_out; unsigned dkey, dthresh; float dscale;
  // riders (extra workgroups behind the B / 4 that do the work; all optional): what the backward pass zero-fills before anything else --
  // zero_ptr[0 .. zero_n) (the dense-gradient buffer), the rows (b, l < zero2_pad[b]) of zero2_ptr [., zero2_L, zero2_d] (zero2_pad
  // null: all zero2_n floats; counts multiples of 4) -- and one int copied
  float* zero_ptr; long long zero_n;
  float* zero2_ptr; long long zero2_n; const int* zero2_pad; int zero2_L, zero2_d;
  const int* copy_src; int* copy_dst;
  int part_floats, scr_floats, n_main;  // (set by the launcher)
  unsigned long long* trace;
};
bool lastrow_shape_ok(int B, int L, int d, int H, int inner);
bool lastrow_supported(int B, int L, int d, int H, int inner);   // shape ok and switched on (CHAIN_LASTROW of the chain mask)
int lastrow_rows_per_block();
int lastrow_fwd(const LastRowFwdArgs& a, int d, int H, hipStream_t st);
int lastrow_bwd(const LastRowBwdArgs& a, int d, int H, hipStream_t st);

// ---- attention.hip
long long attn_lse_floats(int B, int H, int L);
// Compacted token rows (padding skipped): seq_base / seq_pad (nullable, int[B]): position l of sequence b lives in row
// seq_base[b] + l of qkv / ctx / dqkv for l >= seq_pad[b]; the padded prefix has no rows.  Supported by the MFMA
// kernels (L <= 64, head dim 4/8/16: attn_compact_supported) and the last-row kernels.
bool attn_compact_supported(int L, int d, int H);
// drop (nullable): dropout on the probabilities (row id = (b*H + h)*L + query position, column = key position)
int attn_fwd(const float* qkv, const int* seq, int B, int L, int d, int H, int causal, float* ctx, float* lse,
             int q_last_only, hipStream_t st, const int* seq_base = nullptr, const int* seq_pad = nullptr,
             const DropSpec* drop = nullptr);
long long attn_bwd_ws_floats(int B, int H, int L);
int attn_bwd(const float* qkv, const int* seq, const float* ctx, const float* dctx, const float* lse, int B, int L,
             int d, int H, int causal, float* dqkv, float* ws, int q_last_only, hipStream_t st,
             const int* seq_base = nullptr, const int* seq_pad = nullptr, const DropSpec* drop = nullptr);

// QProj (optional): the query rows are not given but PROJECTED by the kernel -- q[b] = x[xrow ? xrow[b] : b * xstride + xoff] Wq^T + bq,
// each wave its head's HD outputs -- and written to q_out (= the q_last the backward reads): the row gather and the B x d x d
// projection GEMM in front of the one-query attention (5 + 8 us at their launch floors) disappear
struct AttnQProj {
  const float* x; const int* xrow; long long xstride, xoff;   // layer input rows [., d]
  const float *wq, *bq;                                        // query.weight [d, d] (row = output feature), query.bias
  float* q_out;                                                // [B, d]
  float* x_out;                                                // nullable [B, d]: the gathered rows (each wave copies its head's slice)
};
int attn_last_fwd(const float* q_last, const float* qkv, const int* seq, int B, int L, int d, int H, float* ctx_last,
                  float* lse_last, hipStream_t st, const int* seq_base = nullptr, const int* seq_pad = nullptr,
                  const DropSpec* drop = nullptr, const AttnQProj* qp = nullptr);
int attn_last_bwd(const float* q_last, const float* qkv, const int* seq, const float* ctx_last, const float* dctx_last,
                  const float* lse_last, int B, int L, int d, int H, float* dq_last, float* dqkv, hipStream_t st,
                  const int* seq_base = nullptr, const int* seq_pad = nullptr, const DropSpec* drop = nullptr);

}  // namespace ur
