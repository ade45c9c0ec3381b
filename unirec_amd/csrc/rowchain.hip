// Row-block "chain" kernels of the SASRec layer: a workgroup owns BM token rows and carries them through a whole
// SEQUENCE of dense contractions; the intermediate [BM, d] / [BM, d-chunk of inner] tiles never leave LDS.
//
//   chain_ffn_fwd  (unirec/model/modules.py:312-316 + 347-355, one launch instead of three GEMMs):
//        a  = LN(drop(ctx Wo^T + bo) + x)                     -> a, ahat, rstd1
//        h1 = a W1^T + b1                                     -> h1 (pre-activation, kept for the backward pass)
//        y  = LN(drop(act(h1) W2^T + b2) + a)                 -> y, yhat, rstd2
//        [next layer's projection  y Wn^T + bn                -> its qkv buffer]
//   chain_ffn_bwd  (the mirror image: two LayerNorm backwards + three activation-gradient GEMMs in one launch):
//        g_tf = LNbwd(g_y)   g_h1 = (g_tf W2) * act'(h1)   g_a = g_h1 W1 + g_tf   g_ta = LNbwd(g_a)   g_ctx = g_ta Wo
//   chain_proj_bwd : g_x = g_qkv Wqkv + g_ta, followed (bottom layer) by the backward of the embedding LayerNorm, rows
//        scattered straight into the padded [B*L, d] row-gradient buffer.
//
// Why: at M = 21 360 token rows the unfused layer is ~12 launches of 1-3 GFLOP each; every one pays a cold prologue, a
// store burst that all of its workgroups issue together, and re-reads what the previous launch just wrote ([M, inner]
// twice per direction).  Here the only HBM traffic is what the backward pass / the weight-gradient GEMMs need later.
//
// Geometry (D = d in {32, 64, 128}): BM = 4096 / D rows per workgroup, 4 waves, one 32 x 32 accumulator tile per wave and
// GEMM (v_mfma_f32_32x32x2_f32: exact fp32), weight slices streamed two ahead across GEMM boundaries (the stream never drains inside a
// workgroup) straight from global memory into the owning wave's registers (below).  inner is walked in D-wide chunks: h1 chunk -> LDS ->
// act -> second GEMM accumulates y over the chunks.  LDS: the two activation tiles, 32 KB at D = 128 (unpadded, XOR-swizzled).
// REGISTERS decide the residency: <= 168 per lane = three workgroups per CU = 768 slots for the 668 row blocks of the headline batch
// (tools/occupancy.sh).  The backward kernel at 196 registers ran them in two rounds, the second 30 % full: 103 us; bounded to 168
// (__launch_bounds__(256, 3)): 88 us, the whole step - 12 us.  (Element-wise epilogues done on the accumulator registers where they are
// -- no LDS round trip, one barrier less per chunk -- were measured and dropped: the 16 dword stores per lane they need are slower than
// the transposed float4 stores, + 4 us per kernel.)  In the exact arithmetic (SP = false: mfma_arith = 0) the K order of every
// contraction equals gemm_nt's, so forward results are bit-identical to the unfused path.
// Round 6c: SP = true (the default arithmetic, mfma_arith = 6 / 9) runs every product of the chain as six bf16 piece products of exactly
// split fp32 operands on v_mfma_f32_32x32x16_bf16 (gemm.hip: fp32-equivalent, error-gated against fp64): the streamed weights are
// pre-split copies (RcW<true> below), the activation fragment is split in registers by the wave that consumes it.  Same tiles, same
// epilogues, same residency; 188 -> 152 us for the four kernels of a step (profiles/r06_l_chain_split.txt).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace ur {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float fx4 __attribute__((ext_vector_type(4)));   // plain LLVM vector: register arrays of it are always promoted (HIP's float4 class is not, in every context)

// The weight slices are NOT staged through LDS.  A wave owns 32 output columns of every GEMM of the chain (WC = D / 32
// waves side by side), so at D = 128 no two waves of a workgroup want the same weight columns: each wave loads ITS fragment of a slice
// straight from global memory (the weights are L2-resident) into registers, two slices ahead -- no staging stores, and ONE barrier per
// GEMM segment (when the activation tile changes hands) instead of one per K slice.  LDS = the two activation tiles (32 KB at D = 128).
//
// The weight operand is K-MAJOR: Wt [K, ldt], element (k, n) = weight of output column n (round 4).  A lane of the B fragment owns
// column n = its wave's block + lane & 31 and k offset 4 (lane >> 5): one dword per k, so the 32 lanes of a half-wave read 128
// CONTIGUOUS bytes of k-row.  Rounds 2-3 read the nn.Linear layout (row = output column, a float4 along k per lane): 64 separate 16-byte
// accesses per wave instruction, which the texture path serves at ~1 lane per clock -- 35 GB/s per workgroup whatever the number of
// workgroups, against 80 (dword) to 115 (float4) GB/s for coalesced accesses (tools/probe/wstream_probe.hip, profiles/r04_a_wstream.txt):
// a 32-row workgroup streams 640 KB of weights, 2.6 workgroups per CU = 47 us of a 89 us kernel with the load path busy.
// Forward chains get the transposed copies of the nn.Linear weights (made at the head of ur_sasrec_fwd), backward chains the weights as stored.
constexpr int RC_BK = 16;          // K-slice of the streamed weight tile

template <int D>
struct RcGeom {
  static constexpr int BM = 4096 / D;               // token rows per workgroup
  static constexpr int WC = D / 32, WR = 4 / WC;    // wave grid: WR x WC accumulator tiles of 32 x 32 = BM x D
  static constexpr int TS = D;                      // LDS row stride of the activation tiles: UNPADDED, 16-byte chunks XOR-swizzled by row
  static constexpr int SWZ = (D / 4 - 1) < 15 ? (D / 4 - 1) : 15;   // (rc_toff) -- 52 KB of LDS at D = 128: three workgroups per CU
  static constexpr int RS = D + 4;                  // row stride of the reduction scratch laid over a dead tile
  static constexpr int TPR = D / 4;                 // lanes per row in the row-wise epilogues (one float4 each)
  static constexpr int RPP = 256 / TPR;             // rows per epilogue pass; 4 passes cover the BM rows
  static constexpr int TILE = BM * TS;              // floats per activation tile
  static constexpr size_t LDS_BYTES = (size_t)(2 * TILE) * sizeof(float);
};

// offset (floats) of 16-byte chunk c4 of row r in a swizzled activation tile: rows are D floats apart (a multiple of the 64 banks), so the
// chunk index is XORed with the row -- the 16 rows a ds_read_b128 lane group touches land on 16 distinct bank quads
template <int D>
__device__ __forceinline__ int rc_toff(int r, int c4) {
  return r * D + ((c4 ^ (r & RcGeom<D>::SWZ)) << 2);
}

// A weight segment (output columns col0 .. col0+D-1, k from k0 of a K-major matrix Wt with row stride ldt) is addressed as a
// WAVE-UNIFORM pointer (rc_wptr: its first element; advanced per K-slice with scalar arithmetic) plus ONE per-lane element offset
// (rc_woff, a function of the row stride only): column = the wave's 32-column block + lane & 31, k offset 4 (lane >> 5) (the
// fragment layout of rc_gemm).  Eight dword loads per slice off one 32-bit lane offset -- with a per-lane 64-bit pointer the eight row
// addresses of a slice cost 16 registers per stream and the forward chain spilled.
// SP = true (round 6): the streamed weight matrix is a SPLIT copy -- every fp32 weight as three bf16 pieces (gemm.hip: split-bf16
// arithmetic) in the fragment layout of v_mfma_f32_32x32x16_bf16: [K / 16][piece][k group g][column n][8 bf16 = k 16 kb + 4 g + {0..3}, 16 kb + 8 + 4 g + {0..3}]
// (weight_split_kernel, gemm.hip), i.e. 16 B per (slice, piece, lane) where the fp32 stream has eight dwords per (slice, lane): three
// b128 loads per slice, 512 contiguous bytes per half-wave.  Pointers into a split copy are kept in float units (4 per 16-byte cell).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool SP> struct RcW { typedef fx4 T; static constexpr int N = 2; };          // fp32 stream: k offsets fk .. fk+3 and 8 + fk .. of ITS column
template <> struct RcW<true> { typedef bf16x8 T; static constexpr int N = 3; };        // split stream: the hi / mid / lo pieces of the same eight k
template <bool SP>
__device__ __forceinline__ long long rc_kstep(int ldt) { return (SP ? 24LL : (long long)RC_BK) * ldt; }   // floats from one K-slice to the next
template <int D, bool SP>
__device__ __forceinline__ const float* rc_wptr(const float* Wt, int ldt, int col0, int k0) {
  if (SP) return Wt + (long long)(k0 / RC_BK) * 24 * ldt + 4LL * col0;
  return Wt + (long long)k0 * ldt + col0;
}
template <int D, bool SP>
__device__ __forceinline__ unsigned rc_woff(int ldt, int tid) {   // BYTE offset, unsigned: base + zext(offset) is the scalar-base addressing mode
  const int lane = tid & 63, wc = (tid >> 6) % RcGeom<D>::WC;
  if (SP) return 16u * (unsigned)((lane >> 5) * ldt + wc * 32 + (lane & 31));
  return 4u * (unsigned)(4 * (lane >> 5) * ldt + wc * 32 + (lane & 31));
}
// (buffer loads: the wave-uniform part of an address -- segment base in the resource, k-row offset in the scalar offset -- stays in
// SGPRs by construction; with global loads the compiler turned most of the eight row addresses of a slice into per-lane 64-bit
// pointers, hoisted them out of the chunk loops and spilled)
__device__ __forceinline__ float rc_ld(__amdgpu_buffer_rsrc_t rs, unsigned boff, int sbyte) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)boff, sbyte, 0));
}
template <int D>
__device__ __forceinline__ void rc_wload(fx4 (&r)[2], const float* p, int ldt, unsigned off) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  const int lb = ldt * 4;
  r[0][0] = rc_ld(rs, off, 0); r[0][1] = rc_ld(rs, off, lb); r[0][2] = rc_ld(rs, off, 2 * lb); r[0][3] = rc_ld(rs, off, 3 * lb);
  r[1][0] = rc_ld(rs, off, 8 * lb); r[1][1] = rc_ld(rs, off, 9 * lb); r[1][2] = rc_ld(rs, off, 10 * lb); r[1][3] = rc_ld(rs, off, 11 * lb);
}
template <int D>
__device__ __forceinline__ void rc_wload(bf16x8 (&r)[3], const float* p, int ldt, unsigned off) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  const int pb = ldt * 32;   // bytes from one piece plane of a slice to the next: 2 k groups x ldt columns x 16 B
#pragma unroll
  for (int q = 0; q < 3; ++q) r[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, q * pb, 0));
}

// the three bf16 pieces of eight fp32 values (gemm.hip: tn_split_wg::split_store), element j of a piece = x[j].  The remainders are
// taken two values at a time (v_pk_add_f32): 9 VALU instructions per pair of values
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rc_split8(const float4 a0, const float4 a1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  const f32x2 x[4] = {{a0.x, a0.y}, {a0.z, a0.w}, {a1.x, a1.y}, {a1.z, a1.w}};
  u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 r = x[j] - __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, x[j]) & 0xFFFF0000u);
    const f32x2 t = r - __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, r) & 0xFFFF0000u);
    h[j] = __builtin_amdgcn_perm(__float_as_uint(x[j][1]), __float_as_uint(x[j][0]), 0x07060302u);
    m[j] = __builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u);
    l[j] = __builtin_amdgcn_perm(__float_as_uint(t[1]), __float_as_uint(t[0]), 0x07060302u);
  }
  hi = __builtin_bit_cast(bf16x8, h); mid = __builtin_bit_cast(bf16x8, m); lo = __builtin_bit_cast(bf16x8, l);
}

// acc += As[BM, D] @ Wt[seg], K = D in NK = D / 16 slices (ldw / ldwn: the k-row strides of this / the next segment's matrix).  As: an LDS activation tile (row stride TS).
// The weight-slice stream runs TWO slices ahead of the MFMAs (an L2 round trip is longer than one K-step of 8 MFMAs) and never drains
// inside a workgroup: wp / wnp are rc_wptr of this / the next segment (wnp nullable: the stream then re-reads this segment, unused).
// Ends with a barrier: every wave is done reading As.
// SP: the lane splits its A fragment (the same eight k of its row) into three bf16 pieces in registers and issues the six piece
// products of order <= 2^-16 on the bf16 pipe -- 6 x 8 passes where the fp32-input MFMA takes 8 x 16: fp32-equivalent results
// (gemm.hip; profiles/r06_a_stage_a.txt), not bit-identical to the exact stream's.
template <int D, bool SP>
__device__ __forceinline__ void rc_gemm(floatx16& acc, const float* As, const float* wp, int ldw, unsigned woff, const float* wnp, int ldwn,
                                        unsigned wnoff, typename RcW<SP>::T (&wreg)[2][RcW<SP>::N], int wr, int lane) {
  constexpr int NK = D / RC_BK;
  static_assert(NK % 2 == 0, "the two-slot ring assumes an even number of slices per segment");
  const int frow = lane & 31, fk = 4 * (lane >> 5);
  const int arow = wr * 32 + frow, ac0 = fk >> 2;   // this lane's tile row and the chunk offset of its K half
  if (!wnp) { wnp = wp; ldwn = ldw; wnoff = woff; }
  // on entry wreg[0] / wreg[1] hold this lane's fragments of slices 0 / 1 of the segment (in flight or landed); step kt consumes
  // slice kt and refills its slot with slice kt + 2 (of this segment or the next): the invariant holds again on exit
  // The swizzle XORs the chunk index with row & 15: chunk = 4 kt + q (q = ac0, 2 + ac0) has its bits above 15 untouched, so the lane's
  // address of (kt, q) is the address of (kt & 3, q) plus the CONSTANT 64 floats x (kt >> 2) -- eight lane addresses for the sixteen
  // fragments of a segment (spelled out: the compiler kept sixteen in registers for the whole kernel).
  constexpr int NA = NK < 4 ? NK : 4;
  int aoff[NA][2];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    aoff[j][0] = rc_toff<D>(arow, j * (RC_BK / 4) + 0 + ac0);
    aoff[j][1] = rc_toff<D>(arow, j * (RC_BK / 4) + 2 + ac0);
  }
#pragma unroll
  for (int kt = 0; kt < NK; ++kt) {
    typename RcW<SP>::T b[RcW<SP>::N];
#pragma unroll
    for (int q = 0; q < RcW<SP>::N; ++q) b[q] = wreg[kt & 1][q];
    if (kt + 2 < NK) rc_wload<D>(wreg[kt & 1], wp + (kt + 2) * rc_kstep<SP>(ldw), ldw, woff);
    else rc_wload<D>(wreg[kt & 1], wnp + (kt + 2 - NK) * rc_kstep<SP>(ldwn), ldwn, wnoff);
    __builtin_amdgcn_sched_barrier(0);   // (the loads stay here: the scheduler would sink them to just ahead of their use)
    const float4 a0 = *(const float4*)(As + aoff[kt % NA][0] + (kt / NA) * 64);
    const float4 a1 = *(const float4*)(As + aoff[kt % NA][1] + (kt / NA) * 64);
    if constexpr (SP) {
      bf16x8 ah, am, al;
      rc_split8(a0, a1, ah, am, al);
      // small terms first, the leading product last
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0], acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b[0][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b[0][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b[0][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b[0][3], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b[1][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b[1][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b[1][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b[1][3], acc, 0, 0, 0);
    }
  }
  __syncthreads();   // every wave is done reading As: the caller may overwrite it
}

// start of the stream: slice 0 of the first segment at kernel entry, slice 1 behind the tile staging loads
template <int D, bool SP>
__device__ __forceinline__ void rc_prime_load(typename RcW<SP>::T (&wreg)[2][RcW<SP>::N], const float* wp, int ldw, unsigned woff) {
  rc_wload<D>(wreg[0], wp, ldw, woff);
}
template <int D, bool SP>
__device__ __forceinline__ void rc_prime_next(typename RcW<SP>::T (&wreg)[2][RcW<SP>::N], const float* wp, int ldw, unsigned woff) {   // slice 1, once the tile staging loads are out
  rc_wload<D>(wreg[1], wp + rc_kstep<SP>(ldw), ldw, woff);
}

// accumulator tile -> LDS tile.  acc[r]: row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31 of the wave's 32 x 32 tile.
template <int D>
__device__ __forceinline__ void rc_acc_to_tile(const floatx16& acc, float* T, int wr, int wc, int lane) {
  const int nl = wc * 32 + (lane & 31), r4 = 4 * (lane >> 5);
#pragma unroll
  for (int r = 0; r < 16; ++r) T[rc_toff<D>(wr * 32 + (r & 3) + 8 * (r >> 2) + r4, nl >> 2) + (nl & 3)] = acc[r];
}

// A lane-dependent index as a value the optimiser cannot see through: what an epilogue derives from it (row pointers, swizzled LDS
// addresses) is then recomputed where it is used -- a few VALU operations -- instead of being hoisted out of the chunk loops and held
// in registers (or spilled) across the whole kernel: the many-row kernels sit at the 168-register edge of three workgroups per CU.
__device__ __forceinline__ int rc_fresh(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ floatx16 zero16() {
  floatx16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// The activation of a thread's 4 x float4 of an epilogue with ONE switch around the sixteen evaluations (each case a straight run of
// code).  With the switch inside the per-element call the unrolled epilogue was 3 000 instructions of which a launch executes a few
// hundred, jumping over the erff / tanhf expansions of the other cases sixteen times: 12 000 cycles per epilogue on a cold instruction
// cache (measured with s_memtime stamps; the few-row kernels run one workgroup per CU, every launch cold), ~2 000 of them arithmetic.
#define RC_ACT_CASES(FN)                                                                       \
  switch (act) {                                                                               \
    case UR_ACT_GELU: RC_ACT_LOOP_SERIAL(FN, UR_ACT_GELU); break;                              \
    case UR_ACT_RELU: RC_ACT_LOOP(FN, UR_ACT_RELU); break;                                     \
    case UR_ACT_SWISH: RC_ACT_LOOP(FN, UR_ACT_SWISH); break;                                   \
    case UR_ACT_TANH: RC_ACT_LOOP_SERIAL(FN, UR_ACT_TANH); break;                              \
    case UR_ACT_SIGMOID: RC_ACT_LOOP(FN, UR_ACT_SIGMOID); break;                               \
    default: RC_ACT_LOOP(FN, -1); break;                                                       \
  }
#define RC_ACT_LOOP(FN, A)                                                                                              \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                                       \
    v[p].x = FN(v[p].x, A); v[p].y = FN(v[p].y, A); v[p].z = FN(v[p].z, A); v[p].w = FN(v[p].w, A);                     \
  }
// (erff / tanhf are ~60 instructions each: evaluated one after the other -- a scheduling barrier between them -- or the scheduler
// interleaves all sixteen and the kernel's register allocation, which is static over every case, loses a workgroup per CU)
#define RC_ACT_LOOP_SERIAL(FN, A)                                                                                       \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                                       \
    v[p].x = FN(v[p].x, A); __builtin_amdgcn_sched_barrier(0);                                                          \
    v[p].y = FN(v[p].y, A); __builtin_amdgcn_sched_barrier(0);                                                          \
    v[p].z = FN(v[p].z, A); __builtin_amdgcn_sched_barrier(0);                                                          \
    v[p].w = FN(v[p].w, A); __builtin_amdgcn_sched_barrier(0);                                                          \
  }
__device__ __forceinline__ void rc_act_fwd16(float4 (&v)[4], int act) { RC_ACT_CASES(act_fwd) }   // v := act(v)
__device__ __forceinline__ void rc_act_bwd16(float4 (&v)[4], int act) { RC_ACT_CASES(act_bwd) }   // v := act'(v)
#undef RC_ACT_LOOP
#undef RC_ACT_LOOP_SERIAL
#undef RC_ACT_CASES

// LayerNorm forward of one row held by TPR lanes (one float4 each): x -> (xhat, y); returns rstd.  Same arithmetic as the
// EPI_BIAS_RES_LN epilogue of gemm_nt.
template <int TPR>
__device__ __forceinline__ float rc_ln_row(float4 x, float4 gm, float4 bt, float inv_n, float eps, float4& h, float4& o) {
  const float mean = group_sum<TPR>((x.x + x.y) + (x.z + x.w)) * inv_n;
  x.x -= mean; x.y -= mean; x.z -= mean; x.w -= mean;
  const float q = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  const float rstd = 1.0f / sqrtf(group_sum<TPR>(q) * inv_n + eps);
  h.x = x.x * rstd; h.y = x.y * rstd; h.z = x.z * rstd; h.w = x.w * rstd;
  o.x = h.x * gm.x + bt.x; o.y = h.y * gm.y + bt.y; o.z = h.z * gm.z + bt.z; o.w = h.w * gm.w + bt.w;
  return rstd;
}

// LayerNorm backward of one row: y = d loss / d LN output, h = xhat, r = rstd -> d loss / d LN input; dg / db accumulate this
// thread's share of d gamma / d beta.  Same arithmetic as ln_bwd_kernel.
template <int TPR>
__device__ __forceinline__ float4 rc_ln_bwd_row(float4 y, float4 h, float4 gm, float r, float inv_d, float4& dg, float4& db) {
  dg.x += y.x * h.x; dg.y += y.y * h.y; dg.z += y.z * h.z; dg.w += y.w * h.w;
  db.x += y.x; db.y += y.y; db.z += y.z; db.w += y.w;
  float4 gy;
  gy.x = y.x * gm.x; gy.y = y.y * gm.y; gy.z = y.z * gm.z; gy.w = y.w * gm.w;
  const float s1 = (gy.x + gy.y) + (gy.z + gy.w);
  const float s2 = (gy.x * h.x + gy.y * h.y) + (gy.z * h.z + gy.w * h.w);
  const float m1 = group_sum<TPR>(s1) * inv_d;
  const float m2 = group_sum<TPR>(s2) * inv_d;
  float4 o;
  o.x = r * (gy.x - m1 - h.x * m2);
  o.y = r * (gy.y - m1 - h.y * m2);
  o.z = r * (gy.z - m1 - h.z * m2);
  o.w = r * (gy.w - m1 - h.w * m2);
  return o;
}

// column sums of the per-thread (dg, db) over the workgroup's rows, in a fixed order -> part[0..D) = d gamma, part[D..2D) = d beta.
// red: an LDS region of at least RPP * 2 * RS floats that nobody reads any more.  Ends with a barrier.
template <int D>
__device__ __forceinline__ void rc_block_colsum(float4 dg, float4 db, float* red, float* part, int eg, int et, int tid) {
  using G = RcGeom<D>;
  *(float4*)(red + (eg * 2 + 0) * G::RS + et * 4) = dg;
  *(float4*)(red + (eg * 2 + 1) * G::RS + et * 4) = db;
  __syncthreads();
  for (int i = tid; i < 2 * D; i += 256) {
    const int which = i / D, col = i % D;
    float acc = 0.f;
#pragma unroll 8
    for (int g = 0; g < G::RPP; ++g) acc += red[(g * 2 + which) * G::RS + col];
    part[i] = acc;
  }
  __syncthreads();
}

// =============================================================================================== forward

template <int D, bool SP>
__global__ __launch_bounds__(256, 3) void chain_ffn_fwd_kernel(ChainFwdArgs a) {
  using G = RcGeom<D>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;                  // [BM][TS]: ctx, then a, then y
  float* Ht = smem + G::TILE;        // [BM][TS]: accumulator staging / act(h1 chunk)
  int M = a.M;
  if (a.m_dev) M = min(M, *a.m_dev);
  const int m0 = blockIdx.x * G::BM;
  if (m0 >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / G::WC, wc = wave % G::WC;
  const float inv_n = 1.0f / (float)D;
  typename RcW<SP>::T wreg[2][RcW<SP>::N];
  const unsigned woff_d = rc_woff<D, SP>(D, tid), woff_i = rc_woff<D, SP>(a.I, tid), woff_n = a.wnT ? rc_woff<D, SP>(a.ldwn, tid) : 0;
  auto WOFF = [&](int ld) { return ld == D ? woff_d : (ld == a.I ? woff_i : woff_n); };   // this lane's element offset in a K-major matrix of row stride ld
  rc_prime_load<D, SP>(wreg, rc_wptr<D, SP>(a.woT, D, 0, 0), D, WOFF(D));   // the first weight slice is in flight while the ctx tile is staged
  {
  const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ml = eg + p * G::RPP, m = m0 + ml;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) v = *(const float4*)(a.ctx + (long long)m * a.ldctx + et * 4);
    *(float4*)(At + rc_toff<D>(ml, et)) = v;
  }
  }
  rc_prime_next<D, SP>(wreg, rc_wptr<D, SP>(a.woT, D, 0, 0), D, WOFF(D));
  __syncthreads();

  // ---- 1. attention output projection + residual + LayerNorm
  {
    floatx16 acc = zero16();
    rc_gemm<D, SP>(acc, At, rc_wptr<D, SP>(a.woT, D, 0, 0), D, WOFF(D), rc_wptr<D, SP>(a.w1T, a.I, 0, 0), a.I, WOFF(a.I), wreg, wr, lane);
    rc_acc_to_tile<D>(acc, Ht, wr, wc, lane);
  }
  __syncthreads();
  {
    const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
    const float4 bs = *(const float4*)(a.bo + et * 4);
    const float4 gm = *(const float4*)(a.g1 + et * 4), bt = *(const float4*)(a.b1ln + et * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        float4 x = *(const float4*)(Ht + rc_toff<D>(ml, et));
        const float4 rs = *(const float4*)(a.res + (long long)m * a.ldres + et * 4);
        if (a.drop_out.thresh) {   // t = dropout(acc + bias) + res
          x.x += bs.x; x.y += bs.y; x.z += bs.z; x.w += bs.w;
          x = drop4(x, drop_rowkey(a.drop_out, m), (unsigned)(et * 4), a.drop_out);
          x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w;
        } else {
          x.x += bs.x + rs.x; x.y += bs.y + rs.y; x.z += bs.z + rs.z; x.w += bs.w + rs.w;
        }
        float4 h;
        const float rstd = rc_ln_row<G::TPR>(x, gm, bt, inv_n, a.eps, h, o);
        *(float4*)(a.ahat + (long long)m * D + et * 4) = h;
        *(float4*)(a.a + (long long)m * D + et * 4) = o;
        if (et == 0) a.rstd1[m] = rstd;
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
  }
  __syncthreads();

  // ---- 2. feed-forward: inner is walked in D-wide chunks; y accumulates over the chunks in registers
  floatx16 accy = zero16();
  const int nc = a.I / D;
  for (int c = 0; c < nc; ++c) {
    const float* w2p = rc_wptr<D, SP>(a.w2T, D, 0, c * D);
    {
      floatx16 acch = zero16();
      rc_gemm<D, SP>(acch, At, rc_wptr<D, SP>(a.w1T, a.I, c * D, 0), a.I, WOFF(a.I), w2p, D, WOFF(D), wreg, wr, lane);
      rc_acc_to_tile<D>(acch, Ht, wr, wc, lane);
    }
    __syncthreads();
    {
      const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
      const float4 bs = *(const float4*)(a.b1 + c * D + et * 4);
      float4 v[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int ml = eg + p * G::RPP, m = m0 + ml;
        v[p] = *(const float4*)(Ht + rc_toff<D>(ml, et));
        v[p].x += bs.x; v[p].y += bs.y; v[p].z += bs.z; v[p].w += bs.w;
        if (m < M) *(float4*)(a.h1 + (long long)m * a.I + c * D + et * 4) = v[p];
      }
      rc_act_fwd16(v, a.act);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int ml = eg + p * G::RPP, m = m0 + ml;
        if (a.u && m < M) *(float4*)(a.u + (long long)m * a.I + c * D + et * 4) = v[p];
        *(float4*)(Ht + rc_toff<D>(ml, et)) = v[p];
      }
    }
    __syncthreads();
    const float* nxp = c + 1 < nc ? rc_wptr<D, SP>(a.w1T, a.I, (c + 1) * D, 0) : (a.wnT ? rc_wptr<D, SP>(a.wnT, a.ldwn, 0, 0) : nullptr);
    rc_gemm<D, SP>(accy, Ht, w2p, D, WOFF(D), nxp, c + 1 < nc ? a.I : a.ldwn, WOFF(c + 1 < nc ? a.I : a.ldwn), wreg, wr, lane);
  }

  // ---- 3. y = LN(drop(acc + b2) + a)
  rc_acc_to_tile<D>(accy, Ht, wr, wc, lane);
  __syncthreads();
  {
    const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
    const float4 bs = *(const float4*)(a.b2 + et * 4);
    const float4 gm = *(const float4*)(a.g2 + et * 4), bt = *(const float4*)(a.b2ln + et * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        float4 x = *(const float4*)(Ht + rc_toff<D>(ml, et));
        const float4 rs = *(const float4*)(At + rc_toff<D>(ml, et));
        if (a.drop_ffn.thresh) {
          x.x += bs.x; x.y += bs.y; x.z += bs.z; x.w += bs.w;
          x = drop4(x, drop_rowkey(a.drop_ffn, m), (unsigned)(et * 4), a.drop_ffn);
          x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w;
        } else {
          x.x += bs.x + rs.x; x.y += bs.y + rs.y; x.z += bs.z + rs.z; x.w += bs.w + rs.w;
        }
        float4 h;
        const float rstd = rc_ln_row<G::TPR>(x, gm, bt, inv_n, a.eps, h, o);
        *(float4*)(a.yhat + (long long)m * D + et * 4) = h;
        *(float4*)(a.y + (long long)m * D + et * 4) = o;
        if (et == 0) a.rstd2[m] = rstd;
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
  }
  if (!a.wnT) return;
  __syncthreads();

  // ---- 4. the next layer's input projection (its K / V -- or Q, K, V -- rows), straight from the y tile
  const int nn = a.Nn / D;
  for (int c = 0; c < nn; ++c) {
    floatx16 acc = zero16();
    rc_gemm<D, SP>(acc, At, rc_wptr<D, SP>(a.wnT, a.ldwn, c * D, 0), a.ldwn, WOFF(a.ldwn), c + 1 < nn ? rc_wptr<D, SP>(a.wnT, a.ldwn, (c + 1) * D, 0) : nullptr, a.ldwn, WOFF(a.ldwn), wreg, wr, lane);
    rc_acc_to_tile<D>(acc, Ht, wr, wc, lane);
    __syncthreads();
    const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
    const float4 bs = *(const float4*)(a.bn + c * D + et * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      if (m < M) {
        float4 v = *(const float4*)(Ht + rc_toff<D>(ml, et));
        v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
        *(float4*)(a.outn + (long long)m * a.ldn + c * D + et * 4) = v;
      }
    }
    __syncthreads();
  }
}

// Handing a tile from one workgroup to another INSIDE a kernel, without a device-scope fence (which on this part writes back the whole
// L2 of the XCD): the data moves by device-scope atomic READ-MODIFY-WRITE instructions, which are performed at the point all XCDs
// agree on and -- returning their old value -- are known to be performed once the value is back.  (sc1 stores are not enough: their
// acknowledgement does not say that the write-through has landed, and the completion counter can overtake them: measured, 6 of 40
// backward passes summed a stale partial.)  8 bytes per instruction: 8 exchanges per thread and tile, 8 fetch-ors per thread and tile
// on the reading side.
// rc_dev_put4 ISSUES the two exchanges of a float4 and returns their old values; the caller hands all of them to rc_dev_wait once every
// exchange of the thread is in flight (one round trip to the coherence point -- ~1 us -- per thread and tile instead of four).
struct RcPut { unsigned long long o0, o1; };
__device__ __forceinline__ RcPut rc_dev_put4(float* p, float4 v) {
  unsigned long long* q = (unsigned long long*)p;
  const unsigned long long lo = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
  const unsigned long long hi = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
  RcPut r;
  r.o0 = __hip_atomic_exchange(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.o1 = __hip_atomic_exchange(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}
__device__ __forceinline__ void rc_dev_wait(const RcPut (&r)[4]) {   // the old values are waited for: the exchanges have been performed
  asm volatile("" ::"v"(r[0].o0), "v"(r[0].o1), "v"(r[1].o0), "v"(r[1].o1), "v"(r[2].o0), "v"(r[2].o1), "v"(r[3].o0), "v"(r[3].o1));
}
__device__ __forceinline__ float4 rc_dev_get4(const float* p) {
  unsigned long long* q = (unsigned long long*)p;
  const unsigned long long lo = __hip_atomic_fetch_or(q, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_fetch_or(q + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                     __uint_as_float((unsigned)(hi >> 32)));
}

// x[p] = sum over the nc chunk partials (chunk order: the order the one-workgroup chain accumulates in) of this thread's float4 of its
// four rows; per chunk the eight fetch-ors of the thread are issued together (one round trip per chunk instead of four)
template <int D>
__device__ __forceinline__ void rc_dev_sum4(float4 (&x)[4], const float* blk, int nc, int m0, int M, int eg, int et) {
  using G = RcGeom<D>;
#pragma unroll
  for (int p = 0; p < 4; ++p) x[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int cc = 0; cc < nc; ++cc) {
    float4 t[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP;
      t[p] = m0 + ml < M ? rc_dev_get4(blk + ((long long)cc * G::BM + ml) * D + et * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) { x[p].x += t[p].x; x[p].y += t[p].y; x[p].z += t[p].z; x[p].w += t[p].w; }
  }
}

// =============================================================================================== forward, few rows
// The B last rows of the last-row layer are 16 row blocks at B = 512: as ONE workgroup per block the chain above is 72 dependent
// K slices on 16 CUs (42 us: latency, not throughput).  Here the inner dimension is split: workgroup (block rb, chunk c) computes the
// attention-output projection + LayerNorm itself (all I/d workgroups of a block do: 1/9 of the work, cheaper than passing the tile
// around), dense_1 + activation for ITS d-wide chunk of inner, and its partial of dense_2; the partials go to memory with device-scope
// stores and the workgroup of the block that finishes LAST (a counter per block, self-resetting) sums them in chunk order -- fixed
// order: bit-reproducible -- and does the bias / residual / LayerNorm epilogue.  24 slices on the critical path instead of 72, 64
// workgroups instead of 16.  (No device-scope FENCE anywhere: on this part it writes back the XCD's whole L2.)
template <int D>
__global__ __launch_bounds__(256) void chain_ffn_fwd_split_kernel(ChainFwdArgs a) {
  using G = RcGeom<D>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;                  // [BM][TS]: ctx, then a
  float* Ht = smem + G::TILE;        // [BM][TS]: accumulator staging / act(h1 chunk)
  __shared__ int is_last;
  const int M = a.M;
  const int nc = a.I / D;
  const int rb = blockIdx.x / nc, c = blockIdx.x % nc;
  const int m0 = rb * G::BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / G::WC, wc = wave % G::WC;
  const int et = tid % G::TPR, eg = tid / G::TPR;
  const float inv_n = 1.0f / (float)D;
  fx4 wreg[2][2];
  const unsigned woff_d = rc_woff<D, false>(D, tid), woff_i = rc_woff<D, false>(a.I, tid);
  auto WOFF = [&](int ld) { return ld == D ? woff_d : woff_i; };
  rc_prime_load<D, false>(wreg, rc_wptr<D, false>(a.woT, D, 0, 0), D, WOFF(D));
  // every small operand of the epilogues is requested HERE (one workgroup per CU, parameters rewritten by the optimizer a moment ago:
  // each of these is a miss all the way to HBM, ~2 us when it is asked for where it is used, behind a barrier)
  const float4 q_bo = *(const float4*)(a.bo + et * 4), q_g1 = *(const float4*)(a.g1 + et * 4), q_b1ln = *(const float4*)(a.b1ln + et * 4);
  const float4 q_b1 = *(const float4*)(a.b1 + c * D + et * 4);
  const float4 q_b2 = *(const float4*)(a.b2 + et * 4), q_g2 = *(const float4*)(a.g2 + et * 4), q_b2ln = *(const float4*)(a.b2ln + et * 4);
  float4 q_res[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) q_res[p] = *(const float4*)(a.res + (long long)min(m0 + eg + p * G::RPP, M - 1) * a.ldres + et * 4);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ml = eg + p * G::RPP, m = m0 + ml;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) v = *(const float4*)(a.ctx + (long long)m * a.ldctx + et * 4);
    *(float4*)(At + rc_toff<D>(ml, et)) = v;
  }
  rc_prime_next<D, false>(wreg, rc_wptr<D, false>(a.woT, D, 0, 0), D, WOFF(D));
  __syncthreads();
  // ---- 1. attention output projection + residual + LayerNorm (every chunk's workgroup; chunk 0 writes a / ahat / rstd1)
  {
    floatx16 acc = zero16();
    rc_gemm<D, false>(acc, At, rc_wptr<D, false>(a.woT, D, 0, 0), D, WOFF(D), rc_wptr<D, false>(a.w1T, a.I, c * D, 0), a.I, WOFF(a.I), wreg, wr, lane);
    rc_acc_to_tile<D>(acc, Ht, wr, wc, lane);
  }
  __syncthreads();
  {
    const float4 bs = q_bo, gm = q_g1, bt = q_b1ln;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        float4 x = *(const float4*)(Ht + rc_toff<D>(ml, et));
        const float4 rs = q_res[p];
        if (a.drop_out.thresh) {   // t = dropout(acc + bias) + res
          x.x += bs.x; x.y += bs.y; x.z += bs.z; x.w += bs.w;
          x = drop4(x, drop_rowkey(a.drop_out, m), (unsigned)(et * 4), a.drop_out);
          x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w;
        } else {
          x.x += bs.x + rs.x; x.y += bs.y + rs.y; x.z += bs.z + rs.z; x.w += bs.w + rs.w;
        }
        float4 h;
        const float rstd = rc_ln_row<G::TPR>(x, gm, bt, inv_n, a.eps, h, o);
        if (c == 0) {
          *(float4*)(a.ahat + (long long)m * D + et * 4) = h;
          *(float4*)(a.a + (long long)m * D + et * 4) = o;
          if (et == 0) a.rstd1[m] = rstd;
        }
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
  }
  __syncthreads();
  // ---- 2. dense_1 + activation for chunk c, then this chunk's partial of dense_2
  const float* w2p = rc_wptr<D, false>(a.w2T, D, 0, c * D);
  {
    floatx16 acch = zero16();
    rc_gemm<D, false>(acch, At, rc_wptr<D, false>(a.w1T, a.I, c * D, 0), a.I, WOFF(a.I), w2p, D, WOFF(D), wreg, wr, lane);
    rc_acc_to_tile<D>(acch, Ht, wr, wc, lane);
  }
  __syncthreads();
  {
    const float4 bs = q_b1;
    float4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      v[p] = *(const float4*)(Ht + rc_toff<D>(ml, et));
      v[p].x += bs.x; v[p].y += bs.y; v[p].z += bs.z; v[p].w += bs.w;
      if (m < M) *(float4*)(a.h1 + (long long)m * a.I + c * D + et * 4) = v[p];
    }
    rc_act_fwd16(v, a.act);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      if (a.u && m < M) *(float4*)(a.u + (long long)m * a.I + c * D + et * 4) = v[p];
      *(float4*)(Ht + rc_toff<D>(ml, et)) = v[p];
    }
  }
  __syncthreads();
  floatx16 accy = zero16();
  rc_gemm<D, false>(accy, Ht, w2p, D, WOFF(D), nullptr, D, WOFF(D), wreg, wr, lane);
  rc_acc_to_tile<D>(accy, Ht, wr, wc, lane);
  __syncthreads();
  // ---- 3. the partial -> memory (device scope), count, and the last workgroup of the row block finishes
  float* mine = a.split_part + ((long long)(rb * nc + c) * G::BM) * D;
  {
    RcPut puts[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP;
      const float4 v = *(const float4*)(Ht + rc_toff<D>(ml, et));
      puts[p] = rc_dev_put4(mine + (long long)ml * D + et * 4, v);
    }
    rc_dev_wait(puts);                                     // (returns once all eight have been performed)
  }
  __syncthreads();                                         // every thread's part of the tile is out
  if (tid == 0) {
    const unsigned prev = ur_arrive(a.split_cnt + rb, a.arrive_mode);
    is_last = prev == (unsigned)nc - 1u;
    if (is_last) __hip_atomic_store(a.split_cnt + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
  __syncthreads();
  if (!is_last) return;
  {
    const float4 bs = q_b2, gm = q_g2, bt = q_b2ln;
    const float* blk = a.split_part + ((long long)rb * nc * G::BM) * D;
    float4 xs[4];
    rc_dev_sum4<D>(xs, blk, nc, m0, M, eg, et);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      if (m < M) {
        float4 x = xs[p];
        const float4 rs = *(const float4*)(At + rc_toff<D>(ml, et));
        if (a.drop_ffn.thresh) {
          x.x += bs.x; x.y += bs.y; x.z += bs.z; x.w += bs.w;
          x = drop4(x, drop_rowkey(a.drop_ffn, m), (unsigned)(et * 4), a.drop_ffn);
          x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w;
        } else {
          x.x += bs.x + rs.x; x.y += bs.y + rs.y; x.z += bs.z + rs.z; x.w += bs.w + rs.w;
        }
        float4 h, o;
        const float rstd = rc_ln_row<G::TPR>(x, gm, bt, inv_n, a.eps, h, o);
        *(float4*)(a.yhat + (long long)m * D + et * 4) = h;
        *(float4*)(a.y + (long long)m * D + et * 4) = o;
        if (et == 0) a.rstd2[m] = rstd;
      }
    }
  }
}

// =============================================================================================== backward of the same block

template <int D, bool SP>
__global__ __launch_bounds__(256, 3) void chain_ffn_bwd_kernel(ChainBwdArgs a) {
  using G = RcGeom<D>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;                  // [BM][TS]: g_tf, then g_ta
  float* Ht = smem + G::TILE;        // [BM][TS]: accumulator staging / g_h1 chunk / reduction scratch
  int M = a.M;
  if (a.m_dev) M = min(M, *a.m_dev);
  const int m0 = blockIdx.x * G::BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* part = a.part + (long long)blockIdx.x * 4 * D;
  if (m0 >= M) {   // surplus workgroup (compacted rows): its partial sums are zeros
    for (int i = tid; i < 4 * D; i += 256) part[i] = 0.f;
    return;
  }
  const int wr = wave / G::WC, wc = wave % G::WC;
  const float inv_d = 1.0f / (float)D;
  typename RcW<SP>::T wreg[2][RcW<SP>::N];
  const unsigned woff_d = rc_woff<D, SP>(D, tid), woff_i = rc_woff<D, SP>(a.I, tid);
  auto WOFF = [&](int ld) { return ld == D ? woff_d : woff_i; };
  rc_prime_load<D, SP>(wreg, rc_wptr<D, SP>(a.w2, a.I, 0, 0), a.I, WOFF(a.I));

  // ---- 0. feed-forward LayerNorm backward: g_tf (also the residual branch of g_a)
  {
    const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
    const float4 gm = *(const float4*)(a.g2 + et * 4);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        const float4 y = *(const float4*)(a.gy + (long long)m * D + et * 4);
        const float4 h = *(const float4*)(a.yhat + (long long)m * D + et * 4);
        o = rc_ln_bwd_row<G::TPR>(y, h, gm, a.rstd2[m], inv_d, dg, db);
        *(float4*)(a.g_tf + (long long)m * D + et * 4) = o;
        if (a.drop_ffn.thresh) {   // the feed-forward branch sees the masked gradient (the residual branch, below, the one just stored)
          o = drop4(o, drop_rowkey(a.drop_ffn, m), (unsigned)(et * 4), a.drop_ffn);
          *(float4*)(a.g_tfd + (long long)m * D + et * 4) = o;
        }
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
    rc_block_colsum<D>(dg, db, Ht, part, eg, et, tid);
  }
  rc_prime_next<D, SP>(wreg, rc_wptr<D, SP>(a.w2, a.I, 0, 0), a.I, WOFF(a.I));
  __syncthreads();

  // ---- 1. g_h1 chunk = (g_tf W2[:, chunk]) * act'(h1 chunk);   g_a += g_h1 chunk W1[chunk, :]
  floatx16 acca = zero16();
  const int nc = a.I / D;
  for (int c = 0; c < nc; ++c) {
    const float* w1p = rc_wptr<D, SP>(a.w1, D, 0, c * D);
    {
      floatx16 accu = zero16();
      rc_gemm<D, SP>(accu, At, rc_wptr<D, SP>(a.w2, a.I, c * D, 0), a.I, WOFF(a.I), w1p, D, WOFF(D), wreg, wr, lane);
      rc_acc_to_tile<D>(accu, Ht, wr, wc, lane);
    }
    __syncthreads();
    {
      const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);   // (shadow the kernel-wide ones: nothing derived from them outlives the block)
      float4 v[4];   // h1 -> act'(h1)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int m = min(m0 + eg + p * G::RPP, M - 1);
        v[p] = *(const float4*)(a.h1 + (long long)m * a.I + c * D + et * 4);
      }
      rc_act_bwd16(v, a.act);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int ml = eg + p * G::RPP, m = m0 + ml;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M) {
          g = *(const float4*)(Ht + rc_toff<D>(ml, et));
          g.x *= v[p].x; g.y *= v[p].y; g.z *= v[p].z; g.w *= v[p].w;
          *(float4*)(a.g_h1 + (long long)m * a.I + c * D + et * 4) = g;
        }
        *(float4*)(Ht + rc_toff<D>(ml, et)) = g;
      }
    }
    __syncthreads();
    const float* nxp = c + 1 < nc ? rc_wptr<D, SP>(a.w2, a.I, (c + 1) * D, 0) : rc_wptr<D, SP>(a.wo, D, 0, 0);
    rc_gemm<D, SP>(acca, Ht, w1p, D, WOFF(D), nxp, c + 1 < nc ? a.I : D, WOFF(c + 1 < nc ? a.I : D), wreg, wr, lane);
  }

  // ---- 2. g_a = acc + g_tf;  attention LayerNorm backward -> g_ta
  rc_acc_to_tile<D>(acca, Ht, wr, wc, lane);
  __syncthreads();
  {
    const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
    const float4 gm = *(const float4*)(a.g1 + et * 4);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        float4 y = *(const float4*)(Ht + rc_toff<D>(ml, et));
        // (hidden dropout: the tile holds the MASKED g_tf; the unmasked one is what this thread stored in phase 0)
        const float4 rs = a.drop_ffn.thresh ? *(const float4*)(a.g_tf + (long long)m * D + et * 4) : *(const float4*)(At + rc_toff<D>(ml, et));
        y.x += rs.x; y.y += rs.y; y.z += rs.z; y.w += rs.w;
        const float4 h = *(const float4*)(a.ahat + (long long)m * D + et * 4);
        o = rc_ln_bwd_row<G::TPR>(y, h, gm, a.rstd1[m], inv_d, dg, db);
        *(float4*)(a.g_ta + (long long)m * D + et * 4) = o;
        if (a.drop_out.thresh) {
          o = drop4(o, drop_rowkey(a.drop_out, m), (unsigned)(et * 4), a.drop_out);
          *(float4*)(a.g_tad + (long long)m * D + et * 4) = o;
        }
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
    __syncthreads();   // every read of the staged accumulators is done: Ht becomes the reduction scratch
    rc_block_colsum<D>(dg, db, Ht, part + 2 * D, eg, et, tid);
  }

  // ---- 3. g_ctx = g_ta Wo
  {
    floatx16 acc = zero16();
    rc_gemm<D, SP>(acc, At, rc_wptr<D, SP>(a.wo, D, 0, 0), D, WOFF(D), nullptr, 0, WOFF(0), wreg, wr, lane);
    rc_acc_to_tile<D>(acc, Ht, wr, wc, lane);
  }
  __syncthreads();
  const int eg = rc_fresh(tid / G::TPR), et = rc_fresh(tid % G::TPR);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ml = eg + p * G::RPP, m = m0 + ml;
    if (m < M) *(float4*)(a.g_ctx + (long long)m * D + et * 4) = *(const float4*)(Ht + rc_toff<D>(ml, et));
  }
}

// =============================================================================================== g_x = g_qkv Wqkv (+ g_ta) (+ LN0 backward)

template <int D, bool SP>
__global__ __launch_bounds__(256) void chain_proj_bwd_kernel(ChainProjBwdArgs a) {
  using G = RcGeom<D>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;
  float* Ht = smem + G::TILE;
  int M = a.M;
  if (a.m_dev) M = min(M, *a.m_dev);
  const int m0 = blockIdx.x * G::BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (m0 >= M) {
    if (a.xhat)
      for (int i = tid; i < 2 * D; i += 256) a.part[(long long)blockIdx.x * 2 * D + i] = 0.f;
    return;
  }
  const int wr = wave / G::WC, wc = wave % G::WC;
  const int et = tid % G::TPR, eg = tid / G::TPR;
  typename RcW<SP>::T wreg[2][RcW<SP>::N];
  const unsigned woff_w = rc_woff<D, SP>(a.ldw, tid);
  auto WOFF = [&](int) { return woff_w; };
  const int nkc = a.K / D;
  rc_prime_load<D, SP>(wreg, rc_wptr<D, SP>(a.w, a.ldw, 0, 0), a.ldw, WOFF(a.ldw));
  float4 ra[4];
  auto load_a = [&](int kc) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = m0 + eg + p * G::RPP;
      ra[p] = m < M ? *(const float4*)(a.g + (long long)m * a.ldg + kc * D + et * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) *(float4*)(At + rc_toff<D>(eg + p * G::RPP, et)) = ra[p];
  };
  load_a(0);
  store_a();
  rc_prime_next<D, SP>(wreg, rc_wptr<D, SP>(a.w, a.ldw, 0, 0), a.ldw, WOFF(a.ldw));
  __syncthreads();
  floatx16 acc = zero16();
  for (int kc = 0; kc < nkc; ++kc) {
    const bool more = kc + 1 < nkc;
    if (more) load_a(kc + 1);   // the next slice of g is in flight underneath this chunk's MFMAs
    rc_gemm<D, SP>(acc, At, rc_wptr<D, SP>(a.w, a.ldw, 0, kc * D), a.ldw, WOFF(a.ldw), more ? rc_wptr<D, SP>(a.w, a.ldw, 0, (kc + 1) * D) : nullptr, a.ldw, WOFF(a.ldw), wreg, wr, lane);
    if (more) {
      store_a();
      __syncthreads();
    }
  }
  rc_acc_to_tile<D>(acc, Ht, wr, wc, lane);
  __syncthreads();
  const float inv_d = 1.0f / (float)D;
  float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.xhat) gm = *(const float4*)(a.gamma + et * 4);
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ml = eg + p * G::RPP, m = m0 + ml;
    if (m < M) {
      float4 v = *(const float4*)(Ht + rc_toff<D>(ml, et));
      if (a.res) {
        const float4 rs = *(const float4*)(a.res + (long long)m * D + et * 4);
        v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
      }
      if (a.xhat) {
        if (a.drop.thresh) v = drop4(v, drop_rowkey(a.drop, m), (unsigned)(et * 4), a.drop);   // (the embedding dropout's mask)
        const float4 h = *(const float4*)(a.xhat + (long long)m * D + et * 4);
        v = rc_ln_bwd_row<G::TPR>(v, h, gm, a.rstd[m], inv_d, dg, db);
      }
      const long long orow = a.out_rows ? a.out_rows[m] : m;
      *(float4*)(a.out + orow * D + et * 4) = v;
    }
  }
  if (a.xhat) {
    __syncthreads();
    rc_block_colsum<D>(dg, db, Ht, a.part + (long long)blockIdx.x * 2 * D, eg, et, tid);
  }
}

// =============================================================================================== backward, few rows
// The mirror image of chain_ffn_fwd_split_kernel: workgroup (row block rb, chunk c) does the feed-forward LayerNorm backward itself
// (chunk 0 writes g_tf and the d gamma / d beta partials), the d act GEMM of ITS chunk of inner (g_h1 chunk out) and its partial of
// the d dense_1 GEMM; the workgroup of the block that finishes last sums the partials in chunk order, adds the residual branch, does
// the attention LayerNorm backward (g_ta, the other two partial sums) and the out-projection gradient GEMM (g_ctx).
template <int D>
__global__ __launch_bounds__(256) void chain_ffn_bwd_split_kernel(ChainBwdArgs a) {
  using G = RcGeom<D>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;                  // [BM][TS]: g_tf, then g_ta
  float* Ht = smem + G::TILE;        // [BM][TS]: accumulator staging / g_h1 chunk / reduction scratch
  __shared__ int is_last;
  const int M = a.M;
  const int nc = a.I / D;
  const int rb = blockIdx.x / nc, c = blockIdx.x % nc;
  const int m0 = rb * G::BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* part = a.part + (long long)rb * 4 * D;
  const int wr = wave / G::WC, wc = wave % G::WC;
  const int et = tid % G::TPR, eg = tid / G::TPR;
  const float inv_d = 1.0f / (float)D;
  fx4 wreg[2][2];
  const unsigned woff_d = rc_woff<D, false>(D, tid), woff_i = rc_woff<D, false>(a.I, tid);
  auto WOFF = [&](int ld) { return ld == D ? woff_d : woff_i; };
  rc_prime_load<D, false>(wreg, rc_wptr<D, false>(a.w2, a.I, c * D, 0), a.I, WOFF(a.I));
  // (what the later epilogues read from memory is requested here: see chain_ffn_fwd_split_kernel)
  const float4 q_g1 = *(const float4*)(a.g1 + et * 4);
  float4 q_h1[4], q_ahat[4];
  float q_rstd1[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int m = min(m0 + eg + p * G::RPP, M - 1);
    q_h1[p] = *(const float4*)(a.h1 + (long long)m * a.I + c * D + et * 4);
    q_ahat[p] = *(const float4*)(a.ahat + (long long)m * D + et * 4);
    q_rstd1[p] = a.rstd1[m];
  }
  // ---- 0. feed-forward LayerNorm backward (every chunk's workgroup; chunk 0 writes)
  float4 gtf_keep[4];   // hidden dropout: the UNMASKED g_tf of this thread's rows (the residual branch of phase 3; the tile holds the masked one)
  {
    const float4 gm = *(const float4*)(a.g2 + et * 4);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        const float4 y = *(const float4*)(a.gy + (long long)m * D + et * 4);
        const float4 h = *(const float4*)(a.yhat + (long long)m * D + et * 4);
        o = rc_ln_bwd_row<G::TPR>(y, h, gm, a.rstd2[m], inv_d, dg, db);
        if (c == 0) *(float4*)(a.g_tf + (long long)m * D + et * 4) = o;
      }
      gtf_keep[p] = o;
      if (a.drop_ffn.thresh && m < M) {
        o = drop4(o, drop_rowkey(a.drop_ffn, m), (unsigned)(et * 4), a.drop_ffn);
        if (c == 0) *(float4*)(a.g_tfd + (long long)m * D + et * 4) = o;
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
    if (c == 0) rc_block_colsum<D>(dg, db, Ht, part, eg, et, tid);   // (workgroup-uniform branch: the barriers inside are fine)
  }
  rc_prime_next<D, false>(wreg, rc_wptr<D, false>(a.w2, a.I, c * D, 0), a.I, WOFF(a.I));
  __syncthreads();
  // ---- 1. g_h1 chunk = (g_tf W2[:, chunk]) * act'(h1 chunk);   partial of g_a = g_h1 chunk W1[chunk, :]
  const float* w1p = rc_wptr<D, false>(a.w1, D, 0, c * D);
  {
    floatx16 accu = zero16();
    rc_gemm<D, false>(accu, At, rc_wptr<D, false>(a.w2, a.I, c * D, 0), a.I, WOFF(a.I), w1p, D, WOFF(D), wreg, wr, lane);
    rc_acc_to_tile<D>(accu, Ht, wr, wc, lane);
  }
  __syncthreads();
  {
    float4 v[4];   // h1 -> act'(h1)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      v[p] = q_h1[p];
    }
    rc_act_bwd16(v, a.act);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        g = *(const float4*)(Ht + rc_toff<D>(ml, et));
        g.x *= v[p].x; g.y *= v[p].y; g.z *= v[p].z; g.w *= v[p].w;
        *(float4*)(a.g_h1 + (long long)m * a.I + c * D + et * 4) = g;
      }
      *(float4*)(Ht + rc_toff<D>(ml, et)) = g;
    }
  }
  __syncthreads();
  floatx16 acca = zero16();
  rc_gemm<D, false>(acca, Ht, w1p, D, WOFF(D), rc_wptr<D, false>(a.wo, D, 0, 0), D, WOFF(D), wreg, wr, lane);
  rc_acc_to_tile<D>(acca, Ht, wr, wc, lane);
  __syncthreads();
  // ---- 2. the partial -> memory (device scope), count; the last workgroup of the row block goes on
  float* mine = a.split_part + ((long long)(rb * nc + c) * G::BM) * D;
  {
    RcPut puts[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP;
      const float4 v = *(const float4*)(Ht + rc_toff<D>(ml, et));
      puts[p] = rc_dev_put4(mine + (long long)ml * D + et * 4, v);
    }
    rc_dev_wait(puts);                                     // (returns once all eight have been performed)
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = ur_arrive(a.split_cnt + rb, a.arrive_mode);
    is_last = prev == (unsigned)nc - 1u;
    if (is_last) __hip_atomic_store(a.split_cnt + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!is_last) return;
  // ---- 3. g_a = sum of the partials (chunk order) + g_tf;  attention LayerNorm backward -> g_ta
  {
    const float4 gm = q_g1;
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    const float* blk = a.split_part + ((long long)rb * nc * G::BM) * D;
    float4 ys[4];
    rc_dev_sum4<D>(ys, blk, nc, m0, M, eg, et);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        float4 y = ys[p];
        const float4 rs = gtf_keep[p];
        y.x += rs.x; y.y += rs.y; y.z += rs.z; y.w += rs.w;
        const float4 h = q_ahat[p];
        o = rc_ln_bwd_row<G::TPR>(y, h, gm, q_rstd1[p], inv_d, dg, db);
        *(float4*)(a.g_ta + (long long)m * D + et * 4) = o;
        if (a.drop_out.thresh) {
          o = drop4(o, drop_rowkey(a.drop_out, m), (unsigned)(et * 4), a.drop_out);
          *(float4*)(a.g_tad + (long long)m * D + et * 4) = o;
        }
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
    __syncthreads();   // (Ht is free: it becomes the reduction scratch)
    rc_block_colsum<D>(dg, db, Ht, part + 2 * D, eg, et, tid);
  }
  // ---- 4. g_ctx = g_ta Wo
  {
    floatx16 acc = zero16();
    rc_gemm<D, false>(acc, At, rc_wptr<D, false>(a.wo, D, 0, 0), D, WOFF(D), nullptr, 0, WOFF(0), wreg, wr, lane);
    rc_acc_to_tile<D>(acc, Ht, wr, wc, lane);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ml = eg + p * G::RPP, m = m0 + ml;
    if (m < M) *(float4*)(a.g_ctx + (long long)m * D + et * 4) = *(const float4*)(Ht + rc_toff<D>(ml, et));
  }
}

// =============================================================================================== input block
// x0 = dropout(LN(E[item_seq] + P)) and the FIRST layer's Q / K / V projection in one launch: the 32 gathered rows of a workgroup go
// through the LayerNorm into the LDS tile (and out to x0 / x0hat / rstd0 for the backward) and are the A operand of the projection
// straight away -- the stand-alone lookup kernel (10-13 us) and the x0 round trip disappear.  Same arithmetic as embed_ln_fwd_kernel
// (rowops.hip) + the stand-alone projection GEMM (same K order).
template <int D, bool SP>
__global__ __launch_bounds__(256) void chain_embed_proj_kernel(ChainEmbedArgs a) {
  using G = RcGeom<D>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;                  // [BM][TS]: x0
  float* Ht = smem + G::TILE;        // [BM][TS]: accumulator staging
  int M = a.M;
  if (a.m_dev) M = min(M, *a.m_dev);
  const int m0 = blockIdx.x * G::BM;
  if (m0 >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / G::WC, wc = wave % G::WC;
  const int et = tid % G::TPR, eg = tid / G::TPR;
  const float inv_n = 1.0f / (float)D;
  typename RcW<SP>::T wreg[2][RcW<SP>::N];
  const unsigned woff_n = rc_woff<D, SP>(a.ldwn, tid);
  auto WOFF = [&](int) { return woff_n; };
  rc_prime_load<D, SP>(wreg, rc_wptr<D, SP>(a.wnT, a.ldwn, 0, 0), a.ldwn, WOFF(a.ldwn));   // the first weight slice is in flight while the rows are gathered
  {
    const float4 gm = *(const float4*)(a.g0 + et * 4), bt = *(const float4*)(a.b0ln + et * 4);
    int full[4];
    float4 x[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {          // the four rows of this lane group: ids first, then all four row loads in flight
      const int m = min(m0 + eg + p * G::RPP, M - 1);
      full[p] = a.tok ? a.tok[m] : m;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long long id = UR_ROW(a.seq[full[p]], a.n_rows);
      x[p] = *(const float4*)(a.table + id * D + et * 4);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        float4 v = x[p];
        if (a.pos) {
          const float4 ps = *(const float4*)(a.pos + (long long)(full[p] % a.L) * D + et * 4);
          v.x += ps.x; v.y += ps.y; v.z += ps.z; v.w += ps.w;
        }
        float4 h;
        const float rstd = rc_ln_row<G::TPR>(v, gm, bt, inv_n, a.eps, h, o);
        *(float4*)(a.x0hat + (long long)m * D + et * 4) = h;
        if (a.drop.thresh) o = drop4(o, mix32((unsigned)full[p] ^ a.drop.key), (unsigned)(et * 4), a.drop);
        *(float4*)(a.x0 + (long long)m * D + et * 4) = o;
        if (et == 0) a.rstd0[m] = rstd;
      }
      *(float4*)(At + rc_toff<D>(ml, et)) = o;
    }
  }
  rc_prime_next<D, SP>(wreg, rc_wptr<D, SP>(a.wnT, a.ldwn, 0, 0), a.ldwn, WOFF(a.ldwn));
  __syncthreads();
  const int nn = a.Nn / D;
  for (int c = 0; c < nn; ++c) {
    floatx16 acc = zero16();
    rc_gemm<D, SP>(acc, At, rc_wptr<D, SP>(a.wnT, a.ldwn, c * D, 0), a.ldwn, WOFF(a.ldwn), c + 1 < nn ? rc_wptr<D, SP>(a.wnT, a.ldwn, (c + 1) * D, 0) : nullptr, a.ldwn, WOFF(a.ldwn), wreg, wr, lane);
    rc_acc_to_tile<D>(acc, Ht, wr, wc, lane);
    __syncthreads();
    const float4 bs = *(const float4*)(a.bn + c * D + et * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ml = eg + p * G::RPP, m = m0 + ml;
      if (m < M) {
        float4 v = *(const float4*)(Ht + rc_toff<D>(ml, et));
        v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
        *(float4*)(a.outn + (long long)m * a.ldn + c * D + et * 4) = v;
      }
    }
    __syncthreads();
  }
}

// =============================================================================================== launchers
// Defaults (kernels.h: CHAIN_DEFAULT = forward chain only), measured in situ on C5 (profiles/r02_g_chain_masks.txt): forward chain
// 0.834 -> 0.812 ms/step (isolated: 99 us against 138 for the four launches it replaces); with the backward chains as well the step is
// SLOWER (0.858): three chain workgroups fill a CU's LDS, so the weight-gradient GEMMs of the side stream (67 KB each) cannot share
// the CUs with them any more -- in the forward pass the side stream is idle and nothing is lost.
static int g_chain_on = -1;   // bit mask: 1 forward chain, 2 backward chain, 4 projection-gradient chain, 8 / 16 forward / backward chain of the last-row layer, 32 input block, 64 the last-row layer as two launches
int chain_set_enabled(int on) {
  if (g_chain_on < 0) g_chain_on = ur_test_hook("chain_mask", CHAIN_DEFAULT) & CHAIN_ALL;   // (test hook: the GEMM-per-op paths of other widths)
  const int prev = g_chain_on;
  if (on >= 0) g_chain_on = on & CHAIN_ALL;
  return prev;
}
bool chain_shape_ok(int d, int inner) { return (d == 32 || d == 64 || d == 128) && inner % d == 0; }
bool chain_supported(int d, int inner, int which) { return (chain_set_enabled(-1) & which) && chain_shape_ok(d, inner); }
int chain_rows_per_block(int d) { return 4096 / d; }
// profiler class: the chain launches of the B last rows (a few dozen workgroups: latency-bound, not a throughput figure) are kept apart
static inline ProfClass chain_class(int M, int d) { return cdiv(M, 4096 / d) <= 64 ? PC_CHAIN_SMALL : PC_CHAIN; }

template <typename KernelT>
static void set_lds(KernelT k, size_t bytes) {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define UR_CHAIN_LAUNCH(TAG, K, LDS, ARGS, GRID)                                                                     \
  do {                                                                                                              \
    static bool attr_##TAG = (set_lds(K, LDS), true);                                                               \
    (void)attr_##TAG;                                                                                               \
    UR_LAUNCH_EV((K), dim3(GRID), dim3(256), LDS, st, ARGS);                                                        \
  } while (0)
#define UR_CHAIN_DISPATCH(D_, KERNEL, ARGS, GRID) UR_CHAIN_LAUNCH(D_, KERNEL<D_>, RcGeom<D_>::LDS_BYTES, ARGS, GRID)
// the kernels with a split-bf16 form: ARGS.wsplit says which weight copies the pointers name (kernels.h)
#define UR_CHAIN_DISPATCH_SP(D_, KERNEL, ARGS, GRID)                                                                \
  do {                                                                                                              \
    if ((ARGS).wsplit) UR_CHAIN_LAUNCH(s##D_, (KERNEL<D_, true>), RcGeom<D_>::LDS_BYTES, ARGS, GRID);               \
    else UR_CHAIN_LAUNCH(e##D_, (KERNEL<D_, false>), RcGeom<D_>::LDS_BYTES, ARGS, GRID);                            \
  } while (0)

int chain_ffn_fwd(const ChainFwdArgs& a, int d, hipStream_t st) {
  if (a.M <= 0) return UR_OK;
  if (!chain_shape_ok(d, a.I) || (a.wnT && a.Nn % d)) return fail(UR_ERR_UNSUPPORTED, "chain_ffn_fwd: d=%d inner=%d", d, a.I);
  ProfScope ps(chain_class(a.M, d), st, 2.0 * a.M * d * ((double)d + 2.0 * a.I + (a.wnT ? a.Nn : 0)), true);
  const int grid = cdiv(a.M, chain_rows_per_block(d));
  switch (d) {
    case 32: UR_CHAIN_DISPATCH_SP(32, chain_ffn_fwd_kernel, a, grid); break;
    case 64: UR_CHAIN_DISPATCH_SP(64, chain_ffn_fwd_kernel, a, grid); break;
    default: UR_CHAIN_DISPATCH_SP(128, chain_ffn_fwd_kernel, a, grid); break;
  }
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int chain_embed_proj(const ChainEmbedArgs& a, int d, hipStream_t st) {
  if (a.M <= 0) return UR_OK;
  if (!(d == 32 || d == 64 || d == 128) || a.Nn <= 0 || a.Nn % d) return fail(UR_ERR_UNSUPPORTED, "chain_embed_proj: d=%d Nn=%d", d, a.Nn);
  ProfScope ps(PC_CHAIN, st, 2.0 * a.M * d * (double)a.Nn, true);
  const int grid = cdiv(a.M, chain_rows_per_block(d));
  switch (d) {
    case 32: UR_CHAIN_DISPATCH_SP(32, chain_embed_proj_kernel, a, grid); break;
    case 64: UR_CHAIN_DISPATCH_SP(64, chain_embed_proj_kernel, a, grid); break;
    default: UR_CHAIN_DISPATCH_SP(128, chain_embed_proj_kernel, a, grid); break;
  }
  UR_LAUNCH_CHECK();
  return UR_OK;
}

long long chain_split_part_floats(int M, int d, int inner) {
  const int bm = chain_rows_per_block(d);
  return (long long)cdiv(M, bm) * (inner / d) * bm * d;
}
// counters of the split kernels: one per row block, zeroed once per device, reset by the kernel that used them
static unsigned* chain_split_counters() {
  // per device AND per context of the calling thread (common.h: g_ctx_id): launches of two rank threads of the loopback transport run
  // concurrently on their own streams and must not share hand-off counters
  static unsigned* z[64][UR_MAX_CTX] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  unsigned*& slot = z[dev][(g_ctx_id >= 0 && g_ctx_id < UR_MAX_CTX) ? g_ctx_id : 0];
  if (!slot) {
    unsigned* p = nullptr;
    if (hipMalloc((void**)&p, 2 * CHAIN_SPLIT_MAX_BLOCKS * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 2 * CHAIN_SPLIT_MAX_BLOCKS * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    slot = p;
  }
  return slot;
}

int chain_ffn_fwd_split(const ChainFwdArgs& a0, int d, hipStream_t st) {
  if (a0.M <= 0) return UR_OK;
  ChainFwdArgs a = a0;
  const int nblk = cdiv(a.M, chain_rows_per_block(d));
  if (!chain_shape_ok(d, a.I) || a.wnT || a.m_dev || nblk > CHAIN_SPLIT_MAX_BLOCKS || !a.split_part)
    return fail(UR_ERR_UNSUPPORTED, "chain_ffn_fwd_split: d=%d inner=%d M=%d", d, a.I, a.M);
  a.split_cnt = chain_split_counters();
  a.arrive_mode = ur_arrive_mode();
  if (!a.split_cnt) return fail(UR_ERR_HIP, "chain_ffn_fwd_split: no device memory for the completion counters");
  ProfScope ps(chain_class(a.M, d), st, 2.0 * a.M * d * ((double)d + 2.0 * a.I), true);
  const int grid = nblk * (a.I / d);
  switch (d) {
    case 32: UR_CHAIN_DISPATCH(32, chain_ffn_fwd_split_kernel, a, grid); break;
    case 64: UR_CHAIN_DISPATCH(64, chain_ffn_fwd_split_kernel, a, grid); break;
    default: UR_CHAIN_DISPATCH(128, chain_ffn_fwd_split_kernel, a, grid); break;
  }
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int chain_ffn_bwd_split(const ChainBwdArgs& a0, int d, hipStream_t st) {
  if (a0.M <= 0) return UR_OK;
  ChainBwdArgs a = a0;
  const int nblk = cdiv(a.M, chain_rows_per_block(d));
  if (!chain_shape_ok(d, a.I) || a.m_dev || nblk > CHAIN_SPLIT_MAX_BLOCKS || !a.split_part)
    return fail(UR_ERR_UNSUPPORTED, "chain_ffn_bwd_split: d=%d inner=%d M=%d", d, a.I, a.M);
  unsigned* cnt = chain_split_counters();
  if (!cnt) return fail(UR_ERR_HIP, "chain_ffn_bwd_split: no device memory for the completion counters");
  a.split_cnt = cnt + CHAIN_SPLIT_MAX_BLOCKS;   // (the forward kernel's counters are the first half)
  a.arrive_mode = ur_arrive_mode();
  ProfScope ps(chain_class(a.M, d), st, 2.0 * a.M * d * ((double)d + 2.0 * a.I), true);
  const int grid = nblk * (a.I / d);
  switch (d) {
    case 32: UR_CHAIN_DISPATCH(32, chain_ffn_bwd_split_kernel, a, grid); break;
    case 64: UR_CHAIN_DISPATCH(64, chain_ffn_bwd_split_kernel, a, grid); break;
    default: UR_CHAIN_DISPATCH(128, chain_ffn_bwd_split_kernel, a, grid); break;
  }
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int chain_ffn_bwd(const ChainBwdArgs& a, int d, hipStream_t st) {
  if (a.M <= 0) return UR_OK;
  if (!chain_shape_ok(d, a.I)) return fail(UR_ERR_UNSUPPORTED, "chain_ffn_bwd: d=%d inner=%d", d, a.I);
  ProfScope ps(chain_class(a.M, d), st, 2.0 * a.M * d * ((double)d + 2.0 * a.I), true);
  const int grid = cdiv(a.M, chain_rows_per_block(d));
  switch (d) {
    case 32: UR_CHAIN_DISPATCH_SP(32, chain_ffn_bwd_kernel, a, grid); break;
    case 64: UR_CHAIN_DISPATCH_SP(64, chain_ffn_bwd_kernel, a, grid); break;
    default: UR_CHAIN_DISPATCH_SP(128, chain_ffn_bwd_kernel, a, grid); break;
  }
  UR_LAUNCH_CHECK();
  return UR_OK;
}

int chain_proj_bwd(const ChainProjBwdArgs& a, int d, hipStream_t st) {
  if (a.M <= 0) return UR_OK;
  if (!(d == 32 || d == 64 || d == 128) || a.K % d) return fail(UR_ERR_UNSUPPORTED, "chain_proj_bwd: d=%d K=%d", d, a.K);
  ProfScope ps(PC_CHAIN, st, 2.0 * a.M * d * (double)a.K, true);
  const int grid = cdiv(a.M, chain_rows_per_block(d));
  switch (d) {
    case 32: UR_CHAIN_DISPATCH_SP(32, chain_proj_bwd_kernel, a, grid); break;
    case 64: UR_CHAIN_DISPATCH_SP(64, chain_proj_bwd_kernel, a, grid); break;
    default: UR_CHAIN_DISPATCH_SP(128, chain_proj_bwd_kernel, a, grid); break;
  }
  UR_LAUNCH_CHECK();
  return UR_OK;
}

}  // namespace ur

extern "C" int ur_sasrec_set_chain(int mask) { return ur::chain_set_enabled(mask < 0 ? 0 : (mask & ur::CHAIN_ALL)); }
