// GRU4Rec-style user encoder (unirec/model/sequential/gru.py:13-35; the arithmetic is torch.nn.GRU's, 1 layer,
// batch_first, h0 = 0, gate order r,z,n):
//   x_t = E[item_seq[:,t]]                      all L steps run, including the left padding (zero rows)
//   gi  = x W_ih^T + b_ih                        ONE MFMA GEMM over all B*L tokens        (gemm_nt)
//   gh  = h_{t-1} W_hh^T + b_hh                  per step: [B,H] x [H,3H]                  (gemm_nt)
//   r = s(gi_r + gh_r); z = s(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h_t = (1-z) n + z h_{t-1}   (cell kernel)
//   user_emb = h_{L-1} W_d^T + b_d               only the last step is projected (gru.py:31-33 projects all L and slices)
// Activations are kept TIME-MAJOR ([L][B][.]) so that every step's matrices are contiguous and the weight gradients
// are two big token-dimension GEMMs after the backward sweep.  Backward = BPTT with saved gates.
#include <stdlib.h>

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
  float *dh, *dh_carry, *dh_parts, *dgi, *dgh, *dx_tm, *w_ihT, *w_hhT, *w_dT, *tn_ws, *tn_ws2, *s_whh;
  long long total_floats;
};
// pieces of the K dimension of a per-step GEMM (few rows, K = H forward / 3H backward): ~192 columns each, so that B / 32 x N / 128 x pieces
// workgroups are several per CU (H = 768: 288 x 4 forward, 96 x 12 backward) instead of one walking the whole K loop alone
static int gru_ksplit(long long K) {
  long long s = std::max(1LL, std::min(16LL, K / 192));
  // gemm_nt gives every piece ceil(K / s) rounded up to its 32-wide K step: with K > 3072 the last of 16 such pieces would start
  // behind K (an empty piece, whose partial product nobody writes) -- as many pieces as are non-empty
  const long long ks = ((K + s - 1) / s + 31) / 32 * 32;
  return (int)((K + ks - 1) / ks);
}

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
  w.x = take(M * d); w.gi = take(M * 3 * H); w.gh = take(gru_ksplit(H) * B * 3 * H); w.h_all = take((L + 1) * B * H);
  w.r = take(M * H); w.z = take(M * H); w.n = take(M * H); w.hn = take(M * H);
  w.dh = take(B * H); w.dh_carry = take(B * H); w.dh_parts = take(gru_ksplit(3 * H) * B * H); w.dgi = take(M * 3 * H); w.dgh = take(M * 3 * H); w.dx_tm = take(M * d);
  w.w_ihT = take(3 * H * d); w.w_hhT = take(3 * H * H); w.w_dT = take(d * H);
  w.s_whh = (H % 128 == 0) ? take(3 * H * H * 3 / 2) : nullptr;   // split-bf16 copy of W_hh for the step kernels (made by the pass that streams it)
  long long tn = gemm_tn_ws_floats((int)M, (int)(3 * H), (int)H);
  if (gemm_tn_ws_floats((int)M, (int)(3 * H), (int)d) > tn) tn = gemm_tn_ws_floats((int)M, (int)(3 * H), (int)d);
  if (gemm_tn_ws_floats((int)B, (int)d, (int)H) > tn) tn = gemm_tn_ws_floats((int)B, (int)d, (int)H);
  w.tn_ws = take(tn);
  w.tn_ws2 = take(gemm_tn_ws_floats((int)M, (int)(3 * H), (int)d));
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

// gh: S partial products [S][B][3H] of h_{t-1} W_hh^T (S = 1: the product itself), summed here in piece order, + b_hh
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh, int S,
                                                           const float* __restrict__ b_hh,
                                                           const float* __restrict__ h_prev, int B, int H, float* __restrict__ h_out,
                                                           float* __restrict__ r_s, float* __restrict__ z_s, float* __restrict__ n_s,
                                                           float* __restrict__ hn_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i % H;
  const float* gib = gi + (long long)b * 3 * H;
  const float* ghb = gh + (long long)b * 3 * H;
  const long long ps = (long long)B * 3 * H;
  float ghr = ghb[j], ghz = ghb[H + j], ghn = ghb[2 * H + j];
  for (int s = 1; s < S; ++s) { ghr += ghb[s * ps + j]; ghz += ghb[s * ps + H + j]; ghn += ghb[s * ps + 2 * H + j]; }
  ghr += b_hh[j]; ghz += b_hh[H + j]; ghn += b_hh[2 * H + j];
  const float r = 1.0f / (1.0f + expf(-(gib[j] + ghr)));
  const float z = 1.0f / (1.0f + expf(-(gib[H + j] + ghz)));
  const float hn = ghn;
  const float n = tanhf(gib[2 * H + j] + r * hn);
  h_out[i] = (1.0f - z) * n + z * h_prev[i];
  r_s[i] = r; z_s[i] = z; n_s[i] = n; hn_s[i] = hn;
}

// dh_t: dh (S = 0: the gradient that enters the sweep) or the S partial products [S][B][H] of dgh_{t+1} W_hh summed in piece order + the
// carry dh_{t+1} z that the previous call left in dh_carry (read, then overwritten with this step's, by the same thread)
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ dh, int S, const float* __restrict__ r_s,
                                                           const float* __restrict__ z_s, const float* __restrict__ n_s,
                                                           const float* __restrict__ hn_s, const float* __restrict__ h_prev, int B,
                                                           int H, float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dh_carry) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i % H;
  float g = dh[i];
  if (S > 0) {
    for (int s = 1; s < S; ++s) g += dh[(long long)s * B * H + i];
    g += dh_carry[i];
  }
  const float r = r_s[i], z = z_s[i], n = n_s[i], hn = hn_s[i];
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


// ------------------------------------------------------------------------------------------------------------
// Persistent recurrence (H <= 128, H % 16 == 0): ONE launch runs all L steps.  The per-step product
// h_{t-1} W_hh^T is tiny ([B,H] x [H,3H]) and strictly sequential in t, so the per-step formulation above is bound by
// 2 L launches (C4: 100 launches of ~10 us per pass).  Here a workgroup owns 16 sequences for the whole sweep:
//   * wave q owns hidden units [16q, 16q+16) of ALL THREE gates, with its slice of W_hh resident in VGPRs for the
//     whole kernel as v_mfma_f32_16x16x4_f32 B-fragments (3 * H/4 = 96 registers at H = 128) -- the weights are
//     read from HBM once per workgroup, not once per step;
//   * h_{t-1} [16, H] lives in LDS (double-buffered) and is the A operand (16-byte LDS reads; the k index of an MFMA
//     step is a free permutation, so lane-quarter kq feeds k = 16 s + 4 kq + i);
//   * the accumulators of the three gates land lane-locally (row = 4 kq + r, column = hidden unit), so the gate
//     non-linearities and the state update need no exchange; gi_t is prefetched from HBM under the MFMAs.
// One __syncthreads per step.  The backward kernel mirrors it: dh lives in registers (the accumulator layout of
// dh_{t-1} = dgh_t W_hh + dh_t z is exactly the layout the cell backward of step t-1 reads), dgh_t [16, 3H] goes through
// LDS as the A operand, W_hh columns stay in registers.
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int GRU_SEQ_ROWS = 16;

template <int H>
__global__ __launch_bounds__(H * 4) void gru_seq_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_hh, int B, int L, float* __restrict__ h_all,
                                                           float* __restrict__ r_s, float* __restrict__ z_s, float* __restrict__ n_s,
                                                           float* __restrict__ hn_s) {
  constexpr int KS = H / 16;          // 16-wide k blocks
  constexpr int LD = H + 4;           // LDS row stride
  __shared__ __attribute__((aligned(16))) float hs[2][GRU_SEQ_ROWS][LD];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c16 = lane & 15, kq = lane >> 4;
  const int j = 16 * q + c16;          // hidden unit of this lane's accumulator column
  const int b0 = blockIdx.x * GRU_SEQ_ROWS;
  // resident weights: wf[g][s][i] = W_hh[g*H + j][16 s + 4 kq + i]
  float4 wf[3][KS];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[g][s] = *(const float4*)(w_hh + (long long)(g * H + j) * H + 16 * s + 4 * kq);
  const float bh_r = b_hh[j], bh_z = b_hh[H + j], bh_n = b_hh[2 * H + j];
  for (int i = threadIdx.x; i < GRU_SEQ_ROWS * LD; i += blockDim.x) (&hs[0][0][0])[i] = 0.f;   // h_0 = 0
  __syncthreads();
  for (int t = 0; t < L; ++t) {
    const int cur = t & 1;
    // this lane's gi values (rows 4 kq + r): issued before the MFMAs, consumed after
    float gir[4], giz[4], gin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = min(b0 + 4 * kq + r, B - 1);
      const float* gp = gi + ((long long)t * B + b) * 3 * H;
      gir[r] = gp[j]; giz[r] = gp[H + j]; gin[r] = gp[2 * H + j];
    }
    floatx4 ar = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f}, an = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 a4 = *(const float4*)&hs[cur][c16][16 * s + 4 * kq];
      ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wf[0][s].x, ar, 0, 0, 0);
      az = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wf[1][s].x, az, 0, 0, 0);
      an = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wf[2][s].x, an, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wf[0][s].y, ar, 0, 0, 0);
      az = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wf[1][s].y, az, 0, 0, 0);
      an = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wf[2][s].y, an, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wf[0][s].z, ar, 0, 0, 0);
      az = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wf[1][s].z, az, 0, 0, 0);
      an = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wf[2][s].z, an, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wf[0][s].w, ar, 0, 0, 0);
      az = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wf[1][s].w, az, 0, 0, 0);
      an = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wf[2][s].w, an, 0, 0, 0);
    }
    // accumulator register r: row 4 kq + r, column j
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * kq + r, b = b0 + m;
      const float rr = 1.0f / (1.0f + expf(-(gir[r] + (ar[r] + bh_r))));
      const float zz = 1.0f / (1.0f + expf(-(giz[r] + (az[r] + bh_z))));
      const float hn = an[r] + bh_n;
      const float nn = tanhf(gin[r] + rr * hn);
      const float hnew = (1.0f - zz) * nn + zz * hs[cur][m][j];
      hs[cur ^ 1][m][j] = hnew;
      if (b < B) {
        const long long o = ((long long)t * B + b) * H + j;
        h_all[o + (long long)B * H] = hnew;
        r_s[o] = rr; z_s[o] = zz; n_s[o] = nn; hn_s[o] = hn;
      }
    }
    __syncthreads();
  }
}

// backward sweep: dh_in [B,H] = gradient w.r.t. h_L; writes dgi, dgh [L][B][3H] (time-major)
template <int H>
__global__ __launch_bounds__(H * 4) void gru_seq_bwd_kernel(const float* __restrict__ dh_in, const float* __restrict__ w_hh,
                                                           const float* __restrict__ r_s, const float* __restrict__ z_s,
                                                           const float* __restrict__ n_s, const float* __restrict__ hn_s,
                                                           const float* __restrict__ h_all, int B, int L, float* __restrict__ dgi,
                                                           float* __restrict__ dgh) {
  constexpr int KS = 3 * H / 16;      // 16-wide blocks of the contraction index c in [0, 3H)
  constexpr int LD = 3 * H + 4;
  __shared__ __attribute__((aligned(16))) float dg[GRU_SEQ_ROWS][LD];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c16 = lane & 15, kq = lane >> 4;
  const int j = 16 * q + c16;          // column of dh this lane owns (= hidden unit of its cell backward)
  const int b0 = blockIdx.x * GRU_SEQ_ROWS;
  // resident weights: wf[s][i] = W_hh[16 s + 4 kq + i][j]   (B operand of dh_{t-1}[m, j] = sum_c dgh[m, c] W_hh[c, j])
  float wf[KS][4];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[s][i] = w_hh[(long long)(16 * s + 4 * kq + i) * H + j];
  float dh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dh[r] = dh_in[(long long)min(b0 + 4 * kq + r, B - 1) * H + j];
  for (int t = L - 1; t >= 0; --t) {
    float carry[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * kq + r, b = min(b0 + m, B - 1);
      const long long o = ((long long)t * B + b) * H + j;
      const float g = dh[r], rr = r_s[o], zz = z_s[o], nn = n_s[o], hn = hn_s[o], hp = h_all[o];
      const float dn = g * (1.0f - zz);
      const float dz = g * (hp - nn);
      const float dan = dn * (1.0f - nn * nn);
      const float daz = dz * zz * (1.0f - zz);
      const float dar = dan * hn * rr * (1.0f - rr);
      const float dhn = dan * rr;
      dg[m][j] = dar; dg[m][H + j] = daz; dg[m][2 * H + j] = dhn;
      carry[r] = g * zz;
      if (b0 + m < B) {
        float* gio = dgi + ((long long)t * B + b) * 3 * H;
        float* gho = dgh + ((long long)t * B + b) * 3 * H;
        gio[j] = dar; gio[H + j] = daz; gio[2 * H + j] = dan;
        gho[j] = dar; gho[H + j] = daz; gho[2 * H + j] = dhn;
      }
    }
    __syncthreads();
    floatx4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};   // two chains: hide the 40-cycle dependent latency
#pragma unroll
    for (int s = 0; s < KS; s += 2) {
      const float4 x0 = *(const float4*)&dg[c16][16 * s + 4 * kq];
      const float4 x1 = *(const float4*)&dg[c16][16 * (s + 1) + 4 * kq];
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, wf[s][0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, wf[s + 1][0], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, wf[s][1], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, wf[s + 1][1], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, wf[s][2], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, wf[s + 1][2], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, wf[s][3], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, wf[s + 1][3], a1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = (a0[r] + a1[r]) + carry[r];
    __syncthreads();   // dg is rewritten by the next step
  }
}


// ------------------------------------------------------------------------------------------------------------
// Round 3: FOUR sequences per workgroup on v_mfma_f32_4x4x1_16b_f32 (H = 64, 128).  The 16-row kernels above put B / 16 = 32
// workgroups on a 256-CU chip at B = 512, and a step's 16 x 3H x H product is 2.6 us of one CU's fp32 MFMA rate (H = 128): the
// sweep is bound by how few CUs it runs on (gru class 0.064 of the roof in round 2).  The 4x4x1 form multiplies sixteen independent
// 4 x 4 blocks per instruction at the same 64 flop / clk / SIMD, and its A operand is only 4 rows tall: with the four rows = four
// sequences and the 64 B-operand lanes = 64 different gate columns (lane l of block b = l / 4: A from lane 4 b + i = row i,
// B / D column = l, D register r = row r; tools/probe/mfma4x4_probe.hip), a workgroup needs 4 sequences, not 16 -- B / 4 = 128
// workgroups, each step 0.64 us of MFMA.
//   forward : 3H columns = 3H / 64 column groups x K splits = 24 units, three per wave (8 waves); a unit's W_hh slice (64 columns x
//             H / splits k) is resident in VGPRs for the whole sweep; the per-unit partial pre-activations go through LDS and thread
//             (row i, hidden unit j) sums them in split order, applies the gate math and writes h_t (LDS for the next step, HBM for
//             the backward) -- gi_t is prefetched under the MFMAs.  Two barriers per step.
//   backward: dh_{t-1} = dgh_t W_hh + dh_t z: H columns x 3H deep = 8 units, one per wave (three interleaved accumulator chains);
//             thread (i, j) keeps dh[i][j] in a register across the sweep.
// Round 6: ROWS sequences per workgroup, 4 (the default) or 2 (test hook gru_rows=2).  A step is a serial chain MFMA -> partials through LDS
// -> gate math -> h through LDS, and its wall time is ONE workgroup's latency (profiles/r05_i_gru_h128_pmc.txt: per step 1 536 MFMA cycles
// + ~1 590 VALU issue cycles per SIMD + waits).  Round 5 proposed 2 rows: the gate math of a step is then one wave per SIMD instead of two
// at the same MFMA count (A rows 2 and 3 of every 4 x 4 block zero), and B = 512 fills all 256 CUs.  Built and measured: SLOWER (C4 encoder
// 0.441 -> 0.492 ms, H = 64 0.320 -> 0.340): the VALU leg is latency, not issue, and twice the workgroups re-read W_hh and gi.
constexpr int GRU4_ROWS = 4;
static int gru_seq4_rows(int B) {
  static const int forced = ur_test_hook("gru_rows", 0);
  if (forced == 2 || forced == 4) return forced;
  (void)B;
  return 4;   // measured (profiles/r06_e_gru_rows_counterexample.txt): 2 rows per workgroup are SLOWER at B = 512 (gru class 0.189 -> 0.238 ms)
}

template <int H, int ROWS>
__global__ __launch_bounds__(512) void gru_seq4_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_hh, int B, int L, float* __restrict__ h_all,
                                                           float* __restrict__ r_s, float* __restrict__ z_s, float* __restrict__ n_s,
                                                           float* __restrict__ hn_s) {
  constexpr int C = 3 * H, NCG = C / 64, KSPL = 24 / NCG, KU = H / KSPL;   // column groups, K splits, k per unit
  constexpr int LDH = H + 4;
  __shared__ __attribute__((aligned(16))) float hs[2][ROWS][LDH];
  __shared__ __attribute__((aligned(16))) float part[KSPL][ROWS][C];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b0 = blockIdx.x * ROWS;
  int uc[3], uk[3], us[3];   // this wave's units: gate column of this lane, first k, split index
  float wf[3][KU];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int u = w + 8 * q;
    us[q] = u / NCG;
    uc[q] = 64 * (u % NCG) + lane;
    uk[q] = us[q] * KU;
#pragma unroll
    for (int k = 0; k < KU; k += 4) {
      const float4 v = *(const float4*)(w_hh + (long long)uc[q] * H + uk[q] + k);
      wf[q][k] = v.x; wf[q][k + 1] = v.y; wf[q][k + 2] = v.z; wf[q][k + 3] = v.w;
    }
  }
  const bool gact = tid < ROWS * H;       // gate thread (row gi_i, hidden unit gj)
  const int gi_i = tid / H, gj = tid % H;
  const int gb = min(b0 + gi_i, B - 1);
  const float bh_r = b_hh[gj], bh_z = b_hh[H + gj], bh_n = b_hh[2 * H + gj];
  for (int i = tid; i < ROWS * LDH; i += 512) (&hs[0][0][0])[i] = 0.f;   // h_0 = 0
  __syncthreads();
  const int ai = lane & 3;   // row of the A operand this lane supplies
  for (int t = 0; t < L; ++t) {
    const int cur = t & 1;
    float g_r = 0.f, g_z = 0.f, g_n = 0.f;
    if (gact) {   // issued before the MFMAs, consumed after
      const float* gp = gi + ((long long)t * B + gb) * C;
      g_r = gp[gj]; g_z = gp[H + gj]; g_n = gp[2 * H + gj];
    }
    floatx4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KU; k += 4) {
      float4 a4[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a4[q] = ai < ROWS ? *(const float4*)&hs[cur][ai][uk[q] + k] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].x, wf[q][k], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].y, wf[q][k + 1], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].z, wf[q][k + 2], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].w, wf[q][k + 3], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) part[us[q]][r][uc[q]] = acc[q][r];
    __syncthreads();
    if (gact) {
      float pr = 0.f, pz = 0.f, pn = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < KSPL; ++s_) { pr += part[s_][gi_i][gj]; pz += part[s_][gi_i][H + gj]; pn += part[s_][gi_i][2 * H + gj]; }
      const float rr = 1.0f / (1.0f + expf(-(g_r + (pr + bh_r))));
      const float zz = 1.0f / (1.0f + expf(-(g_z + (pz + bh_z))));
      const float hn = pn + bh_n;
      const float nn = tanhf(g_n + rr * hn);
      const float hnew = (1.0f - zz) * nn + zz * hs[cur][gi_i][gj];
      hs[cur ^ 1][gi_i][gj] = hnew;
      if (b0 + gi_i < B) {
        const long long o = ((long long)t * B + gb) * H + gj;
        h_all[o + (long long)B * H] = hnew;
        r_s[o] = rr; z_s[o] = zz; n_s[o] = nn; hn_s[o] = hn;
      }
    }
    __syncthreads();
  }
}

template <int H, int ROWS>
__global__ __launch_bounds__(512) void gru_seq4_bwd_kernel(const float* __restrict__ dh_in, const float* __restrict__ w_hh,
                                                           const float* __restrict__ r_s, const float* __restrict__ z_s,
                                                           const float* __restrict__ n_s, const float* __restrict__ hn_s,
                                                           const float* __restrict__ h_all, int B, int L, float* __restrict__ dgi,
                                                           float* __restrict__ dgh) {
  constexpr int C = 3 * H, NCG = H / 64, KSPL = 8 / NCG, KU = C / KSPL;   // H = 128: 2 column groups x 4 splits of 96; H = 64: 1 x 8 of 24
  constexpr int LDG = C + 4;
  __shared__ __attribute__((aligned(16))) float dg[ROWS][LDG];
  __shared__ __attribute__((aligned(16))) float part[KSPL][ROWS][H];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b0 = blockIdx.x * ROWS;
  const int ks = w / NCG, col = 64 * (w % NCG) + lane, k0 = ks * KU;
  float wf[KU];   // W_hh[k0 + k][col]: the B operand of dh_{t-1}[i, col] = sum_c dgh[i, c] W_hh[c, col]
#pragma unroll
  for (int k = 0; k < KU; ++k) wf[k] = w_hh[(long long)(k0 + k) * H + col];
  const bool gact = tid < ROWS * H;
  const int gi_i = tid / H, gj = tid % H;
  const int gb = min(b0 + gi_i, B - 1);
  float dh = gact ? dh_in[(long long)gb * H + gj] : 0.f;
  const int ai = lane & 3;
  for (int t = L - 1; t >= 0; --t) {
    float carry = 0.f;
    if (gact) {
      const long long o = ((long long)t * B + gb) * H + gj;
      const float g = dh, rr = r_s[o], zz = z_s[o], nn = n_s[o], hn = hn_s[o], hp = h_all[o];
      const float dn = g * (1.0f - zz);
      const float dz = g * (hp - nn);
      const float dan = dn * (1.0f - nn * nn);
      const float daz = dz * zz * (1.0f - zz);
      const float dar = dan * hn * rr * (1.0f - rr);
      const float dhn = dan * rr;
      dg[gi_i][gj] = dar; dg[gi_i][H + gj] = daz; dg[gi_i][2 * H + gj] = dhn;
      carry = g * zz;
      if (b0 + gi_i < B) {
        float* gio = dgi + ((long long)t * B + gb) * C;
        float* gho = dgh + ((long long)t * B + gb) * C;
        gio[gj] = dar; gio[H + gj] = daz; gio[2 * H + gj] = dan;
        gho[gj] = dar; gho[H + gj] = daz; gho[2 * H + gj] = dhn;
      }
    }
    __syncthreads();
    floatx4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KU; k += 12) {   // three interleaved accumulator chains (KU is a multiple of 12 for H = 64, 128)
      float4 a4[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a4[q] = ai < ROWS ? *(const float4*)&dg[ai][k0 + k + 4 * q] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].x, wf[k + 4 * q], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].y, wf[k + 4 * q + 1], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].z, wf[k + 4 * q + 2], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q].w, wf[k + 4 * q + 3], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) part[ks][r][col] = (acc[0][r] + acc[1][r]) + acc[2][r];
    __syncthreads();
    if (gact) {
      float sum = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < KSPL; ++s_) sum += part[s_][gi_i][gj];
      dh = sum + carry;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Round 4: the recurrence of a WIDE hidden state (H = 768: unirec/config/model/GRU.yaml:4, the reference's default; any H % 64 == 0
// outside the persistent kernels' range) as ONE launch per time step and direction.  W_hh is 7 MB at H = 768: it does not live in one
// CU, and a step's product is 1.8 GFLOP -- 11.5 us of the chip's fp32 MFMA rate -- so a launch per step is not the bound; what was:
// the per-step path of rounds 1-3 ran gemm_nt with its K dimension split into 4 (forward) / 12 (backward) pieces, wrote the partial
// products to HBM (19 MB a step) and summed them in a separate cell kernel: 29 + 10 us a step, 309 launches a pass, 4.85 ms.  Here:
//   * a workgroup owns 32 rows x NU hidden units (48 at H = 768: 16 x 16 = 256 workgroups) and ALL THREE gates of those units (forward: 9
//     column tiles of 16 = r, z, n of the same units), so the gate arithmetic is the epilogue of the product and h_t, r, z, n, hn are
//     the only things written;
//   * its four waves split K (no two waves read the same operand bytes: fragments go straight from L2 into registers, a ring of NS
//     16-wide k blocks deep), v_mfma_f32_16x16x4_f32, 2 x NT accumulator tiles per wave; the four partial tiles meet in LDS and are summed
//     in wave order (fixed order: bit-reproducible);
//   * blockIdx -> (unit block, row block) keeps the workgroups of one XCD on 1 / 8 of the unit blocks: 0.9 MB of W_hh + the 1.5 MB of
//     h_{t-1} per XCD stay in its 4 MB L2 (PMC: 88 % L2 hits, 13 MB from HBM a step = the operands once);
//   * backward: dh_{t-1} = dgh_t W_hh + dh_t z with the carry added in the epilogue (the cell kernel in front of it no longer sums pieces).
// Measured (B = 512, L = 50, one MI355X, tools/probe/gru_host.py): forward sweep 1.83 -> 1.43 ms, backward 2.89 -> 2.40 ms; a step's
// launch 24 / 20 us = 0.47 / 0.57 of the MFMA rate (PMC 0.34-0.38 busy incl. the ramps), the rest ~7 us of fixed cost per launch.
constexpr int GS_ROWS = 32;
struct __attribute__((packed, aligned(4))) GsF3 { float a, b, c; };
struct __attribute__((packed, aligned(4))) GsF2 { float a, b; };
struct GruStepArgs {
  const float* A; int lda;            // [B][K]: h_{t-1} (forward) / dgh_t (backward)
  const float* W; int ldw;            // [K][cols], K-major: W_hh^T [H][3H] (forward: column g * H + unit) / W_hh [3H][H] (backward)
  const float* Ws = nullptr;          // nullable: the split-bf16 copy of W (kernels.h: TransposeBatch::add_split, lay = 32) -> gru_step_split_kernel
  int B, K, H, unit_blocks;
  const float *gi, *b_hh, *hprev;     // forward: gi_t [B][3H], b_hh [3H], h_{t-1} [B][H]
  float *h_out, *r_s, *z_s, *n_s, *hn_s;
  const float* carry; float* dh_out;  // backward: dh_t z_t [B][H] -> dh_{t-1} [B][H]
};

template <int NUT, bool FWD>
__global__ __launch_bounds__(256) void gru_step_kernel(GruStepArgs a) {
  constexpr int NU = 16 * NUT;                  // hidden units (forward) / dh columns (backward) per workgroup
  constexpr int NT = FWD ? 3 * NUT : NUT;       // column tiles of 16
  constexpr int NG = FWD ? 3 : 1;               // gates (column groups of NU)
  constexpr int LDP = 16 * NT + 4;              // LDS row stride of a partial tile
  extern __shared__ __attribute__((aligned(16))) float part[];   // [4 waves][32 rows][LDP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
  // XCD-aware tile order (workgroup b runs on XCD b % 8)
  int ub, rb;
  const int bid = blockIdx.x;
  if (a.unit_blocks % 8 == 0) {
    const int per = a.unit_blocks / 8, xcd = bid & 7, loc = bid >> 3;
    ub = xcd * per + loc % per;
    rb = loc / per;
  } else {
    ub = bid % a.unit_blocks;
    rb = bid / a.unit_blocks;
  }
  const int m0 = rb * GS_ROWS, u0 = ub * NU;
  const int KQ = a.K / 4, k0 = w * KQ + 4 * kq;
  const float* ap[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) ap[rt] = a.A + (long long)min(m0 + 16 * rt + c16, a.B - 1) * a.lda + k0;
  // W is K-MAJOR ([K][cols]) and a lane reads NUT CONSECUTIVE columns of a k row in one load: the 16 lanes of a k row cover the
  // workgroup's 16 NUT columns of a gate contiguously (192 B at NUT = 3), four k rows per wave instruction.  Column tile t of a gate then
  // holds the columns u0 + NUT c16 + t (a permutation undone where the partial tiles are written); MFMA i of a 16-wide k block
  // multiplies k = 4 kq + i on both operands.  ([cols][K] with one 16-byte load per lane -- 16 rows x 64 B per wave instruction -- ran
  // at 26 us a step; one dword per (column, k) is 38 loads a block, and three blocks of those overflow the 6-bit count of loads in flight.)
  const float* wp[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wp[g] = a.W + (long long)k0 * a.ldw + (FWD ? g * a.H : 0) + u0 + NUT * c16;
  const int ldw = a.ldw;
  struct WFrag { float v[4][NUT]; };            // [k step i][column tile t]
  auto wload = [&](const float* p, int kb) -> WFrag {   // k block kb (multiple of 16) of this lane's NUT columns
    WFrag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* q = p + (long long)(kb + i) * ldw;
      if constexpr (NUT == 3) { const GsF3 x = *(const GsF3*)q; f.v[i][0] = x.a; f.v[i][1] = x.b; f.v[i][2] = x.c; }   // global_load_dwordx3 (dword-aligned)
      else if constexpr (NUT == 2) { const GsF2 x = *(const GsF2*)q; f.v[i][0] = x.a; f.v[i][1] = x.b; }
      else f.v[i][0] = q[0];
    }
    return f;
  };
  // the epilogue's operands (thread -> EPT (row, unit) pairs) are requested HERE, in front of the product they do not depend on: behind it
  // they were EPT dependent round trips of a lone wave per SIMD
  constexpr int EPT = GS_ROWS * NU / 256;
  float e_gi[FWD ? EPT : 1][3], e_bh[FWD ? EPT : 1][3], e_x[EPT];     // forward: gi_t, b_hh (r, z, n), h_{t-1}; backward: the carry
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    const int e = tid + 256 * it, m = e / NU, u = e % NU, b = min(m0 + m, a.B - 1), j = u0 + u;
    if constexpr (FWD) {
      const float* gp = a.gi + (long long)b * 3 * a.H;
#pragma unroll
      for (int g = 0; g < 3; ++g) { e_gi[it][g] = gp[g * a.H + j]; e_bh[it][g] = a.b_hh[g * a.H + j]; }
      e_x[it] = a.hprev[(long long)b * a.H + j];
    } else {
      e_x[it] = a.carry[(long long)b * a.H + j];
    }
  }
  floatx4 acc[2][NT];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
  // fragments of NS 16-wide k blocks in a ring: a block is requested NS - 1 blocks before it is consumed -- ONE wave per SIMD lives here
  // (72 accumulator + 44 fragment registers per stage at NT = 9), so nothing else hides an L2 round trip
  constexpr int NS = FWD ? 4 : 6;
  float4 af[NS][2];
  WFrag wf[NS][NG];
  const int ns = KQ / 16;
#pragma unroll
  for (int p = 0; p < NS - 1; ++p) {
    const int sp = min(p, ns - 1) * 16;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) af[p][rt] = *(const float4*)(ap[rt] + sp);
#pragma unroll
    for (int g = 0; g < NG; ++g) wf[p][g] = wload(wp[g], sp);
  }
  for (int s = 0; s < ns; s += NS) {
#pragma unroll
    for (int ph = 0; ph < NS; ++ph) {
      const int cur = ph, nxt = (ph + NS - 1) % NS;
      const int sn = min(s + ph + NS - 1, ns - 1) * 16;     // (beyond the end: re-reads the last block, no branch around a load)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) af[nxt][rt] = *(const float4*)(ap[rt] + sn);
#pragma unroll
      for (int g = 0; g < NG; ++g) wf[nxt][g] = wload(wp[g], sn);
      if (s + ph < ns) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              const float av = i == 0 ? af[cur][rt].x : i == 1 ? af[cur][rt].y : i == 2 ? af[cur][rt].z : af[cur][rt].w;
              acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[cur][ct / NUT].v[i][ct % NUT], acc[rt][ct], 0, 0, 0);
            }
      }
    }
  }
  // partial tiles -> LDS (accumulator register r: row 4 kq + r, column c16)
  float* mine = part + (long long)w * GS_ROWS * LDP;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r)   // tile ct = (gate, t): its column c16 is column NUT c16 + t of the gate
        mine[(16 * rt + 4 * kq + r) * LDP + (ct / NUT) * NU + NUT * c16 + ct % NUT] = acc[rt][ct][r];
  __syncthreads();
  // epilogue: thread -> (row, unit); the four waves' partials in wave order
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    const int e = tid + 256 * it, m = e / NU, u = e % NU, b = m0 + m;
    if (b >= a.B) continue;
    if constexpr (FWD) {
      float pr = 0.f, pz = 0.f, pn = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* p = part + ((long long)q * GS_ROWS + m) * LDP;
        pr += p[u]; pz += p[NU + u]; pn += p[2 * NU + u];
      }
      const float rr = 1.0f / (1.0f + expf(-(e_gi[it][0] + (pr + e_bh[it][0]))));
      const float zz = 1.0f / (1.0f + expf(-(e_gi[it][1] + (pz + e_bh[it][1]))));
      const float hn = pn + e_bh[it][2];
      const float nn = tanhf(e_gi[it][2] + rr * hn);
      const long long o = (long long)b * a.H + u0 + u;
      a.h_out[o] = (1.0f - zz) * nn + zz * e_x[it];
      a.r_s[o] = rr; a.z_s[o] = zz; a.n_s[o] = nn; a.hn_s[o] = hn;
    } else {
      float sacc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) sacc += part[((long long)q * GS_ROWS + m) * LDP + u];
      a.dh_out[(long long)b * a.H + u0 + u] = sacc + e_x[it];
    }
  }
}

// The same launch in split-bf16 arithmetic (round 6d; gemm.hip: every fp32 value = the exact sum of three bf16 pieces, six piece products
// per product accumulated in fp32): the step product is MATRIX-PIPE bound (one wave per SIMD, 0.47 / 0.57 of the fp32-input rate), and
// v_mfma_f32_16x16x32_bf16 does a 16 x 16 x 32 block of six piece products in 6 x 4 passes where v_mfma_f32_16x16x4_f32 takes 8 x 8.
// Same decomposition, same partial tiles, same epilogue.  W comes PRE-SPLIT (a.Ws, made once per pass: [K/32][piece][k group of 4]
// [column][8 consecutive k] bf16 -- a lane's B fragment of a column tile is one 16-byte load per piece, the 16 lanes of a k group read
// 256 contiguous bytes); the A fragment (8 consecutive k of the lane's row: two float4 loads) is split in registers -- the four waves
// split K, so no value is split twice.  K / 4 must be a multiple of 32 (H % 128 == 0).
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int gs_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int gs_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gs_split8(const float4 a0, const float4 a1, gs_bf16x8 (&pc)[3]) {
  const gs_f32x2 x[4] = {{a0.x, a0.y}, {a0.z, a0.w}, {a1.x, a1.y}, {a1.z, a1.w}};
  gs_u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const gs_f32x2 r = x[j] - __builtin_bit_cast(gs_f32x2, __builtin_bit_cast(gs_u32x2, x[j]) & 0xFFFF0000u);
    const gs_f32x2 t = r - __builtin_bit_cast(gs_f32x2, __builtin_bit_cast(gs_u32x2, r) & 0xFFFF0000u);
    h[j] = __builtin_amdgcn_perm(__float_as_uint(x[j][1]), __float_as_uint(x[j][0]), 0x07060302u);
    m[j] = __builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u);
    l[j] = __builtin_amdgcn_perm(__float_as_uint(t[1]), __float_as_uint(t[0]), 0x07060302u);
  }
  pc[0] = __builtin_bit_cast(gs_bf16x8, h); pc[1] = __builtin_bit_cast(gs_bf16x8, m); pc[2] = __builtin_bit_cast(gs_bf16x8, l);
}

template <int NUT, bool FWD>
__global__ __launch_bounds__(256) void gru_step_split_kernel(GruStepArgs a) {
  constexpr int NU = 16 * NUT;
  constexpr int NT = FWD ? 3 * NUT : NUT;
  constexpr int NG = FWD ? 3 : 1;
  constexpr int LDP = 16 * NT + 4;
  extern __shared__ __attribute__((aligned(16))) float part[];   // [4 waves][32 rows][LDP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, kq = lane >> 4;
  int ub, rb;
  const int bid = blockIdx.x;
  if (a.unit_blocks % 8 == 0) {
    const int per = a.unit_blocks / 8, xcd = bid & 7, loc = bid >> 3;
    ub = xcd * per + loc % per;
    rb = loc / per;
  } else {
    ub = bid % a.unit_blocks;
    rb = bid / a.unit_blocks;
  }
  const int m0 = rb * GS_ROWS, u0 = ub * NU;
  const int KQ = a.K / 4, ns = KQ / 32;
  const int kb0 = __builtin_amdgcn_readfirstlane(w) * ns;   // this wave's 32-wide k blocks: kb0 .. kb0 + ns - 1 (an SGPR: the scalar offset of
                                                            // every W load derives from it -- as a VGPR each load became a waterfall loop)
  const float* ap[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) ap[rt] = a.A + (long long)min(m0 + 16 * rt + c16, a.B - 1) * a.lda + w * KQ + 8 * kq;
  // cell (kb, piece, k group, column) of the split copy = 16 B at ((kb * 3 + piece) * 4 + k group) * N + column: the lane part (k group,
  // its column inside a tile) is ONE 32-bit offset, everything else is wave-uniform and goes into the scalar offset
  const int N = a.ldw;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.Ws, 0, 0x7fffffff, 0x00020000);
  const int wvoff = (kq * N + u0 + c16) * 16;
  constexpr int EPT = GS_ROWS * NU / 256;
  float e_gi[FWD ? EPT : 1][3], e_bh[FWD ? EPT : 1][3], e_x[EPT];
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    const int e = tid + 256 * it, m = e / NU, u = e % NU, b = min(m0 + m, a.B - 1), j = u0 + u;
    if constexpr (FWD) {
      const float* gp = a.gi + (long long)b * 3 * a.H;
#pragma unroll
      for (int g = 0; g < 3; ++g) { e_gi[it][g] = gp[g * a.H + j]; e_bh[it][g] = a.b_hh[g * a.H + j]; }
      e_x[it] = a.hprev[(long long)b * a.H + j];
    } else {
      e_x[it] = a.carry[(long long)b * a.H + j];
    }
  }
  floatx4 acc[2][NT];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
  // a ring of NS 32-wide k blocks, requested NS - 1 blocks ahead (forward: 27 + 4 loads and 108 + 16 registers a block: two blocks are
  // what the 6-bit count of loads in flight and the register file hold; backward: 9 + 4 loads a block)
  constexpr int NS = FWD ? 2 : 4;
  float4 af[NS][2][2];
  gs_bf16x8 wf[NS][NT][3];
  auto fetch = [&](int slot, int blk) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      af[slot][rt][0] = *(const float4*)(ap[rt] + blk * 32);
      af[slot][rt][1] = *(const float4*)(ap[rt] + blk * 32 + 4);
    }
    const int sb = (kb0 + blk) * 3 * 4 * N;      // cells in front of this block
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        wf[slot][ct][q] = __builtin_bit_cast(gs_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, (sb + q * 4 * N + (FWD ? (ct / NUT) * a.H : 0) + 16 * (ct % NUT)) * 16, 0));
  };
#pragma unroll
  for (int p = 0; p < NS - 1; ++p) fetch(p, min(p, ns - 1));
  for (int s = 0; s < ns; s += NS) {
#pragma unroll
    for (int ph = 0; ph < NS; ++ph) {
      const int cur = ph, nxt = (ph + NS - 1) % NS;
      fetch(nxt, min(s + ph + NS - 1, ns - 1));          // (beyond the end: re-reads the last block, no branch around a load)
      if (s + ph < ns) {
        gs_bf16x8 pa[2][3];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) gs_split8(af[cur][rt][0], af[cur][rt][1], pa[rt]);
        // small terms first, the leading product last (gemm_tn_split_kernel's order)
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int tm = 0; tm < 6; ++tm)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
              acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[rt][PA[tm]], wf[cur][ct][PB[tm]], acc[rt][ct], 0, 0, 0);
      }
    }
  }
  // partial tiles -> LDS (accumulator register r: row 4 kq + r, column c16 of tile ct = columns 16 t .. 16 t + 15 of its gate)
  float* mine = part + (long long)w * GS_ROWS * LDP;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(16 * rt + 4 * kq + r) * LDP + (ct / NUT) * NU + 16 * (ct % NUT) + c16] = acc[rt][ct][r];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    const int e = tid + 256 * it, m = e / NU, u = e % NU, b = m0 + m;
    if (b >= a.B) continue;
    if constexpr (FWD) {
      float pr = 0.f, pz = 0.f, pn = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* p = part + ((long long)q * GS_ROWS + m) * LDP;
        pr += p[u]; pz += p[NU + u]; pn += p[2 * NU + u];
      }
      const float rr = 1.0f / (1.0f + expf(-(e_gi[it][0] + (pr + e_bh[it][0]))));
      const float zz = 1.0f / (1.0f + expf(-(e_gi[it][1] + (pz + e_bh[it][1]))));
      const float hn = pn + e_bh[it][2];
      const float nn = tanhf(e_gi[it][2] + rr * hn);
      const long long o = (long long)b * a.H + u0 + u;
      a.h_out[o] = (1.0f - zz) * nn + zz * e_x[it];
      a.r_s[o] = rr; a.z_s[o] = zz; a.n_s[o] = nn; a.hn_s[o] = hn;
    } else {
      float sacc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) sacc += part[((long long)q * GS_ROWS + m) * LDP + u];
      a.dh_out[(long long)b * a.H + u0 + u] = sacc + e_x[it];
    }
  }
}

// units per workgroup (in 16s) of the step kernels, 0 = not applicable: the widest tile that still gives ~a workgroup per CU
static int gru_step_nut(int B, int H) {
  static const bool off = ur_test_hook("gru_no_step") != 0;   // test hook: the gemm_nt + cell-kernel path (what H % 64 != 0 takes)
  if (off || H % 64 != 0) return 0;
  const int rbs = cdiv(B, GS_ROWS);
  for (int nut = 3; nut >= 1; --nut)
    if (H % (16 * nut) == 0 && (nut == 1 || (long long)rbs * (H / (16 * nut)) >= 200)) return nut;
  return 0;
}
// the step kernels' arithmetic: split bf16 when the cfg's mfma_arith names a split form and the shape allows (test hook gru_step_split=0: exact)
static bool gru_step_split_on(const UrGruCfg& c, bool fwd) {
  // measured (tools/r6_gru3.sh, ms per encoder step, exact / both / forward only / backward only): H = 768 3.63 / 3.47 / 3.47 / 3.64,
  // H = 512 2.30 / 2.26 / 2.26 / 2.31, H = 256 1.38 / 1.43 / 1.36 / 1.44 -- the forward launch (nine column tiles a wave) was matrix-pipe
  // bound, the backward one (three) is bound by its operand stream, which the 6-byte weights make longer: the FORWARD sweep only
  static const int hook = ur_test_hook("gru_step_split", 2);   // 0: exact, 1: both sweeps, 2 / 3: the forward / backward sweep only
  const int base = c.mfma_arith & 0xFF;
  static const int hmin = ur_test_hook("gru_step_split_hmin", 128);
  return (hook == 1 || hook == (fwd ? 2 : 3)) && (base == 6 || base == 9) && c.H % 128 == 0 && c.H >= hmin;
}
// the split copy of W_hh one pass's step launches stream: forward W_hh^T [K = H][3H], backward W_hh as stored [K = 3H][H]
static int gru_split_whh(const float* w_hh, int H, float* dst, bool fwd, hipStream_t st) {
  TransposeBatch tb;
  tb.add_split(w_hh, 3 * H, H, dst, !fwd, 32);
  return transpose_batch(tb, st);
}
template <bool FWD>
static int gru_step_launch(int nut, GruStepArgs a, hipStream_t st) {
  a.unit_blocks = a.H / (16 * nut);
  const int grid = cdiv(a.B, GS_ROWS) * a.unit_blocks;
  const int nt = FWD ? 3 * nut : nut;
  const size_t lds = (size_t)4 * GS_ROWS * (16 * nt + 4) * sizeof(float);
  ProfScope ps(PC_GRU, st, 2.0 * a.B * (double)a.K * (FWD ? 3.0 * a.H : (double)a.H), true);   // (the launch's own start / end: 100 launches a pass)
#define GO(N_) do {                                                                                                                   \
    static bool big_lds_set = false;   /* (76 KB at 48 units forward: above the 64 KB a kernel gets without asking) */                 \
    if (lds > 64 * 1024 && !big_lds_set) {                                                                                            \
      UR_HIP(hipFuncSetAttribute((const void*)gru_step_kernel<N_, FWD>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));        \
      big_lds_set = true;                                                                                                             \
    }                                                                                                                                 \
    if (a.Ws) {                                                                                                                       \
      static bool big_lds_set_s = false;                                                                                              \
      if (lds > 64 * 1024 && !big_lds_set_s) {                                                                                        \
        UR_HIP(hipFuncSetAttribute((const void*)gru_step_split_kernel<N_, FWD>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
        big_lds_set_s = true;                                                                                                         \
      }                                                                                                                               \
      UR_LAUNCH_EV((gru_step_split_kernel<N_, FWD>), dim3(grid), dim3(256), lds, st, a);                                              \
    } else                                                                                                                            \
    UR_LAUNCH_EV((gru_step_kernel<N_, FWD>), dim3(grid), dim3(256), lds, st, a);                                                      \
  } while (0)
  switch (nut) {
    case 3: GO(3); break;
    case 2: GO(2); break;
    default: GO(1); break;
  }
#undef GO
  UR_LAUNCH_CHECK();
  return UR_OK;
}

static bool gru_seq4_supported(int H) { return H == 64 || H == 128; }

static bool gru_seq_supported(int H) {
  static const bool off = ur_test_hook("gru_no_seq") != 0;   // test hook
  return !off && (H == 32 || H == 64 || H == 128);   // (H = 64 / 128: the four-sequence kernels, H = 32: the 16-sequence ones)
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
  UR_TRACE_SCOPE();
  int rc = gru_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(item_table && dense && item_seq && user_emb && ws && n_items > 0, UR_ERR_ARG, "ur_gru_fwd: null pointer");
  const UrGruCfg& c = *cfg;
  const ur::ArithScope arith_scope(c.mfma_arith);   // (the input projection: gemm.hip dispatch_tile)
  hipStream_t st = as_stream(stream);
  const GruLayout lay = gru_layout(c);
  GruWs w = gru_carve(c, (float*)ws);
  const int B = c.B, L = c.L, d = c.d, H = c.H, M = B * L;
  {
    ProfScope ps(PC_GRU, st, 0);
    hipLaunchKernelGGL(ids_time_major_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, item_seq, B, L, w.seq_tm);
    UR_LAUNCH_CHECK();
  }
  if ((rc = gather_rows(item_table, w.seq_tm, 4, M, d, w.x, st, n_items))) return rc;
  if (c.p_drop > 0.f && (rc = drop_rows(w.x, M, d, drop_spec(c.p_drop, c.drop_seed, c.drop_step, 0), w.x, st))) return rc;
  GemmArgs g{};
  g.A = w.x; g.lda = d; g.W = dense + lay.w_ih; g.ldw = d; g.C = w.gi; g.ldc = 3 * H; g.M = M; g.N = 3 * H; g.K = d; g.bias = dense + lay.b_ih;
  if ((rc = gemm_nt(g, PRO_NONE, EPI_BIAS, st))) return rc;
  UR_HIP(hipMemsetAsync(w.h_all, 0, sizeof(float) * B * H, st));
  if (gru_seq_supported(H)) {   // the whole recurrence in one launch
    ProfScope ps(PC_GRU, st, 2.0 * B * L * 3.0 * H * H);   // h_{t-1} W_hh^T of every step
    if (gru_seq4_supported(H)) {   // four sequences per workgroup (4x4x1 MFMA): B / 4 workgroups
#define GO4(HH) GO4R(HH, 4)
#define GO4R(HH, RR) hipLaunchKernelGGL((gru_seq4_fwd_kernel<HH, RR>), dim3(cdiv(B, RR)), dim3(512), 0, st, w.gi, dense + lay.w_hh, dense + lay.b_hh, \
                                   B, L, w.h_all, w.r, w.z, w.n, w.hn)
      // two rows per workgroup while that still leaves CUs free at four (B <= 640 on 256 CUs): test hook gru_rows = 2 / 4 forces
      const int rows = gru_seq4_rows(B);
      if (H == 64) { if (rows == 2) GO4R(64, 2); else GO4R(64, 4); }
      else { if (rows == 2) GO4R(128, 2); else GO4R(128, 4); }
#undef GO4
#undef GO4R
    } else {
      const dim3 grid(cdiv(B, GRU_SEQ_ROWS));
#define GO(HH) hipLaunchKernelGGL((gru_seq_fwd_kernel<HH>), grid, dim3(HH * 4), 0, st, w.gi, dense + lay.w_hh, dense + lay.b_hh, B, L, w.h_all, \
                                  w.r, w.z, w.n, w.hn)
      GO(32);
#undef GO
    }
    UR_LAUNCH_CHECK();
  } else if (const int nut = gru_step_nut(B, H)) {   // wide hidden state: one fused launch per step (product + gates)
    if ((rc = transpose(dense + lay.w_hh, 3 * H, H, w.w_hhT, st))) return rc;   // [H, 3H]: the K-major operand of the step kernel
    const bool ssp = gru_step_split_on(c, true) && w.s_whh;
    if (ssp && (rc = gru_split_whh(dense + lay.w_hh, H, w.s_whh, true, st))) return rc;
    for (int t = 0; t < L; ++t) {
      const long long o = (long long)t * B * H;
      GruStepArgs sa{};
      sa.A = w.h_all + o; sa.lda = H; sa.W = w.w_hhT; sa.ldw = 3 * H; sa.B = B; sa.K = H; sa.H = H;
      if (ssp) sa.Ws = w.s_whh;
      sa.gi = w.gi + (long long)t * B * 3 * H; sa.b_hh = dense + lay.b_hh; sa.hprev = w.h_all + o;
      sa.h_out = w.h_all + o + (long long)B * H; sa.r_s = w.r + o; sa.z_s = w.z + o; sa.n_s = w.n + o; sa.hn_s = w.hn + o;
      if ((rc = gru_step_launch<true>(nut, sa, st))) return rc;
    }
  } else
  for (int t = 0; t < L; ++t) {
    const long long o = (long long)t * B * H;
    g = GemmArgs{};
    g.A = w.h_all + o; g.lda = H; g.W = dense + lay.w_hh; g.ldw = H; g.C = w.gh; g.ldc = 3 * H; g.M = B; g.N = 3 * H; g.K = H;
    g.ksplit = gru_ksplit(H); g.split_stride = (long long)B * 3 * H;   // (the cell kernel sums the pieces and adds b_hh)
    if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
    ProfScope ps(PC_GRU, st, 0);
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(cdiv((long long)B * H, 256)), dim3(256), 0, st, w.gi + (long long)t * B * 3 * H, w.gh,
                       g.ksplit, dense + lay.b_hh, w.h_all + o, B, H, w.h_all + o + (long long)B * H, w.r + o, w.z + o, w.n + o, w.hn + o);
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
  UR_TRACE_SCOPE();
  const ur::ArithScope arith_scope(cfg ? cfg->mfma_arith : 0);   // (the weight-gradient products of this pass: gemm.hip)
  int rc = gru_check(cfg);
  if (rc) return rc;
  UR_REQUIRE(dense && d_user_emb && ws && dense_grad && d_emb_rows, UR_ERR_ARG, "ur_gru_bwd: null pointer");
  (void)item_table; (void)n_items; (void)item_seq;
  const UrGruCfg& c = *cfg;
  hipStream_t st = as_stream(stream);
  const GruLayout lay = gru_layout(c);
  GruWs w = gru_carve(c, (float*)ws);
  const int B = c.B, L = c.L, d = c.d, H = c.H, M = B * L;
  {   // the K-major copies this pass's GEMMs read, one launch (w_hhT: only the GEMM-per-step path reads it)
    TransposeBatch tb;
    tb.add(dense + lay.w_ih, 3 * H, d, w.w_ihT);   // [d, 3H]
    tb.add(dense + lay.w_d, d, H, w.w_dT);         // [H, d]
    if (!gru_seq_supported(H) && !gru_step_nut(B, H)) tb.add(dense + lay.w_hh, 3 * H, H, w.w_hhT);   // [H, 3H]
    if ((rc = transpose_batch(tb, st))) return rc;
  }
  // dense head: dW_d = d_out^T h_L ; db_d ; dh_L = d_out W_d
  const float* hL = w.h_all + (long long)L * B * H;
  if ((rc = gemm_tn(d_user_emb, d, hL, H, B, d, H, 0, 0, dense_grad + lay.w_d, H, dense_grad + lay.b_d, w.tn_ws, st))) return rc;
  GemmArgs g{};
  g.A = d_user_emb; g.lda = d; g.W = w.w_dT; g.ldw = d; g.C = w.dh; g.ldc = H; g.M = B; g.N = H; g.K = d;
  if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
  if (gru_seq_supported(H)) {   // the whole backward sweep in one launch
    ProfScope ps(PC_GRU, st, 2.0 * B * L * 3.0 * H * H);   // dgh_t W_hh of every step
    if (gru_seq4_supported(H)) {
#define GO4(HH) GO4R(HH, 4)
#define GO4R(HH, RR) hipLaunchKernelGGL((gru_seq4_bwd_kernel<HH, RR>), dim3(cdiv(B, RR)), dim3(512), 0, st, w.dh, dense + lay.w_hh, w.r, w.z, w.n, w.hn, \
                                   w.h_all, B, L, w.dgi, w.dgh)
      const int rows = gru_seq4_rows(B);
      if (H == 64) { if (rows == 2) GO4R(64, 2); else GO4R(64, 4); }
      else { if (rows == 2) GO4R(128, 2); else GO4R(128, 4); }
#undef GO4
#undef GO4R
    } else {
      const dim3 grid(cdiv(B, GRU_SEQ_ROWS));
#define GO(HH) hipLaunchKernelGGL((gru_seq_bwd_kernel<HH>), grid, dim3(HH * 4), 0, st, w.dh, dense + lay.w_hh, w.r, w.z, w.n, w.hn, w.h_all, B, L, \
                                  w.dgi, w.dgh)
      GO(32);
#undef GO
    }
    UR_LAUNCH_CHECK();
  } else if (const int nut = gru_step_nut(B, H)) {   // wide hidden state: cell backward + one fused launch (product + carry) per step
    const float* dh_cur = w.dh;
    const bool ssp = gru_step_split_on(c, false) && w.s_whh;
    if (ssp && (rc = gru_split_whh(dense + lay.w_hh, H, w.s_whh, false, st))) return rc;
    for (int t = L - 1; t >= 0; --t) {
      const long long o = (long long)t * B * H, o3 = (long long)t * B * 3 * H;
      {
        ProfScope ps(PC_GRU, st, 0, true);
        UR_LAUNCH_EV(gru_cell_bwd_kernel, dim3(cdiv((long long)B * H, 256)), dim3(256), 0, st, dh_cur, 0, w.r + o, w.z + o, w.n + o,
                     w.hn + o, w.h_all + o, B, H, w.dgi + o3, w.dgh + o3, w.dh_carry);
        UR_LAUNCH_CHECK();
      }
      if (t == 0) break;   // (dh_{-1} has no reader: h_0 = 0)
      GruStepArgs sa{};
      sa.A = w.dgh + o3; sa.lda = 3 * H; sa.W = dense + lay.w_hh; sa.ldw = H; sa.B = B; sa.K = 3 * H; sa.H = H;
      if (ssp) sa.Ws = w.s_whh;
      sa.carry = w.dh_carry; sa.dh_out = w.dh_parts;
      if ((rc = gru_step_launch<false>(nut, sa, st))) return rc;
      dh_cur = w.dh_parts;
    }
  } else
  for (int t = L - 1; t >= 0; --t) {
    const long long o = (long long)t * B * H, o3 = (long long)t * B * 3 * H;
    {
      ProfScope ps(PC_GRU, st, 0);
      const bool first = t == L - 1;   // (the first step of the sweep reads the head's gradient; the others the pieces of the step before)
      hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(cdiv((long long)B * H, 256)), dim3(256), 0, st, first ? w.dh : w.dh_parts,
                         first ? 0 : gru_ksplit(3 * H), w.r + o, w.z + o, w.n + o, w.hn + o, w.h_all + o, B, H, w.dgi + o3, w.dgh + o3, w.dh_carry);
      UR_LAUNCH_CHECK();
    }
    if (t == 0) break;   // (dh_{-1} has no reader: h_0 = 0)
    g = GemmArgs{};   // the pieces of dgh_t W_hh; dh_{t-1} = their sum + dh_t * z is formed by the next cell kernel
    g.A = w.dgh + o3; g.lda = 3 * H; g.W = w.w_hhT; g.ldw = 3 * H; g.C = w.dh_parts; g.ldc = H; g.M = B; g.N = H; g.K = 3 * H;
    g.ksplit = gru_ksplit(3 * H); g.split_stride = (long long)B * H;
    if ((rc = gemm_nt(g, PRO_NONE, EPI_NONE, st))) return rc;
  }
  // weight gradients over all (t, b) tokens at once (time-major rows on both operands)
  {   // (one grouped launch: dW_hh and dW_ih share the token dimension.  On a side stream under the input-gradient GEMM below -- fork event,
      // join at the end of the call -- the step was SLOWER: H = 128 0.451 / 0.436, H = 768 3.65 / 3.48 ms, profiles/r06_m_gru_nt_split.txt)
    const TnReq rq[2] = {{w.dgh, 3 * H, w.h_all, H, M, 3 * H, H, 0, 0, dense_grad + lay.w_hh, H, dense_grad + lay.b_hh, w.tn_ws, nullptr},
                         {w.dgi, 3 * H, w.x, d, M, 3 * H, d, 0, 0, dense_grad + lay.w_ih, d, dense_grad + lay.b_ih, w.tn_ws2, nullptr}};
    if ((rc = gemm_tn_group(rq, 2, st))) return rc;
  }
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
