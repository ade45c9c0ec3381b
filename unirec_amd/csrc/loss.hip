// Fused candidate gather + dot-product scorer + loss, forward and backward.
// One workgroup per batch row b; a group of TPR lanes owns one candidate at a time: it reads the
// candidate's table row once (coalesced float4), dots it with the user vector held in registers and
// emits one float -- the [B,G,d] candidate tensor of the reference never exists.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace ur {

constexpr float kEps = 1e-8f;  // unirec/constants/global_variables.py:4
constexpr int MAXV = 4;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = group_max<64>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = -INFINITY;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s = fmaxf(s, red[w]);
  return s;
}

// ---- the loss of ONE score row sc[0..G) (LDS; tau and the clip applied): numerator `part` and denominator share `cnt` of the batch mean
// (AbstractRecommender._cal_loss, unirec/model/base/reco_abc.py:238-266; modules.py:15-67).  Every thread of the workgroup calls it.
__device__ __forceinline__ void row_loss(const UrLossCfg& c, const int G, const float* sc, const int* __restrict__ label, float* red,
                                         float& part, float& cnt) {
  part = 0.f; cnt = 0.f;
  if (c.loss_type < 0) {  // UR_LOSS_NONE: scores only
    cnt = 1.f;
  } else if (c.loss_type == UR_LOSS_BPR) {
    const float s0 = sc[0];
    for (int g = 1 + threadIdx.x; g < G; g += blockDim.x) part += -logf(kEps + 1.0f / (1.0f + expf(-(s0 - sc[g]))));
    part = block_sum(part, red) / (float)(G - 1);
    cnt = 1.f;
  } else if (c.loss_type == UR_LOSS_SOFTMAX) {
    float m = -INFINITY;
    for (int g = threadIdx.x; g < G; g += blockDim.x) m = fmaxf(m, sc[g]);
    m = block_max(m, red);
    float l = 0.f;
    for (int g = threadIdx.x; g < G; g += blockDim.x) l += expf(sc[g] - m);
    const float lse = m + logf(block_sum(l, red));
    for (int g = threadIdx.x; g < G; g += blockDim.x)
      if (label[g] > 0) {
        part += lse - sc[g];
        cnt += 1.f;
      }
    part = block_sum(part, red);
    cnt = block_sum(cnt, red);
  } else if (c.loss_type == UR_LOSS_BCE) {
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      const float p = 1.0f / (1.0f + expf(-sc[g]));  // clamp(sigmoid, -EPS, 1-EPS) is the identity in fp32
      const float y = (float)label[g];
      part += -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
    }
    part = block_sum(part, red);
    cnt = (float)G;
  } else {  // CCL
    for (int g = 1 + threadIdx.x; g < G; g += blockDim.x) part += fmaxf(sc[g] - c.ccl_m, 0.f);
    part = 1.f - sc[0] + c.ccl_w * block_sum(part, red) / (float)(G - 1);
    cnt = 1.f;
  }
}

// NT threads per workgroup: 256, or 1024 for few rows with many candidates (C3: B = 128 rows of 1001 candidates leave half of the
// CUs without a workgroup; four times the lane groups per row = four times the candidate rows in flight)
// KV = float4 chunks of a row per lane = ceil(d / 4 / TPR): 1 for every d <= 128.  (Round 5: the kernel used to carry MAXV = 4 chunks for
// every shape -- 152 VGPRs at 256 threads, scratch spills at 1024 -- for loops whose upper three trips never run at d = 128; the gather
// ran at 0.70-0.72 of the HBM peak where the same loop without them reaches 0.77: tools/probe/scorer_probe.hip, profiles/r05_d_*.)
template <int TPR, int UNR, int NT, int KV>
__global__ __launch_bounds__(NT) void scorer_loss_fwd_kernel(UrLossCfg c, const float4* __restrict__ user_emb,
                                                              const float4* __restrict__ table, const long long* __restrict__ item_id,
                                                              const int* __restrict__ label, const float* __restrict__ user_bias,
                                                              const float* __restrict__ item_bias, const long long* __restrict__ user_id,
                                                              float* __restrict__ scores, float* __restrict__ loss_rows,
                                                              float* __restrict__ cnt_rows, long long n_items) {
  extern __shared__ float sc[];  // [G] scores of this row, then 16 floats of reduction scratch
  float* red = sc + c.G;
  const int b = blockIdx.x, G = c.G, d4 = c.d / 4;
  const int groups = NT / TPR, g0 = threadIdx.x / TPR, t = threadIdx.x % TPR;
  float4 u[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int col = t + k * TPR;
    u[k] = col < d4 ? user_emb[(long long)b * d4 + col] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float ub = user_bias ? user_bias[user_id[b]] : 0.f;
  const float inv_tau = 1.0f / c.tau;
  // UNR candidate rows in flight per lane group: the rows are random 512-B reads of a table far larger than any
  // cache, so memory-level parallelism (not arithmetic) sets the rate
  for (int gb = g0 * UNR; gb < G; gb += groups * UNR) {
    long long id[UNR];
    float s[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      // unconditional loads at a clamped position, the range check behind ALL of them: a load inside a select is branched around and
      // waited for on the spot (eight dependent round trips per trip: measured -5 % on the gather, profiles/r05_d_scorer_probe.txt)
      id[q] = item_id[(long long)b * G + min(gb + q, G - 1)];
      s[q] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) id[q] = UR_ROW(id[q], n_items);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int col = t + k * TPR;
      if (col < d4) {
        float4 e[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
          typedef float vf4_t __attribute__((ext_vector_type(4)));
          const vf4_t t4 = __builtin_nontemporal_load((const vf4_t*)&table[id[q] * d4 + col]);   // rows are read once: do not keep them in L2
          e[q] = make_float4(t4.x, t4.y, t4.z, t4.w);
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) s[q] += (e[q].x * u[k].x + e[q].y * u[k].y) + (e[q].z * u[k].z + e[q].w * u[k].w);
      }
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) s[q] = group_sum<TPR>(s[q]);
    if (t == 0) {
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const int g = gb + q;
        if (g < G) {
          float v = s[q] + ub;
          if (item_bias) v += item_bias[id[q]];
          sc[g] = v;
        }
      }
    }
  }
  (void)inv_tau;
  __syncthreads();
  // the division by tau, the clip and the stores in ONE coalesced pass over the row's LDS copy, every thread at work: a 4-byte store per
  // candidate row from inside the gather loop cost the gather 5-7 % (tools/probe/scorer_probe.hip "V2 without the global score store" =
  // the random-row ceiling, profiles/r05_d_*), and the divide ran on one lane of 32 there.  Same operations per score, same order.
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float v = sc[g] / c.tau;
    if (c.score_clip > 0.f) v = fminf(fmaxf(v, -c.score_clip), c.score_clip);
    sc[g] = v;
    scores[(long long)b * G + g] = v;
  }
  __syncthreads();
  // ---- per-row loss
  float part, cnt;
  row_loss(c, G, sc, label ? label + (long long)b * G : nullptr, red, part, cnt);
  if (threadIdx.x == 0) {
    loss_rows[b] = part;
    cnt_rows[b] = cnt;
  }
}

// loss = sum(loss_rows) / sum(cnt_rows)   (single block, fixed order)
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ loss_rows, const float* __restrict__ cnt_rows,
                                                          int B, float* __restrict__ out) {
  __shared__ float red[8];
  float s = 0.f, n = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    s += loss_rows[i];
    n += cnt_rows[i];
  }
  s = block_sum(s, red);
  n = block_sum(n, red);
  if (threadIdx.x == 0) {
    out[0] = s / n;
    out[1] = n;
    out[2] = (s / n) != (s / n) ? -1.0f : 1.0f;   // update guard: -1 when the loss is NaN (the optimizer kernels then skip the step)
  }
}

// ---- d (batch-mean loss) / d score' of ONE score row s[0..G) -> cf[0..G) (LDS), `total` = the mean's denominator; the caller
// scales by the upstream gradient / tau and applies the clip's gate.
__device__ __forceinline__ void row_coef(const UrLossCfg& c, const int G, const float* __restrict__ s, const int* __restrict__ label,
                                         float total, float* cf, float* red) {
  if (c.loss_type == UR_LOSS_BPR) {
    float a0 = 0.f;
    for (int g = 1 + threadIdx.x; g < G; g += blockDim.x) {
      const float sg = 1.0f / (1.0f + expf(-(s[0] - s[g])));
      const float w = sg * (1.f - sg) / (kEps + sg) / ((float)(G - 1) * total);
      cf[g] = w;
      a0 -= w;
    }
    a0 = block_sum(a0, red);
    if (threadIdx.x == 0) cf[0] = a0;
  } else if (c.loss_type == UR_LOSS_SOFTMAX) {
    float m = -INFINITY;
    for (int g = threadIdx.x; g < G; g += blockDim.x) m = fmaxf(m, s[g]);
    m = block_max(m, red);
    float l = 0.f, np = 0.f;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      l += expf(s[g] - m);
      np += (label[g] > 0) ? 1.f : 0.f;
    }
    l = block_sum(l, red);
    np = block_sum(np, red);
    for (int g = threadIdx.x; g < G; g += blockDim.x)
      cf[g] = (np * expf(s[g] - m) / l - ((label[g] > 0) ? 1.f : 0.f)) / total;
  } else if (c.loss_type == UR_LOSS_BCE) {
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      const float p = 1.0f / (1.0f + expf(-s[g]));
      const float y = (float)label[g];
      // d/ds of -(y log p + (1-y) log(1-p)) with torch's log clamp at -100 (gradient 0 where clamped)
      const float gp = (logf(p) > -100.f ? y * (1.f - p) : 0.f) - (logf(1.f - p) > -100.f ? (1.f - y) * p : 0.f);
      cf[g] = -gp / total;
    }
  } else {  // CCL
    for (int g = threadIdx.x; g < G; g += blockDim.x)
      cf[g] = (g == 0) ? -1.f / total : ((s[g] - c.ccl_m > 0.f) ? c.ccl_w / ((float)(G - 1) * total) : 0.f);
  }
}

template <int TPR, int NT, int KV>   // KV: float4 chunks of a row per lane (see the forward kernel)
__global__ __launch_bounds__(NT) void scorer_loss_bwd_kernel(UrLossCfg c, const float4* __restrict__ user_emb,
                                                              const float4* __restrict__ table, const long long* __restrict__ item_id,
                                                              const int* __restrict__ label, const float* __restrict__ scores,
                                                              const float* __restrict__ d_loss, const float* __restrict__ norm,
                                                              float* __restrict__ coef, float4* __restrict__ d_user,
                                                              float* __restrict__ d_user_bias_rows, long long n_items) {
  extern __shared__ float sh[];  // [G] coef, then [groups][d] partial d_user, then 16 scratch
  constexpr int groups = NT / TPR;
  const int b = blockIdx.x, G = c.G, d4 = c.d / 4, d = c.d;
  float* cf = sh;
  float* acc_lds = sh + G;
  float* red = acc_lds + groups * d;
  const int g0 = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const float up = (d_loss ? d_loss[0] : 1.0f);
  const float total = norm[1];  // denominator of the mean (rows, pairs or positives)
  const float* s = scores + (long long)b * G;
  // ---- d loss / d score'
  row_coef(c, G, s, label ? label + (long long)b * G : nullptr, total, cf, red);
  __syncthreads();
  float bsum = 0.f;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float v = cf[g] * up / c.tau;
    if (c.score_clip > 0.f && fabsf(s[g]) >= c.score_clip) v = 0.f;
    cf[g] = v;
    coef[(long long)b * G + g] = v;
    bsum += v;
  }
  if (d_user_bias_rows) {
    bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) d_user_bias_rows[b] = bsum;
  }
  __syncthreads();
  // ---- d_user[b,:] = sum_g coef[g] * E[item_id[b,g],:]
  float4 acc[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int UNR = KV == 1 ? 8 : 4;   // candidate rows in flight per lane group (see the forward kernel; 8 where one chunk per lane leaves the registers)
  for (int gb = g0 * UNR; gb < G; gb += groups * UNR) {
    long long id[UNR];
    float w[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const bool in = gb + q < G;
      id[q] = item_id[(long long)b * G + min(gb + q, G - 1)];   // (unconditional, clamped position: see the forward kernel)
      w[q] = in ? cf[gb + q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) id[q] = UR_ROW(id[q], n_items);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int col = t + k * TPR;
      if (col < d4) {
        float4 e[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
          typedef float vf4_t __attribute__((ext_vector_type(4)));
          const vf4_t t4 = __builtin_nontemporal_load((const vf4_t*)&table[id[q] * d4 + col]);   // rows are read once: do not keep them in L2
          e[q] = make_float4(t4.x, t4.y, t4.z, t4.w);
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
          acc[k].x = fmaf(w[q], e[q].x, acc[k].x); acc[k].y = fmaf(w[q], e[q].y, acc[k].y);
          acc[k].z = fmaf(w[q], e[q].z, acc[k].z); acc[k].w = fmaf(w[q], e[q].w, acc[k].w);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int col = t + k * TPR;
    if (col < d4) *(float4*)(acc_lds + g0 * d + col * 4) = acc[k];
  }
  __syncthreads();
  for (int col = threadIdx.x; col < d; col += blockDim.x) {
    float v = 0.f;
#pragma unroll
    for (int gg = 0; gg < groups; ++gg) v += acc_lds[gg * d + col];
    ((float*)d_user)[(long long)b * d + col] = v;
  }
}

// ---- training step in ONE launch: scores + per-row loss + d loss / d score + d_user, and the batch loss finished by the LAST
// workgroup.  ur_gather_dot_loss_fwd + ur_gather_dot_loss_bwd are three launches (scores / loss rows, the one-workgroup mean, the
// gradient) of 5-8 us each with nothing to hide their latency behind: at a few candidates per row (BPR with 4 negatives) the whole
// loss section of a step is launch latency.  The backward needs the denominator of the mean before it can scale a single
// coefficient -- for bpr / bce / ccl that is B or B * G, known on the host -- so one workgroup can carry a row from the gather to
// d_user: the candidate rows it gathered for the scores stay in LDS ([G][d], G * d * 4 <= 32 KB) and are the operand of
// d_user = sum_g coef_g E_g (no second gather).  The mean over the batch (and the NaN guard the optimizer kernels read) is summed by
// the workgroup that finishes last, in row order (fixed order: bit-reproducible), behind a device-scope fence and a counter that
// resets itself.  Same arithmetic per element as the two kernels above (softmax, whose denominator is data-dependent, keeps them).
template <int TPR>
__global__ __launch_bounds__(256) void scorer_loss_fused_kernel(UrLossCfg c, float total, const float4* __restrict__ user_emb,
                                                                const float4* __restrict__ table, const long long* __restrict__ item_id,
                                                                const int* __restrict__ label, const float* __restrict__ user_bias,
                                                                const float* __restrict__ item_bias, const long long* __restrict__ user_id,
                                                                float* __restrict__ scores, float* __restrict__ loss_rows,
                                                                float* __restrict__ cnt_rows, float* __restrict__ coef,
                                                                float* __restrict__ d_user, float* __restrict__ d_user_bias_rows,
                                                                float* __restrict__ loss_out, unsigned* __restrict__ done_counter,
                                                                int arrive_mode, long long n_items) {
  extern __shared__ __attribute__((aligned(16))) float sh[];   // [G][d] rows, [G] scores, [G] coefficients, 16 floats of scratch
  const int b = blockIdx.x, G = c.G, d4 = c.d / 4, d = c.d;
  float* rows = sh;
  float* sc = rows + (long long)G * d;
  float* cf = sc + G;
  float* red = cf + G;
  __shared__ int is_last;
  constexpr int groups = 256 / TPR, UNR = 4;
  const int g0 = threadIdx.x / TPR, t = threadIdx.x % TPR;
  float4 u[MAXV];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int col = t + k * TPR;
    u[k] = col < d4 ? user_emb[(long long)b * d4 + col] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float ub = user_bias ? user_bias[user_id[b]] : 0.f;
  for (int gb = g0 * UNR; gb < G; gb += groups * UNR) {
    long long id[UNR];
    float s[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      // unconditional loads at a clamped position, the range check behind ALL of them: a load inside a select is branched around and
      // waited for on the spot (eight dependent round trips per trip: measured -5 % on the gather, profiles/r05_d_scorer_probe.txt)
      id[q] = item_id[(long long)b * G + min(gb + q, G - 1)];
      s[q] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) id[q] = UR_ROW(id[q], n_items);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int col = t + k * TPR;
      if (col < d4) {
        float4 e[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
          typedef float vf4_t __attribute__((ext_vector_type(4)));
          const vf4_t t4 = __builtin_nontemporal_load((const vf4_t*)&table[id[q] * d4 + col]);   // rows are read once: do not keep them in L2
          e[q] = make_float4(t4.x, t4.y, t4.z, t4.w);
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
          s[q] += (e[q].x * u[k].x + e[q].y * u[k].y) + (e[q].z * u[k].z + e[q].w * u[k].w);
          if (gb + q < G) *(float4*)(rows + (long long)(gb + q) * d + col * 4) = e[q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) s[q] = group_sum<TPR>(s[q]);
    if (t == 0) {
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const int g = gb + q;
        if (g < G) {
          float v = s[q] + ub;
          if (item_bias) v += item_bias[id[q]];
          v = v / c.tau;
          if (c.score_clip > 0.f) v = fminf(fmaxf(v, -c.score_clip), c.score_clip);
          sc[g] = v;
          scores[(long long)b * G + g] = v;
        }
      }
    }
  }
  __syncthreads();
  // ---- per-row loss and d loss / d score (the arithmetic of scorer_loss_fwd_kernel / scorer_loss_bwd_kernel)
  float part = 0.f, cnt = 1.f;
  if (c.loss_type == UR_LOSS_BPR) {
    const float s0 = sc[0];
    float a0 = 0.f;
    for (int g = 1 + threadIdx.x; g < G; g += blockDim.x) {
      const float sg = 1.0f / (1.0f + expf(-(s0 - sc[g])));
      part += -logf(kEps + sg);
      const float w = sg * (1.f - sg) / (kEps + sg) / ((float)(G - 1) * total);
      cf[g] = w;
      a0 -= w;
    }
    part = block_sum(part, red) / (float)(G - 1);
    a0 = block_sum(a0, red);
    if (threadIdx.x == 0) cf[0] = a0;
  } else if (c.loss_type == UR_LOSS_BCE) {
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      const float p = 1.0f / (1.0f + expf(-sc[g]));  // clamp(sigmoid, -EPS, 1-EPS) is the identity in fp32
      const float y = (float)label[(long long)b * G + g];
      part += -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
      const float gp = (logf(p) > -100.f ? y * (1.f - p) : 0.f) - (logf(1.f - p) > -100.f ? (1.f - y) * p : 0.f);
      cf[g] = -gp / total;
    }
    part = block_sum(part, red);
    cnt = (float)G;
  } else {  // CCL
    for (int g = 1 + threadIdx.x; g < G; g += blockDim.x) part += fmaxf(sc[g] - c.ccl_m, 0.f);
    part = 1.f - sc[0] + c.ccl_w * block_sum(part, red) / (float)(G - 1);
    for (int g = threadIdx.x; g < G; g += blockDim.x)
      cf[g] = (g == 0) ? -1.f / total : ((sc[g] - c.ccl_m > 0.f) ? c.ccl_w / ((float)(G - 1) * total) : 0.f);
  }
  if (threadIdx.x == 0) {
    // device-scope atomics carry the two numbers to the workgroup that finishes last (maybe on another XCD, behind another L2) --
    // WITHOUT a device-scope fence, which on this part writes back the whole L2 of the XCD (the forward pass's activations are sitting
    // there dirty: 512 such fences cost the step 20 us)
    // (exchanges, not stores: a read-modify-write is performed at the point all XCDs agree on, and once its old value is back it HAS
    // been performed -- a store's acknowledgement does not say that its write-through has landed, and the counter could overtake it)
    const float o0 = __hip_atomic_exchange(loss_rows + b, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float o1 = __hip_atomic_exchange(cnt_rows + b, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(o0), "v"(o1));
  }
  __syncthreads();
  float bsum = 0.f;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float v = cf[g] / c.tau;
    if (c.score_clip > 0.f && fabsf(sc[g]) >= c.score_clip) v = 0.f;
    cf[g] = v;
    coef[(long long)b * G + g] = v;
    bsum += v;
  }
  if (d_user_bias_rows) {
    bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) d_user_bias_rows[b] = bsum;
  }
  __syncthreads();
  // ---- d_user[b,:] = sum_g coef[g] * E[item_id[b,g],:]   (rows from LDS, candidates in order)
  for (int col = threadIdx.x; col < d4; col += blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < G; ++g) {
      const float w = cf[g];
      const float4 e = *(const float4*)(rows + (long long)g * d + col * 4);
      acc.x = fmaf(w, e.x, acc.x); acc.y = fmaf(w, e.y, acc.y); acc.z = fmaf(w, e.z, acc.z); acc.w = fmaf(w, e.w, acc.w);
    }
    *(float4*)(d_user + (long long)b * d + col * 4) = acc;
  }
  // ---- the batch loss, by whichever workgroup finishes last
  if (threadIdx.x == 0) {
    const unsigned prev = ur_arrive(done_counter, arrive_mode);
    is_last = prev == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  float s = 0.f, n = 0.f;
  for (int i = threadIdx.x; i < c.B; i += blockDim.x) {
    s += __uint_as_float(__hip_atomic_fetch_or((unsigned*)loss_rows + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    n += __uint_as_float(__hip_atomic_fetch_or((unsigned*)cnt_rows + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  s = block_sum(s, red);
  n = block_sum(n, red);
  if (threadIdx.x == 0) {
    loss_out[0] = s / n;
    loss_out[1] = n;
    loss_out[2] = (s / n) != (s / n) ? -1.0f : 1.0f;   // update guard (see loss_finish_kernel)
    __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
}

// ---- group mode (UrLossCfg.group_size > 0): the `user-item-label` row format.  The batch holds ONE (user, item, label) triple per row
// and AbstractRecommender._cal_loss views the [rows] scores as [rows / group_size, group_size] before the loss
// (unirec/model/base/reco_abc.py:233-236): every group_size consecutive rows are one score row, each with its own user vector.  The scorer
// runs as the G = 1 kernel over the rows; these kernels apply the loss / its gradient to the regrouped scores (same row_loss / row_coef
// as the one-launch-per-row kernels), and the user gradient of a row is its coefficient times its single candidate row.
__global__ __launch_bounds__(256) void group_loss_fwd_kernel(UrLossCfg c, const float* __restrict__ scores, const int* __restrict__ label,
                                                             float* __restrict__ loss_rows, float* __restrict__ cnt_rows) {
  extern __shared__ float sc[];   // [group_size] scores, then 16 floats of reduction scratch
  const int gs = c.group_size, b = blockIdx.x;
  float* red = sc + gs;
  for (int g = threadIdx.x; g < gs; g += blockDim.x) sc[g] = scores[(long long)b * gs + g];
  __syncthreads();
  float part, cnt;
  row_loss(c, gs, sc, label ? label + (long long)b * gs : nullptr, red, part, cnt);
  if (threadIdx.x == 0) {
    loss_rows[b] = part;
    cnt_rows[b] = cnt;
  }
}
__global__ __launch_bounds__(256) void group_loss_bwd_kernel(UrLossCfg c, const float* __restrict__ scores, const int* __restrict__ label,
                                                             const float* __restrict__ d_loss, const float* __restrict__ norm,
                                                             float* __restrict__ coef, float* __restrict__ d_user_bias_rows) {
  extern __shared__ float cf[];   // [group_size] coefficients, then 16 floats of reduction scratch
  const int gs = c.group_size, b = blockIdx.x;
  float* red = cf + gs;
  const float* s = scores + (long long)b * gs;
  row_coef(c, gs, s, label ? label + (long long)b * gs : nullptr, norm[1], cf, red);
  __syncthreads();
  const float up = d_loss ? d_loss[0] : 1.0f;
  for (int g = threadIdx.x; g < gs; g += blockDim.x) {
    float v = cf[g] * up / c.tau;
    if (c.score_clip > 0.f && fabsf(s[g]) >= c.score_clip) v = 0.f;
    coef[(long long)b * gs + g] = v;
    if (d_user_bias_rows) d_user_bias_rows[(long long)b * gs + g] = v;
  }
}
// d_user[r,:] = coef[r] * E[item_id[r],:]  (one 32-lane group per row)
__global__ __launch_bounds__(256) void coef_rows_kernel(const float* __restrict__ coef, const float4* __restrict__ table,
                                                        const long long* __restrict__ item_id, float4* __restrict__ d_user, int rows, int d4,
                                                        long long n_items) {
  const int r = blockIdx.x * 8 + (threadIdx.x >> 5), t = threadIdx.x & 31;
  if (r >= rows) return;
  const long long id = UR_ROW(item_id[r], n_items);
  const float w = coef[r];
  for (int col = t; col < d4; col += 32) {
    const float4 e = table[id * d4 + col];
    d_user[(long long)r * d4 + col] = make_float4(w * e.x, w * e.y, w * e.z, w * e.w);
  }
}

static inline int pick_tpr(int d) {
  int d4 = d / 4, t = 4;
  while (t < d4 && t < 32) t <<= 1;
  return t;
}

}  // namespace ur

using namespace ur;

static int check_loss_cfg(const UrLossCfg* c, const char* who, bool fwd = false) {
  UR_REQUIRE(c != nullptr, UR_ERR_ARG, "%s: null cfg", who);
  UR_REQUIRE(c->B > 0 && c->G > 0, UR_ERR_ARG, "%s: B=%d G=%d", who, c->B, c->G);
  UR_REQUIRE(c->d > 0 && c->d % 4 == 0 && c->d <= 512, UR_ERR_ARG, "%s: d=%d must be a multiple of 4, <= 512", who, c->d);
  UR_REQUIRE(c->G <= 8192, UR_ERR_UNSUPPORTED, "%s: group size G=%d > 8192", who, c->G);
  UR_REQUIRE(c->loss_type >= (fwd ? UR_LOSS_NONE : UR_LOSS_BCE) && c->loss_type <= UR_LOSS_CCL, UR_ERR_UNSUPPORTED,
             "%s: loss_type=%d is not a sampled loss (fullsoftmax scores all N items: not implemented)", who, c->loss_type);
  UR_REQUIRE(c->group_size >= 0 && c->group_size <= 8192, UR_ERR_ARG, "%s: group_size=%d", who, c->group_size);
  if (c->group_size > 0) {   // user-item-label rows: one candidate per row, group_size consecutive rows = one score row of the loss
    UR_REQUIRE(c->G == 1 && c->B % c->group_size == 0, UR_ERR_ARG, "%s: group_size=%d needs G == 1 and B %% group_size == 0 (B=%d G=%d)", who,
               c->group_size, c->B, c->G);
    UR_REQUIRE(c->loss_type != UR_LOSS_NONE, UR_ERR_ARG, "%s: group_size with loss_type none", who);
  }
  UR_REQUIRE((c->loss_type != UR_LOSS_BPR && c->loss_type != UR_LOSS_CCL) || (c->group_size > 0 ? c->group_size : c->G) >= 2, UR_ERR_ARG,
             "%s: pairwise loss needs G >= 2", who);
  UR_REQUIRE(c->tau != 0.f, UR_ERR_ARG, "%s: tau == 0", who);
  return UR_OK;
}

extern "C" int ur_gather_dot_loss_fwd(const UrLossCfg* cfg, const float* user_emb, const float* item_table, int64_t n_items,
                                      const int64_t* item_id, const int32_t* label, const float* user_bias,
                                      const float* item_bias, const int64_t* user_id, float* scores, float* loss_rows,
                                      float* loss_out, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_loss_cfg(cfg, "ur_gather_dot_loss_fwd", true);
  if (rc) return rc;
  UR_REQUIRE(user_emb && item_table && item_id && scores && loss_rows && loss_out, UR_ERR_ARG, "ur_gather_dot_loss_fwd: null pointer");
  UR_REQUIRE(n_items > 0, UR_ERR_ARG, "ur_gather_dot_loss_fwd: n_items");
  UR_REQUIRE(label || (cfg->loss_type != UR_LOSS_BCE && cfg->loss_type != UR_LOSS_SOFTMAX), UR_ERR_ARG,
             "ur_gather_dot_loss_fwd: label is required for bce/softmax");
  UR_REQUIRE(!user_bias || user_id, UR_ERR_ARG, "ur_gather_dot_loss_fwd: user_bias needs user_id");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_LOSS, st, (double)cfg->B * cfg->G * cfg->d * 4.0);
  const int gs = cfg->group_size, loss_type_in = cfg->loss_type;
  UrLossCfg rows_cfg = *cfg;
  if (gs > 0) {   // group mode: the scorer alone over the rows (G = 1), the loss on the regrouped scores below
    rows_cfg.group_size = 0;
    rows_cfg.loss_type = UR_LOSS_NONE;
    cfg = &rows_cfg;
  }
  const int tpr = pick_tpr(cfg->d);
  const size_t lds = (cfg->G + 16) * sizeof(float);
  const bool wide = cfg->G >= 512 && cfg->B <= 512;   // few rows, many candidates: 1024-thread workgroups
  float* cnt_rows = loss_rows + cfg->B;  // loss_rows buffer is [2*B]: losses then counts
  const int kv = cdiv(cfg->d / 4, tpr);   // float4 chunks per lane (1 for d <= 128)
#define GO3(T, U, NT, KV_) hipLaunchKernelGGL((scorer_loss_fwd_kernel<T, U, NT, KV_>), dim3(cfg->B), dim3(NT), lds, st, *cfg, (const float4*)user_emb, \
                                 (const float4*)item_table, (const long long*)item_id, label, user_bias, item_bias,            \
                                 (const long long*)user_id, scores, loss_rows, cnt_rows, (long long)n_items)
#define GO2(T, U, NT) do { if (kv <= 1) GO3(T, U, NT, 1); else if (kv == 2) GO3(T, U, NT, 2); else GO3(T, U, NT, 4); } while (0)
#define GO1(T, U) do { if (wide) GO2(T, U, 1024); else GO2(T, U, 256); } while (0)
#define GO(T) GO1(T, 8)   /* 8 candidate rows in flight per lane group (2 / 4 / 16 measured slower: round 1) */
  switch (tpr) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    default: GO(32); break;
  }
#undef GO
#undef GO1
#undef GO2
#undef GO3
  UR_LAUNCH_CHECK();
  int n_loss_rows = cfg->B;
  if (gs > 0) {
    UrLossCfg gc = *cfg;
    gc.group_size = gs;
    gc.loss_type = loss_type_in;
    n_loss_rows = cfg->B / gs;
    hipLaunchKernelGGL(group_loss_fwd_kernel, dim3(n_loss_rows), dim3(256), (gs + 16) * sizeof(float), st, gc, scores, label, loss_rows, cnt_rows);
    UR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, loss_rows, cnt_rows, n_loss_rows, loss_out);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_gather_dot_loss_bwd(const UrLossCfg* cfg, const float* user_emb, const float* item_table, int64_t n_items,
                                      const int64_t* item_id, const int32_t* label, const float* scores,
                                      const float* loss_out, const float* d_loss, float* coef, float* d_user,
                                      float* d_user_bias_rows, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_loss_cfg(cfg, "ur_gather_dot_loss_bwd");
  if (rc) return rc;
  UR_REQUIRE(user_emb && item_table && item_id && scores && loss_out && coef && d_user, UR_ERR_ARG,
             "ur_gather_dot_loss_bwd: null pointer");
  UR_REQUIRE(label || (cfg->loss_type != UR_LOSS_BCE && cfg->loss_type != UR_LOSS_SOFTMAX), UR_ERR_ARG,
             "ur_gather_dot_loss_bwd: label is required for bce/softmax");
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_LOSS, st, (double)cfg->B * cfg->G * cfg->d * 4.0);
  if (cfg->group_size > 0) {   // group mode: coefficients from the regrouped scores, then d_user[r] = coef[r] * E[item_id[r]]
    const int gs = cfg->group_size;
    hipLaunchKernelGGL(group_loss_bwd_kernel, dim3(cfg->B / gs), dim3(256), (gs + 16) * sizeof(float), st, *cfg, scores, label, d_loss, loss_out,
                       coef, d_user_bias_rows);
    UR_LAUNCH_CHECK();
    hipLaunchKernelGGL(coef_rows_kernel, dim3(cdiv(cfg->B, 8)), dim3(256), 0, st, coef, (const float4*)item_table, (const long long*)item_id,
                       (float4*)d_user, cfg->B, cfg->d / 4, (long long)n_items);
    UR_LAUNCH_CHECK();
    return UR_OK;
  }
  const int tpr = pick_tpr(cfg->d);
  bool wide = cfg->G >= 512 && cfg->B <= 512;   // few rows, many candidates: 1024-thread workgroups
  if (wide && ((size_t)cfg->G + (size_t)(1024 / tpr) * cfg->d + 16) * sizeof(float) > 64 * 1024) wide = false;
  const int groups = (wide ? 1024 : 256) / tpr;
  const size_t lds = ((size_t)cfg->G + (size_t)groups * cfg->d + 16) * sizeof(float);
  UR_REQUIRE(lds <= 64 * 1024, UR_ERR_UNSUPPORTED, "ur_gather_dot_loss_bwd: LDS need %zu bytes", lds);
  const int kv = cdiv(cfg->d / 4, tpr);
#define GO(T) do { if (wide) GOB(T, 1024); else GOB(T, 256); } while (0)
#define GOK(T, NT, KV_) hipLaunchKernelGGL((scorer_loss_bwd_kernel<T, NT, KV_>), dim3(cfg->B), dim3(NT), lds, st, *cfg, (const float4*)user_emb, \
                                 (const float4*)item_table, (const long long*)item_id, label, scores, d_loss, loss_out, coef,  \
                                 (float4*)d_user, d_user_bias_rows, (long long)n_items)
#define GOB(T, NT) do { if (kv <= 1) GOK(T, NT, 1); else if (kv == 2) GOK(T, NT, 2); else GOK(T, NT, 4); } while (0)
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

// One counter per (device, stream) for the "last workgroup finishes the loss" step of the fused kernel (zeroed once; the kernel resets
// it): launches on different streams never share one, launches on one stream are ordered.  (The hand-off itself uses relaxed
// device-scope read-modify-writes, not release / acquire: a device-scope release on this part writes back the XCD's whole L2 -- 512
// of them cost the step 20 us -- and an RMW is performed at the point all XCDs agree on; see scorer_loss_fused_kernel.)
static unsigned* fused_counter(hipStream_t st) {
  struct Slot { hipStream_t st; unsigned* p; };
  static Slot z[64][16] = {};
  static int used[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  for (int i = 0; i < used[dev]; ++i)
    if (z[dev][i].st == st) return z[dev][i].p;
  if (used[dev] == 16) {   // more than 16 training streams on a device: the last counter changes hands (it is zero between launches)
    z[dev][15].st = st;
    return z[dev][15].p;
  }
  unsigned* p = nullptr;
  if (hipMalloc((void**)&p, 256) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
  z[dev][used[dev]++] = Slot{st, p};
  return p;
}

extern "C" int ur_gather_dot_loss_fused_supported(const UrLossCfg* cfg) {
  if (!cfg || cfg->group_size > 0) return 0;
  if (!(cfg->loss_type == UR_LOSS_BPR || cfg->loss_type == UR_LOSS_BCE || cfg->loss_type == UR_LOSS_CCL)) return 0;
  return ((size_t)cfg->G * cfg->d + 2 * (size_t)cfg->G + 16) * sizeof(float) <= 32 * 1024 && cfg->d % 4 == 0 && cfg->d <= 512;
}

extern "C" int ur_gather_dot_loss_fwd_bwd(const UrLossCfg* cfg, const float* user_emb, const float* item_table, int64_t n_items,
                                          const int64_t* item_id, const int32_t* label, const float* user_bias,
                                          const float* item_bias, const int64_t* user_id, float* scores, float* loss_rows,
                                          float* loss_out, float* coef, float* d_user, float* d_user_bias_rows, void* stream) {
  UR_TRACE_SCOPE();
  int rc = check_loss_cfg(cfg, "ur_gather_dot_loss_fwd_bwd");
  if (rc) return rc;
  UR_REQUIRE(user_emb && item_table && item_id && scores && loss_rows && loss_out && coef && d_user, UR_ERR_ARG,
             "ur_gather_dot_loss_fwd_bwd: null pointer");
  UR_REQUIRE(n_items > 0, UR_ERR_ARG, "ur_gather_dot_loss_fwd_bwd: n_items");
  UR_REQUIRE(label || cfg->loss_type != UR_LOSS_BCE, UR_ERR_ARG, "ur_gather_dot_loss_fwd_bwd: label is required for bce");
  UR_REQUIRE(!user_bias || user_id, UR_ERR_ARG, "ur_gather_dot_loss_fwd_bwd: user_bias needs user_id");
  UR_REQUIRE(ur_gather_dot_loss_fused_supported(cfg), UR_ERR_UNSUPPORTED,
             "ur_gather_dot_loss_fwd_bwd: loss_type=%d G=%d d=%d (bpr / bce / ccl with G * d * 4 <= 32 KB; otherwise call _fwd and _bwd)",
             cfg->loss_type, cfg->G, cfg->d);
  hipStream_t st = as_stream(stream);
  unsigned* counter = fused_counter(st);
  UR_REQUIRE(counter != nullptr, UR_ERR_HIP, "ur_gather_dot_loss_fwd_bwd: no device memory for the completion counter");
  ProfScope ps(PC_LOSS, st, (double)cfg->B * cfg->G * cfg->d * 4.0);
  const int tpr = pick_tpr(cfg->d);
  const size_t lds = ((size_t)cfg->G * cfg->d + 2 * (size_t)cfg->G + 16) * sizeof(float);
  const float total = cfg->loss_type == UR_LOSS_BCE ? (float)cfg->B * (float)cfg->G : (float)cfg->B;
  float* cnt_rows = loss_rows + cfg->B;
#define GO(T) hipLaunchKernelGGL((scorer_loss_fused_kernel<T>), dim3(cfg->B), dim3(256), lds, st, *cfg, total, (const float4*)user_emb, \
                                 (const float4*)item_table, (const long long*)item_id, label, user_bias, item_bias,                 \
                                 (const long long*)user_id, scores, loss_rows, cnt_rows, coef, d_user, d_user_bias_rows, loss_out, counter, ur_arrive_mode(), (long long)n_items)
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
