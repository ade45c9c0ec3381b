// Shared device/host helpers for the unirec_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/unirec_amd.h"

namespace ur {

constexpr int kWave = 64;  // CDNA wavefront width

// ---- error plumbing (thread-local message read back through ur_last_error) ----------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define UR_REQUIRE(cond, code, ...)              \
  do {                                           \
    if (!(cond)) return ::ur::fail((code), __VA_ARGS__); \
  } while (0)

#define UR_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::ur::fail(UR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define UR_LAUNCH_CHECK()                                                                    \
  do {                                                                                       \
    hipError_t _e = hipGetLastError();                                                       \
    if (_e != hipSuccess)                                                                    \
      return ::ur::fail(UR_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// ---- a launch that carries its own completion event.  hipEventRecord puts a marker packet behind the kernel and the NEXT kernel of the
// stream waits for that packet: ~5 us of idle stream at every fork of the backward pass.  hipExtLaunchKernelGGL binds the event to the
// kernel's own completion signal instead: nothing is inserted.  A caller arms `g_stop_event` right before calling a helper whose LAST
// launch goes through UR_LAUNCH_EV; that launch consumes it (other launches of the helper leave it alone).
// Context id of the calling thread: 0 = the process's default context; ur_loop_attach (exchange.hip: the in-process loopback transport, W
// rank threads in one process) sets rank + 1.  Per-context state -- the encoder's side stream and its events (sasrec.hip), the split
// kernels' hand-off counters (rowchain.hip) -- is indexed by it, so that rank threads sharing a process do not share it.
constexpr int UR_MAX_CTX = 65;
extern thread_local int g_ctx_id;
extern thread_local hipEvent_t g_stop_event;
// The profiler's kernel-bound brackets (ProfScope with kernel_events): the next UR_LAUNCH_EV launch carries the scope's two events as ITS
// start / completion timestamps -- the dispatch's own begin and end, not the stream's: an event RECORDED in front of a kernel is stamped
// when the stream gets there, so a cross-stream wait queued between the record and the kernel (the forward pass's late join of the side
// stream, a fork's wait) was counted as kernel time (round 3: the row-chain class read 361 us / step where the kernel trace said 259).
extern thread_local hipEvent_t g_prof_start, g_prof_stop;
#define UR_LAUNCH_EV(kernel, grid, block, lds, st, ...)                                        \
  do {                                                                                       \
    if (::ur::g_prof_start) {                                                                \
      hipEvent_t ea_ = ::ur::g_prof_start, eb_ = ::ur::g_prof_stop;                          \
      ::ur::g_prof_start = nullptr; ::ur::g_prof_stop = nullptr;                             \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), st, ea_, eb_, 0, __VA_ARGS__); \
    } else if (::ur::g_stop_event) {                                                         \
      hipEvent_t ev_ = ::ur::g_stop_event;                                                   \
      ::ur::g_stop_event = nullptr;                                                          \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), st, nullptr, ev_, 0, __VA_ARGS__); \
    } else {                                                                                 \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                         \
    }                                                                                        \
  } while (0)

// ---- live kernel-class profiler (HIP events recorded on the launch stream; see ur_prof_* in the header)
enum ProfClass { PC_GEMM_NT = 0, PC_GEMM_TN, PC_ATTN_FWD, PC_ATTN_BWD, PC_ROWOPS, PC_LOSS, PC_SORT, PC_REDUCE, PC_ADAM, PC_GATHER,
                 PC_GRU, PC_CHAIN, PC_CHAIN_SMALL, PC_MISC,
                 PC_A2A_IDS, PC_A2A_ROWS, PC_A2A_GRADS, PC_ALLREDUCE,   // the library's RCCL collectives (exchange.hip): stream time of the group, work = bytes sent to peers
                 PC_COUNT };
bool prof_brackets(int cls);   // launches of this class are being bracketed with events right now
struct ProfScope {
  int cls; hipStream_t st; int slot; bool kernel_events;
  // kernel_events: the scope holds exactly ONE launch and that launch goes through UR_LAUNCH_EV -- its own start / end timestamps are the bracket
  ProfScope(int cls_, hipStream_t st_, double work = 0.0, bool kernel_events_ = false);
  ~ProfScope();
};

// ---- tracing hooks ("roctx ranges per op"; the reference wraps a run in cProfile, unirec/main/main.py:490-499).  UR_ROCTX=1:
// every C-ABI entry that enqueues device work pushes a roctx range named after itself for the duration of the call (roctxRangePushA /
// roctxRangePop, resolved at run time from libroctx64.so): a `rocprofv3 --marker-trace --kernel-trace` timeline then shows which entry
// point each launch belongs to.  Off (the default): one predictable branch per call.
struct TraceScope {
  bool on;
  explicit TraceScope(const char* name);
  ~TraceScope();
};
#define UR_TRACE_SCOPE() ::ur::TraceScope ur_trace_scope_(__func__)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -----------------------------------------------------------------------
// Reduction across WIDTH consecutive lanes (WIDTH a power of two <= 64), result in EVERY lane of the group.  Inside a row of 16
// lanes the exchange is DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: VALU-rate, no LDS crossbar); only the steps that
// cross rows (16, 32) go through ds_bpermute.  A LayerNorm row needs two such reductions: 2 x 5 bpermutes -> 2 x 1.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (WIDTH >= 2) v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (WIDTH >= 4) v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (WIDTH >= 8) v += dpp_move<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of its 8
  if constexpr (WIDTH >= 16) v += dpp_move<0x140>(v);  // row_mirror:      lane i <-> 15 - i of its 16
  if constexpr (WIDTH >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (WIDTH >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
  if constexpr (WIDTH >= 2) v = fmaxf(v, dpp_move<0xB1>(v));
  if constexpr (WIDTH >= 4) v = fmaxf(v, dpp_move<0x4E>(v));
  if constexpr (WIDTH >= 8) v = fmaxf(v, dpp_move<0x141>(v));
  if constexpr (WIDTH >= 16) v = fmaxf(v, dpp_move<0x140>(v));
  if constexpr (WIDTH >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  if constexpr (WIDTH >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
// ---- test hooks.  ONE environment variable, read once per process: UR_TEST="name[=value],name[=value],..." -- the kernel families that
// are not the default at the benchmark shapes (but ARE the path of other shapes) forced on at test shapes, and timing skews:
//   attn_no_mfma / attn_no_m16 / attn_no_m16t / attn_no_m16w   attention kernel families (attention.hip)
//   gru_no_seq / gru_no_step                                   GRU recurrence paths (gru.hip)
//   plan_multi                                                 the multi-launch id sort at every batch size (rows.hip)
//   chain_mask=<bits>                                          which row-chain kernels are on (rowchain.hip; all on by default)
//   topk_cap=<n>  arrival_skew_us=<n>  side_delay_us=<n>       list overflows, hand-off skew, a late side stream
// ur_test_hook(name): the hook's value (1 when listed without a value), `absent` when not listed.  The SUPPORTED switches are separate,
// documented environment variables (README.md): UR_SASREC_SIDE, UR_COMM_SINGLE and the Python-level ones.
int ur_test_hook(const char* name, int absent = 0);

// ---- arrival at a cross-workgroup hand-off ("whoever arrives last finishes the job").  The counter increment is release-acquire at
// agent scope, the form the HIP memory model recognises -- buffer_wbl2 sc1 in front of the RMW (the XCD's L2 written back, dirty lines
// of every other kernel included), buffer_inv sc1 behind it -- by ONE thread per workgroup: +2 us on the 0.55 ms step (round 4,
// profiles/r04_c_strict_order.txt; rounds 2-3 had priced a __threadfence() by all 256 threads at +20 us and kept a relaxed counter).
// The DATA still travels in device-scope RMWs (no reader can hit a stale L2 line whatever the order of the arrivals).
// mode >> 1 (test hook arrival_skew_us): the arriving thread first waits (hash of the workgroup) % skew microseconds, so that the
// last arriver moves over all XCDs (tests/test_handoff_order_gpu.py).
static inline int ur_arrive_mode() {   // read once per process
  static const int mode = 1 | (ur_test_hook("arrival_skew_us") << 1);
  return mode;
}
__device__ __forceinline__ unsigned ur_arrive(unsigned* cnt, int mode) {
  const int skew = mode >> 1;
  if (skew > 0) {
    unsigned h = (blockIdx.x + 1u) * 2654435761u;
    h ^= h >> 15;
    const long long t0 = wall_clock64(), wait = (long long)(h % (unsigned)skew) * 100;   // wall_clock64: 100 MHz
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
  return __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// ---- id guard (round 5; what nn.Embedding's range check does at unirec/model/base/reco_abc.py:170 -- an id >= n_items or < 0 raises
// IndexError there).  Every training batch's ids pass through the row plan ONCE (rows.hip: the first pass of the sort): an id outside
// [0, n_rows) raises the device's guard word there, is recorded (first offender wins) and is treated as the padding id 0 from then on
// -- so no kernel that WRITES through a plan (row reduce, sparse update, exchange) can touch memory outside a table.  The guard word is
// sticky: every update kernel reads it next to its gradient scale and skips the step like a NaN step while it is raised (no extra
// launch, no host round trip).  The plan kernel also stores the verdict into a HOST-mapped mirror, which the host polls with a plain
// load at the head of every step (ops.id_guard_check): IndexError with the offending id one or two steps later, tables untouched since.
// dev[0] = raised, dev[1..2] = offending id (lo, hi), dev[3] = table rows (saturated to int); host: the same four words.
struct IdGuard { int* dev; int* host; };   // host: the device-visible address of the host-mapped mirror
IdGuard id_guard();                         // this device's guard (allocated on first use; {nullptr, nullptr} if that failed)
__device__ __forceinline__ long long ur_guard_id(long long id, long long n_rows, IdGuard gd) {
  if ((unsigned long long)id < (unsigned long long)n_rows) return id;
  if (gd.dev && atomicOr(gd.dev, 1) == 0) {   // first offender of this device: record it, publish to the host mirror
    gd.dev[1] = (int)(id & 0xFFFFFFFFll); gd.dev[2] = (int)(id >> 32); gd.dev[3] = (int)(n_rows > 0x7FFFFFFFll ? 0x7FFFFFFFll : n_rows);
    if (gd.host) {
      gd.host[1] = gd.dev[1]; gd.host[2] = gd.dev[2]; gd.host[3] = gd.dev[3];
      __hip_atomic_store(gd.host, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  return 0;
}
// gradient scale of an update kernel: the caller's device scalar (< 0 = skip the step: NaN loss, trainer.py:343-350), -1 while the
// device's id guard is raised
__device__ __forceinline__ float ur_step_scale(const float* scale_dev, const int* guard_dev) {
  float s = scale_dev ? *scale_dev : 1.0f;
  if (guard_dev && *guard_dev) s = -1.0f;
  return s;
}
// ---- bounds-checked build (python -m unirec_amd.build --debug-bounds -> libunirec_amd_dbg.so, loaded when UR_DEBUG_BOUNDS=1): every row
// index a kernel gathers from or scatters to is checked where it is used; a bad one prints the site and traps (the launch fails loudly,
// like a device-side assert).  The release build clamps the index to the padding row instead (below).
#ifdef UR_DEBUG_BOUNDS
__device__ __forceinline__ long long ur_dbg_row(long long id, long long n, const char* file, int line) {
  if (n > 0 && (unsigned long long)id >= (unsigned long long)n) {   // (n <= 0: the caller did not say how many rows the table has)
    printf("unirec_amd bounds check: row index %lld outside [0, %lld) at %s:%d\n", id, n, file, line);
    __builtin_trap();
  }
  return id;
}
#define UR_ROW(id, n) ::ur::ur_dbg_row((long long)(id), (long long)(n), __FILE__, __LINE__)
#else
// release build: an index outside the table reads the padding row 0 instead (one compare + select per id; a gather at row -1 of a table
// that starts its allocation is a page fault, i.e. an aborted process, long before the host gets to raise the guard's IndexError)
__device__ __forceinline__ long long ur_clamp_row(long long id, long long n) {
  return (n > 0 && (unsigned long long)id >= (unsigned long long)n) ? 0 : id;
}
#define UR_ROW(id, n) ::ur::ur_clamp_row((long long)(id), (long long)(n))
#endif


// runtime-width variant (width power of two)
__device__ __forceinline__ float group_sum_rt(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- optimizer update rules (torch.optim semantics as constructed at unirec/facility/trainer.py:134-152: only lr and
// weight_decay are passed, everything else is torch's default).  State: m, v (Adam / AdamW), v = state_sum (Adagrad),
// v = square_avg (RMSprop, alpha carried in b2), none (SGD).
struct AdamK {
  float lr, b1, b2, eps, wd;
  int step, algo;
  float lb1 = 0.f, lb2 = 0.f;   // log2(b1), log2(b2) (host, double precision): b^j = exp2(j * lb) in the lazy-replay series (rows.hip)
  int background = 0;           // the launch runs UNDER a step's compute (plan stream): default wave priority, a small grid
};
// one element, gradient gr (already scaled / clipped); bc1 = 1 - b1^t, bc2s = sqrt(1 - b2^t)
// (no fp contraction: whether a product is fused into the following add would otherwise be decided per call site, and the same rule
// inlined into two kernels -- the row update and the reduce kernel's fused epilogue -- could differ in the last bit)
__device__ __forceinline__ void opt_elem(float& w, float& m, float& v, float gr, const AdamK& a, float bc1, float bc2s) {
#pragma clang fp contract(off)
  switch (a.algo) {
    case UR_OPT_ADAMW:      // decoupled decay, then Adam on the raw gradient
      w *= 1.f - a.lr * a.wd;
      m = a.b1 * m + (1.f - a.b1) * gr;
      v = a.b2 * v + (1.f - a.b2) * gr * gr;
      w -= (a.lr / bc1) * (m / (sqrtf(v) / bc2s + a.eps));
      break;
    case UR_OPT_SGD:        // momentum 0
      gr += a.wd * w;
      w -= a.lr * gr;
      break;
    case UR_OPT_ADAGRAD:    // lr_decay 0, initial accumulator 0
      gr += a.wd * w;
      v += gr * gr;
      w -= a.lr * (gr / (sqrtf(v) + a.eps));
      break;
    case UR_OPT_RMSPROP:    // momentum 0, not centered; alpha = b2
      gr += a.wd * w;
      v = a.b2 * v + (1.f - a.b2) * gr * gr;
      w -= a.lr * (gr / (sqrtf(v) + a.eps));
      break;
    default:                // Adam: L2 folded into the gradient
      gr += a.wd * w;
      m = a.b1 * m + (1.f - a.b1) * gr;
      v = a.b2 * v + (1.f - a.b2) * gr * gr;
      w -= (a.lr / bc1) * (m / (sqrtf(v) / bc2s + a.eps));
      break;
  }
}

// ---- dropout (training only).  Inverted dropout on a grid of elements (row id, column): the keep decision is a pure
// function of (stream key, row id, column) -- nothing is stored, the backward re-evaluates it -- built from the 32-bit
// integer finaliser `mix32` (xorshift-multiply; the "lowbias32" constants):
//   rowkey = mix32(row_id ^ key)      keep <=> mix32(rowkey + column * 0x9E3779B9) >= thresh        value *= keep ? 1/(1-p) : 0
// key = drop_stream_key(seed, step, site) is one per (seed, training step, dropout site); thresh = floor(p * 2^32).
// The random source is this hash, not torch's Philox stream: parity is against oracle/dropout_ref.py, which restates it.
struct DropSpec {
  unsigned key, thresh;   // thresh == 0: dropout off
  float scale;            // 1 / (1 - p)
  const int* rows;        // row id of buffer row r: rows[r] when non-null, else r * mul + add
  int mul, add;
};
__host__ __device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
static inline unsigned drop_stream_key(long long seed, long long step, unsigned site) {
  unsigned k = mix32((unsigned)seed ^ 0x85ebca6bu);
  k = mix32(k ^ (unsigned)((unsigned long long)seed >> 32));
  k = mix32(k ^ (unsigned)step);
  k = mix32(k ^ (unsigned)((unsigned long long)step >> 32));
  return mix32(k + site * 0x9E3779B9u);
}
// p in [0,1): the spec of one site (row map filled in by the caller)
static inline DropSpec drop_spec(float p, long long seed, long long step, unsigned site) {
  DropSpec s{};
  if (p > 0.f) {
    double t = (double)p * 4294967296.0;
    s.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    s.scale = 1.0f / (1.0f - p);
    s.key = drop_stream_key(seed, step, site);
  }
  s.mul = 1;
  return s;
}
__device__ __forceinline__ unsigned drop_rowkey(const DropSpec& s, int r) {
  const int id = s.rows ? s.rows[r] : r * s.mul + s.add;
  return mix32((unsigned)id ^ s.key);
}
__device__ __forceinline__ float drop_mul(unsigned rowkey, unsigned col, unsigned thresh, float scale) {
  return mix32(rowkey + col * 0x9E3779B9u) >= thresh ? scale : 0.f;
}
__device__ __forceinline__ float4 drop4(float4 v, unsigned rowkey, unsigned col0, const DropSpec& s) {
  v.x *= drop_mul(rowkey, col0, s.thresh, s.scale); v.y *= drop_mul(rowkey, col0 + 1, s.thresh, s.scale);
  v.z *= drop_mul(rowkey, col0 + 2, s.thresh, s.scale); v.w *= drop_mul(rowkey, col0 + 3, s.thresh, s.scale);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// activation ids shared with the host (UR_ACT_*)
__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case UR_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    case UR_ACT_RELU: return fmaxf(x, 0.0f);
    case UR_ACT_SWISH: return __fdividef(x, 1.0f + __expf(-x));   // v_exp_f32 + v_rcp_f32: ~1e-7 relative
    case UR_ACT_TANH: return tanhf(x);
    case UR_ACT_SIGMOID: return __fdividef(1.0f, 1.0f + __expf(-x));
    default: return x;
  }
}
// derivative d act(x) / dx evaluated at the pre-activation x
__device__ __forceinline__ float act_bwd(float x, int act) {
  switch (act) {
    case UR_ACT_GELU: {
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case UR_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
    case UR_ACT_SWISH: {
      const float s = __fdividef(1.0f, 1.0f + __expf(-x));
      return s * (1.0f + x * (1.0f - s));
    }
    case UR_ACT_TANH: {
      const float t = tanhf(x);
      return 1.0f - t * t;
    }
    case UR_ACT_SIGMOID: {
      const float s = 1.0f / (1.0f + expf(-x));
      return s * (1.0f - s);
    }
    default: return 1.0f;
  }
}

}  // namespace ur
