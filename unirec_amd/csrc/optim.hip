// Dense Adam over the flat dense-parameter buffer, gradient-norm helpers, and the library's error plumbing.
#include <stdarg.h>

#include <atomic>
#include <dlfcn.h>
#include <string>

#include "common.h"
#include "kernels.h"

namespace ur {
thread_local int g_ctx_id = 0;                    // common.h: context id of the calling thread
thread_local hipEvent_t g_stop_event = nullptr;   // common.h: UR_LAUNCH_EV
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;   // common.h: ProfScope with kernel_events
}
namespace ur {

static thread_local char g_err[512] = "";
static int* g_guard_host[64] = {};   // host addresses of the id guard's mirrors (id_guard below)

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// torch.optim single-tensor update of the flat dense-parameter buffer (rule: opt_elem in common.h)
__global__ __launch_bounds__(256) void dense_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, AdamK a, float bc1, float bc2s,
                                                         const float* __restrict__ scale_dev, const int* __restrict__ guard_dev) {
  const float scale = ur_step_scale(scale_dev, guard_dev);
  if (scale < 0.f) return;   // update guard: the step's loss was NaN (Trainer skips such a step, trainer.py:343-350), or the id guard is raised
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float w = p[i], mi = m[i], vi = v[i];
    opt_elem(w, mi, vi, g[i] * scale, a, bc1, bc2s);
    m[i] = mi;
    v[i] = vi;
    p[i] = w;
  }
}

// stage 1: part[blk] = sum over a contiguous slice; stage 2: single block sums the parts in order
__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long b = (long long)blockIdx.x * per, e = b + per < n ? b + per : n;
  float s = 0.f;
  for (long long i = b + threadIdx.x; i < e; i += blockDim.x) s = fmaf(x[i], x[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_stage2(const float* __restrict__ part, int nparts, float* __restrict__ out, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = accumulate ? out[0] + t : t;
  }
}
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ out,
                                 const float* __restrict__ guard) {
  const float c = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
  out[0] = (guard && guard[0] < 0.f) ? -1.f : (c < 1.f ? c : 1.f);
}

// ---- roctx ranges (common.h: TraceScope)
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool on = false;
};
const Roctx& roctx() {
  static const Roctx r = [] {
    Roctx x;
    const char* e = getenv("UR_ROCTX");
    if (!e || atoi(e) == 0) return x;
    void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return x;
    x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    x.on = x.push && x.pop;
    return x;
  }();
  return r;
}
}  // namespace
std::atomic<long long> g_ranges_pushed{0};
TraceScope::TraceScope(const char* name) : on(roctx().on) {
  if (on) {
    (void)roctx().push(name);
    g_ranges_pushed.fetch_add(1, std::memory_order_relaxed);
  }
}
TraceScope::~TraceScope() {
  if (on) (void)roctx().pop();
}

// ---- the test hooks (common.h): UR_TEST parsed once
int ur_test_hook(const char* name, int absent) {
  static const std::string spec = [] { const char* e = getenv("UR_TEST"); return std::string(e ? e : ""); }();
  const size_t n = strlen(name);
  size_t pos = 0;
  while (pos <= spec.size()) {
    size_t end = spec.find(',', pos);
    if (end == std::string::npos) end = spec.size();
    if (end - pos >= n && spec.compare(pos, n, name) == 0 && (pos + n == end || spec[pos + n] == '=')) {
      return pos + n == end ? 1 : atoi(spec.c_str() + pos + n + 1);
    }
    pos = end + 1;
  }
  return absent;
}

// ---- the id guard's storage (common.h): per device, allocated once
IdGuard id_guard() {
  static IdGuard g[64] = {};
  static bool tried[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return IdGuard{nullptr, nullptr};
  if (!tried[dev]) {
    tried[dev] = true;
    int* d = nullptr;
    int* h = nullptr;
    if (hipMalloc((void**)&d, 16) == hipSuccess && hipMemset(d, 0, 16) == hipSuccess &&
        hipHostMalloc((void**)&h, 16, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      memset(h, 0, 16);
      void* hd = nullptr;
      if (hipHostGetDevicePointer(&hd, h, 0) == hipSuccess) g[dev] = IdGuard{d, (int*)hd};
      g_guard_host[dev] = h;
      (void)hipDeviceSynchronize();
    }
  }
  return g[dev];
}

}  // namespace ur

using namespace ur;

// the host's view of this device's id guard, no synchronisation (a plain load of the host-mapped mirror): 0 = clear; 1 = raised, with
// out3 = {offending id, rows of the table it was aimed at (saturated to 2^31 - 1), 0}
extern "C" int ur_id_guard_state(int64_t* out3) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  (void)id_guard();
  volatile int* h = g_guard_host[dev];
  if (!h || h[0] == 0) return 0;
  if (out3) {
    out3[0] = (int64_t)(((uint64_t)(uint32_t)h[2] << 32) | (uint32_t)h[1]);
    out3[1] = h[3];
    out3[2] = 0;
  }
  return 1;
}
// clear the guard (after the IndexError has been handled): device word on `stream`, host mirror at once
extern "C" int ur_id_guard_reset(void* stream) {
  UR_TRACE_SCOPE();
  int dev = 0;
  UR_HIP(hipGetDevice(&dev));
  IdGuard g = id_guard();
  UR_REQUIRE(g.dev && dev >= 0 && dev < 64 && g_guard_host[dev], UR_ERR_HIP, "ur_id_guard_reset: no guard on this device");
  UR_HIP(hipMemsetAsync(g.dev, 0, 16, as_stream(stream)));
  UR_HIP(hipStreamSynchronize(as_stream(stream)));
  memset(g_guard_host[dev], 0, 16);
  return UR_OK;
}

extern "C" int64_t ur_trace_ranges_pushed(void) { return g_ranges_pushed.load(std::memory_order_relaxed); }
extern "C" const char* ur_last_error(void) { return g_err; }
extern "C" int ur_version(void) { return 100; }

extern "C" int ur_dense_adam(const UrAdamCfg* cfg, float* param, const float* grad, float* m, float* v, int64_t n,
                             const float* grad_scale_dev, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(cfg && param && grad && m && v, UR_ERR_ARG, "ur_dense_adam: null pointer");
  UR_REQUIRE(cfg->step >= 1 && n >= 0, UR_ERR_ARG, "ur_dense_adam: step=%d n=%lld", cfg->step, (long long)n);
  UR_REQUIRE(cfg->algo >= UR_OPT_ADAM && cfg->algo <= UR_OPT_RMSPROP, UR_ERR_ARG, "ur_dense_adam: algo=%d", cfg->algo);
  if (n == 0) return UR_OK;
  const float bc1 = 1.f - powf(cfg->beta1, (float)cfg->step);
  const float bc2s = sqrtf(1.f - powf(cfg->beta2, (float)cfg->step));
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  ProfScope ps(PC_ADAM, as_stream(stream), (double)n * 4.0 * 7);
  hipLaunchKernelGGL(dense_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), param, grad, m, v, (long long)n,
                     AdamK{cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->weight_decay, cfg->step, cfg->algo}, bc1, bc2s, grad_scale_dev, id_guard().dev);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_sumsq(const float* x, int64_t n, float* out, int accumulate, void* ws_2048_floats, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(x && out && ws_2048_floats && n >= 0, UR_ERR_ARG, "ur_sumsq: bad argument");
  hipStream_t st = as_stream(stream);
  int nparts = (int)((n + 4095) / 4096);
  if (nparts > 2048) nparts = 2048;
  if (nparts < 1) nparts = 1;
  hipLaunchKernelGGL(sumsq_stage1, dim3(nparts), dim3(256), 0, st, x, (long long)n, (float*)ws_2048_floats);
  UR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, (const float*)ws_2048_floats, nparts, out, accumulate);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

static int clip_coef_impl(const float* sumsq, float max_norm, float* scale_out, const float* guard, void* stream) {
  UR_REQUIRE(sumsq && scale_out && max_norm > 0.f, UR_ERR_ARG, "ur_clip_coef: bad argument");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, as_stream(stream), sumsq, max_norm, scale_out, guard);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
extern "C" int ur_clip_coef(const float* sumsq, float max_norm, float* scale_out, void* stream) {
  UR_TRACE_SCOPE();
  return clip_coef_impl(sumsq, max_norm, scale_out, nullptr, stream);
}
extern "C" int ur_clip_coef_guarded(const float* sumsq, float max_norm, const float* guard, float* scale_out, void* stream) {
  UR_TRACE_SCOPE();
  return clip_coef_impl(sumsq, max_norm, scale_out, guard, stream);
}

// ------------------------------------------------------------------------------------------------
// Live profiler: when enabled, every internal launcher brackets its launches with a pair of HIP events
// on the launch stream; ur_prof_read sums the elapsed times per kernel class.  Used by bench.py for the
// roofline object (kernel time measured inside the timed region, on the stream the kernel runs on).
#include <vector>
namespace ur {
struct ProfRec { hipEvent_t a, b; int cls; double work; };
static bool g_prof_on = false;
static unsigned g_prof_mask = 0xffffffffu;   // bit c set = kernel class c is bracketed
static std::vector<ProfRec> g_prof;      // recorded pairs since the last reset
static std::vector<ProfRec> g_prof_pool; // reusable events

ProfScope::ProfScope(int cls_, hipStream_t st_, double work, bool kernel_events_) : cls(cls_), st(st_), slot(-1), kernel_events(kernel_events_) {
  if (!g_prof_on || !((g_prof_mask >> cls_) & 1u)) return;
  ProfRec r;
  if (!g_prof_pool.empty()) { r = g_prof_pool.back(); g_prof_pool.pop_back(); }
  else { if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return; }
  r.cls = cls; r.work = work;
  if (kernel_events) { g_prof_start = r.a; g_prof_stop = r.b; }   // bound to the scope's launch (UR_LAUNCH_EV)
  else (void)hipEventRecord(r.a, st);
  slot = (int)g_prof.size();
  g_prof.push_back(r);
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  if (kernel_events) {
    if (g_prof_start == g_prof[slot].a) {   // no launch took them (an early return): an empty bracket
      g_prof_start = nullptr; g_prof_stop = nullptr;
      (void)hipEventRecord(g_prof[slot].a, st);
      (void)hipEventRecord(g_prof[slot].b, st);
    }
    return;
  }
  (void)hipEventRecord(g_prof[slot].b, st);
}
// is a launch of class `cls` being bracketed right now?  (A launch that carries a fork's completion event, UR_LAUNCH_EV, would have the
// event's ~5 us inside the bracket: the forks fall back to hipEventRecord while their producers are being timed.)
bool prof_brackets(int cls) { return g_prof_on && ((g_prof_mask >> cls) & 1u); }
}  // namespace ur

extern "C" int ur_prof_enable(int on) {
  ur::g_prof_on = on != 0;
  return UR_OK;
}
extern "C" int ur_prof_set_mask(uint32_t class_mask) {
  ur::g_prof_mask = class_mask;
  return UR_OK;
}
extern "C" int ur_prof_reset(void) {
  for (auto& r : ur::g_prof) ur::g_prof_pool.push_back(r);
  ur::g_prof.clear();
  return UR_OK;
}
extern "C" int ur_prof_num_classes(void) { return ur::PC_COUNT; }
extern "C" const char* ur_prof_class_name(int cls) {
  static const char* names[] = {"gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "rowops", "scorer_loss", "rows_sort", "rows_reduce",
                                "adam", "gather", "gru", "row_chain", "row_chain_last", "misc", "a2a_ids", "a2a_rows", "a2a_row_grads", "allreduce"};
  return (cls >= 0 && cls < ur::PC_COUNT) ? names[cls] : "?";
}
extern "C" int ur_prof_read(double* host_ms, int64_t* host_count, double* host_work) {
  UR_REQUIRE(host_ms && host_count && host_work, UR_ERR_ARG, "ur_prof_read: null pointer");
  for (int i = 0; i < ur::PC_COUNT; ++i) { host_ms[i] = 0; host_count[i] = 0; host_work[i] = 0; }
  for (auto& r : ur::g_prof) {
    UR_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    UR_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    host_ms[r.cls] += ms; host_count[r.cls] += 1; host_work[r.cls] += r.work;
  }
  return UR_OK;
}
