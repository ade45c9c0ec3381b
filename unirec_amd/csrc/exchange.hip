// Row exchange of the row-sharded embedding tables (SURVEY.md 8b `a2a_embedding_exchange`, 8e): fixed-capacity pack / unpack kernels and
// the RCCL transport on the caller's stream.
//
// Not in the reference (its only multi-GPU strategy is DDP over a replicated table, unirec/facility/trainer.py:67,261,346-349).  Here row i
// lives on rank i % W; a step moves (1) the batch's unique row ids to their owners, (2) the rows back, (3) the row gradients to the owners.
// Every (source, destination) pair carries EXACTLY `cap` slots in all three exchanges, so the split sizes are compile-time constants of
// the step: no per-step count exchange, no host synchronisation, every buffer preallocated.  A rank's block for owner o holds its
// count[o] requests right-aligned -- slots [o * cap + cap - count[o], (o + 1) * cap) -- behind padding slots that ask for local row 0 (the
// padding row of every shard: gathers zeros, never updated), so each block stays ascending (what the owner's W-way merge plan needs).
// A count above `cap - 1` raises a device flag; the flags of all ranks travel in slot 0 of every block of the gradient exchange (an
// all-gather riding in the all-to-all): every rank then skips the update (as for a NaN loss anywhere), and the host, which reads the
// flags two steps later, doubles the capacity and trains the batch again.
//
// Transport: the library owns one RCCL communicator per process (ur_comm_init; the unique id travels through the host's process group
// once) and issues ncclSend / ncclRecv groups on the stream it is given.  RCCL is resolved at run time from the librccl.so.1 the process
// has already loaded (torch's): nothing is linked, and a host without RCCL still loads the library (the gloo route of the tests).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace ur {

// slot q = o * cap + p of the send block.  Slot 0 of EVERY block is reserved padding (it carries the step's flags in the gradient
// exchange, and compact row 0 -- owner 0's slot 0 -- is the padding row the re-indexed lookups of id 0 read), so a block holds up to
// cap - 1 keys:  pad = cap - min(cap - 1, count[o]);  p < pad: padding (local row 0), else the (p - pad)-th key of owner o
// The keys are sorted by (owner, local row), so owner o's keys are the range [lower_bound(o * n_local), lower_bound((o + 1) * n_local)) of
// the unique list: W + 1 binary searches per workgroup (the list sits in L2), no counting pass -- counting the owners with one atomic per
// unique key in the plan's head kernel was 27 K atomics on W addresses: 300 us at W = 1, and it slowed every kernel running beside it.
__global__ __launch_bounds__(256) void shard_pack_kernel(const int* __restrict__ uniq_key, const int* __restrict__ n_uniq_dev,
                                                         int* __restrict__ counts_out, long long n_local, int W, int cap,
                                                         int* __restrict__ send_ids, int* __restrict__ slot_of_uniq,
                                                         int* __restrict__ u_of_slot, int* __restrict__ flags) {
  __shared__ int pre[65];
  if ((int)threadIdx.x <= W) {
    const int n_uniq = *n_uniq_dev;
    const unsigned long long bound = (unsigned long long)threadIdx.x * (unsigned long long)n_local;   // first key of owner threadIdx.x
    int lo = 0, hi = n_uniq;
    if ((int)threadIdx.x == W) lo = n_uniq;
    else if (W > 1)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((unsigned long long)(unsigned)uniq_key[mid] < bound) lo = mid + 1; else hi = mid;
      }
    else lo = 0;
    pre[threadIdx.x] = lo;
  }
  __syncthreads();
  if (counts_out && blockIdx.x == 0 && (int)threadIdx.x < W) counts_out[threadIdx.x] = pre[threadIdx.x + 1] - pre[threadIdx.x];
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= W * cap) return;
  const int o = q / cap, p = q % cap;
  const int cnt = pre[o + 1] - pre[o];
  // key 0 (the padding id, owner 0's first key when present) needs no slot of its own: it reads the reserved slot 0
  const int has0 = (o == 0 && cnt > 0 && uniq_key[0] == 0) ? 1 : 0;
  const int need = cnt - has0;
  if (p == 0 && need > cap - 1) atomicOr(flags, 1);   // overflow: the rows beyond the capacity are cut (the step is skipped, see the header)
  const int pad = cap - min(cap - 1, need);
  if (p == 0 && has0) slot_of_uniq[0] = 0;   // the padding id (unique key 0 of the plan) reads compact row 0
  if (p < pad) {
    send_ids[q] = 0;
    u_of_slot[q] = -1;
    return;
  }
  const int u = pre[o] + has0 + (p - pad);
  const unsigned key = (unsigned)uniq_key[u];
  send_ids[q] = W > 1 ? (int)(key % (unsigned long long)n_local) : (int)key;
  slot_of_uniq[u] = q;
  u_of_slot[q] = u;
}

// out[q, :] = u_of_slot[q] >= 0 ? rows[u_of_slot[q], :] : 0      (unique-order rows -> the fixed-capacity slot layout)
// The reserved slot 0 of every block carries this rank's step flags to every owner (an all-gather riding in the all-to-all):
// [loss is NaN, a capacity overflow, the loss, 1].  The owner's segment sum ignores the row (it belongs to local row 0).
__global__ __launch_bounds__(256) void shard_scatter_rows_kernel(const float4* __restrict__ rows, const int* __restrict__ u_of_slot,
                                                                 long long n_slots, int cap, int d4, const float* __restrict__ loss_out,
                                                                 const int* __restrict__ flags, float4* __restrict__ out,
                                                                 const int* __restrict__ guard_dev) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_slots * d4) return;
  const long long q = i / d4;
  const int c = (int)(i % d4);
  if (q % cap == 0 && c == 0) {
    const float loss = loss_out ? loss_out[0] : 0.f;
    // (a raised id guard on this rank reads as a NaN loss to every rank: the step is skipped EVERYWHERE, common.h)
    const float nan = ((loss_out && (loss_out[2] < 0.f || loss != loss)) || (guard_dev && *guard_dev)) ? 1.f : 0.f;
    out[i] = make_float4(nan, (flags && (flags[0] & 1)) ? 1.f : 0.f, nan != 0.f ? 0.f : loss, 1.f);
    return;
  }
  const int u = u_of_slot[q];
  out[i] = u >= 0 ? rows[(long long)u * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// the flag rows alone (the gradient rows were written into their slots by the reduction itself)
__global__ void shard_flag_rows_kernel(int world, int cap, int d4, const float* __restrict__ loss_out, const int* __restrict__ flags,
                                       float4* __restrict__ out, const int* __restrict__ guard_dev) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= world) return;
  const float loss = loss_out ? loss_out[0] : 0.f;
  const float nan = ((loss_out && (loss_out[2] < 0.f || loss != loss)) || (guard_dev && *guard_dev)) ? 1.f : 0.f;
  out[(long long)s * cap * d4] = make_float4(nan, (flags && (flags[0] & 1)) ? 1.f : 0.f, nan != 0.f ? 0.f : loss, 1.f);
}

// ---- round 4: the row exchange of batch t + 1 made a step AHEAD (plan stream, under step t's forward / backward); what the step in
// flight still changes -- the rows its owner-side plan updates -- is re-sent afterwards in a SMALL fixed-capacity exchange.
// shard_fixup_plan (ids only, next to the plan): a thread per slot of the NEXT batch's request list; a slot whose row is in `prev_uniq`
// (the sorted unique rows of THIS step's owner-side plan) takes the next free entry of its source block's list (one atomic per wave and
// block: the order inside a list is arbitrary, every entry carries its slot): req2[s * cap2 + i] = the row, slot2[..] = its slot inside
// the block (the requester adds owner * cap); the lists are pre-filled with padding (row 0, slot -1).  More than cap2 hot rows for one
// source: flags |= 1 -- the step is skipped everywhere and re-trained with doubled capacities, as for the main exchange.
// (One workgroup per source block walking its cap slots was 490 us at world 1 -- 28 K slots, 110 trips of three barriers.)
__global__ __launch_bounds__(256) void shard_fixup_plan_kernel(const int* __restrict__ recv_ids, int W, int cap, const int* __restrict__ prev_uniq,
                                                               const int* __restrict__ prev_n_dev, int prev_max, int cap2,
                                                               int* __restrict__ req2, int* __restrict__ slot2, int* __restrict__ counts,
                                                               int* __restrict__ flags) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int n_prev = min(*prev_n_dev, prev_max);
  const bool in = q < (long long)W * cap;
  const int s = in ? (int)(q / cap) : -1, p = in ? (int)(q % cap) : 0;
  const int row = (in && p > 0) ? recv_ids[q] : 0;
  bool hot = false;
  if (row != 0 && n_prev > 0) {
    int lo = 0, hi = n_prev;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (prev_uniq[mid] < row) lo = mid + 1; else hi = mid;
    }
    hot = lo < n_prev && prev_uniq[lo] == row;
  }
  // a wave spans at most two source blocks (cap >= 64 whenever world > 1)
  const int s0 = __shfl(s, 0, 64);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const bool mine = hot && (pass == 0 ? s == s0 : s != s0);
    const unsigned long long m = __ballot(mine);
    if (!m) continue;
    const int leader = __ffsll((long long)m) - 1;
    const int sb = __shfl(s, leader, 64);
    int base = 0;
    if (lane == leader) base = atomicAdd(&counts[sb], __popcll(m));
    base = __shfl(base, leader, 64);
    if (mine) {
      const int i = base + __popcll(m & ((1ULL << lane) - 1ULL));
      if (i < cap2) {
        req2[(long long)s * cap2 + i] = row;
        slot2[(long long)s * cap2 + i] = p;
      } else {
        atomicOr(flags, 1);
      }
    }
  }
}

// compact[(q / cap2) * cap + slot2[q], :] = rows2[q, :] for the received fix-up slots (slot2 >= 0)
__global__ __launch_bounds__(256) void shard_fixup_apply_kernel(float4* __restrict__ compact, const float4* __restrict__ rows2,
                                                                const int* __restrict__ slot2, long long n2, int cap, int cap2, int d4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2 * d4) return;
  const long long q = i / d4;
  const int c = (int)(i % d4), sl = slot2[q];
  if (sl >= 0) compact[((q / cap2) * cap + sl) * d4 + c] = rows2[i];
}

// the flags of all ranks, as received in slot 0 of every block of grads_in -> out[0] = gradient scale of the update kernels (1 / W =
// DDP's mean, or -1 = skip the step: a NaN loss or an overflow on ANY rank), out[1] = mean loss over the ranks, out[2] / out[3] = number
// of ranks with a NaN loss / an overflow
__global__ void shard_step_flags_kernel(const float* __restrict__ grads_in, int W, int cap, int d, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float nan = 0.f, ovf = 0.f, loss = 0.f;
  for (int s = 0; s < W; ++s) {   // source-rank order: every rank sums the same values in the same order
    const float* r = grads_in + (long long)s * cap * d;
    nan += r[0]; ovf += r[1]; loss += r[2];
  }
  out[0] = (nan > 0.f || ovf > 0.f) ? -1.f : 1.f / (float)W;
  out[1] = nan > 0.f ? __builtin_nanf("") : loss / (float)W;
  out[2] = nan;
  out[3] = ovf;
}

// ---- RCCL, resolved from the already-loaded library
namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;       // (optional: only ur_comm_count needs them)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  bool ok = false;
};
Rccl* rccl() {
  static Rccl* r = []() -> Rccl* {
    Rccl* x = new Rccl();
    x->h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!x->h) x->h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!x->h) return x;
#define UR_SYM(field, name) x->field = reinterpret_cast<decltype(x->field)>(dlsym(x->h, name))
    UR_SYM(GetUniqueId, "ncclGetUniqueId"); UR_SYM(CommInitRank, "ncclCommInitRank"); UR_SYM(CommDestroy, "ncclCommDestroy");
    UR_SYM(GroupStart, "ncclGroupStart"); UR_SYM(GroupEnd, "ncclGroupEnd"); UR_SYM(Send, "ncclSend"); UR_SYM(Recv, "ncclRecv");
    UR_SYM(AllReduce, "ncclAllReduce"); UR_SYM(GetErrorString, "ncclGetErrorString");
    UR_SYM(CommCount, "ncclCommCount"); UR_SYM(CommUserRank, "ncclCommUserRank");
#undef UR_SYM
    x->ok = x->GetUniqueId && x->CommInitRank && x->CommDestroy && x->GroupStart && x->GroupEnd && x->Send && x->Recv && x->AllReduce &&
            x->GetErrorString;
    return x;
  }();
  return r;
}
// two communicators: [0] the row exchanges (plan stream / caller's stream), [1] the dense all-reduce (the encoder's side stream) --
// operations on ONE communicator are serialised in issue order, and the all-reduce of step t must not hold up the rows of step t + 1
// UR_COMM_SINGLE=1 (read at ur_comm_init; the conservative rung of bench.py's fallback ladder): comm2 IS comm -- every collective of the
// process goes through one communicator in issue order (the caller then keeps them on one stream: UR_DENSE_SIDE=0, no lookahead)
struct Comm { ncclComm_t comm = nullptr, comm2 = nullptr; int rank = 0, world = 0; bool single = false; };
Comm g_comm;
}  // namespace

#define UR_NCCL(expr)                                                                                            \
  do {                                                                                                           \
    ncclResult_t _r = (expr);                                                                                    \
    if (_r != ncclSuccess) return ::ur::fail(UR_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(_r));    \
  } while (0)

// equal-split all-to-all: `bytes` per peer, block p of `send` -> rank p, block p of `recv` <- rank p
// RCCL runs the operations of ONE communicator in the order they were issued, whatever streams they are on (every launch waits for the
// communicator's previous one).  The id exchange of the NEXT batch is issued first in a step, on the plan stream, behind that batch's id
// sort: on the row communicator it would hold this step's row exchange -- issued later, on the main stream -- until the sort is done.
// It therefore goes through the SECOND communicator (the dense all-reduce's: that one is issued at the end of the step and waits for an
// id exchange that finished long before); rows and row gradients keep the first to themselves.
static int a2a_bytes(const void* send, void* recv, size_t bytes, hipStream_t st, ncclComm_t comm, int prof_class) {
  Rccl* r = rccl();
  const int W = g_comm.world;
  ProfScope ps(prof_class, st, (double)bytes * (W - 1));   // (events on the caller's stream around the group: what the stream spends in it)
  UR_NCCL(r->GroupStart());
  for (int p = 0; p < W; ++p) {
    UR_NCCL(r->Send((const char*)send + (size_t)p * bytes, bytes, ncclInt8, p, comm, st));
    UR_NCCL(r->Recv((char*)recv + (size_t)p * bytes, bytes, ncclInt8, p, comm, st));
  }
  UR_NCCL(r->GroupEnd());
  return UR_OK;
}

}  // namespace ur

using namespace ur;

// ---- in-process loopback transport (round 5; VERDICT r4 "missing" 2): W rank contexts in ONE process on ONE device -- own shards, own
// streams -- so that the step's real schedule (plan-stream prefetch, fix-up exchange, side-stream all-reduce) runs at W > 1 with true stream
// concurrency on a 1-GPU box: gloo stages every block through the host and thereby synchronises it.  A rank is a THREAD that called
// ur_loop_attach.  A collective is three non-blocking calls per rank with a host rendezvous of the rank threads (the caller's: a
// threading.Barrier -- it orders the ENQUEUEING only, the device never waits for the host) between them:
//   ur_loop_post        the stream first waits for the previous collective of the same communicator index (RCCL runs the operations of
//                       one communicator in issue order whatever their streams: reproduced here), then publishes the send buffer and
//                       records this rank's "ready" event;
//   ur_loop_*_pull      (all ranks have posted) the stream waits for every peer's "ready"; all-to-all: W device-to-device copies of
//                       the blocks addressed to this rank; all-reduce: one kernel sums the W buffers in rank order into a private
//                       buffer; then records this rank's "done" event;
//   ur_loop_finish      (all ranks have pulled) the stream waits for every peer's "done" -- nobody still reads this rank's send buffer --
//                       copies the all-reduce result in place, and records the communicator's "last operation" event.
// Every rank issues the same collectives in the same order (one Python thread per rank, the same program).  The slots (send pointer, ready /
// done events, the all-reduce's private buffer) are PER COMMUNICATOR INDEX, as RCCL's are: operations of the two indices run on different
// streams and may overlap on the device (one shared slot was correct only because every all-reduce used index 1).
namespace {
struct LoopRank {
  const void* send[2] = {nullptr, nullptr};
  hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, last[2] = {nullptr, nullptr};
  bool last_valid[2] = {false, false};
  float* ar_tmp[2] = {nullptr, nullptr};
  long long ar_tmp_n[2] = {0, 0};
};
struct LoopGroup { int world = 0; LoopRank rank[64]; };
struct LoopCtx { LoopGroup* g = nullptr; int rank = 0; };
thread_local LoopCtx t_loop;
struct LoopPtrs { const float* p[64]; };
__global__ __launch_bounds__(256) void loop_all_reduce_kernel(LoopPtrs src, int W, long long n, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float s = 0.f;
    for (int p = 0; p < W; ++p) s += src.p[p][i];   // rank order: every rank computes the same bits
    out[i] = s;
  }
}
}  // namespace

extern "C" void* ur_loop_create(int32_t world) {
  if (world < 1 || world > 64) { fail(UR_ERR_ARG, "ur_loop_create: world=%d", world); return nullptr; }
  LoopGroup* g = new LoopGroup();
  g->world = world;
  const unsigned evf = hipEventDisableTiming | (unsigned)hipEventDisableSystemFence;   // (one device: no system-scope fence, see sasrec.hip)
  for (int r = 0; r < world; ++r) {
    LoopRank& k = g->rank[r];
    if (hipEventCreateWithFlags(&k.ready[0], evf) != hipSuccess || hipEventCreateWithFlags(&k.done[0], evf) != hipSuccess ||
        hipEventCreateWithFlags(&k.ready[1], evf) != hipSuccess || hipEventCreateWithFlags(&k.done[1], evf) != hipSuccess ||
        hipEventCreateWithFlags(&k.last[0], evf) != hipSuccess || hipEventCreateWithFlags(&k.last[1], evf) != hipSuccess) {
      fail(UR_ERR_HIP, "ur_loop_create: event creation failed");
      return nullptr;
    }
  }
  return g;
}
extern "C" int ur_loop_destroy(void* group) {
  LoopGroup* g = (LoopGroup*)group;
  if (!g) return UR_OK;
  for (int r = 0; r < g->world; ++r) {
    LoopRank& k = g->rank[r];
    for (int c = 0; c < 2; ++c) {
      (void)hipEventDestroy(k.ready[c]); (void)hipEventDestroy(k.done[c]); (void)hipEventDestroy(k.last[c]);
      if (k.ar_tmp[c]) (void)hipFree(k.ar_tmp[c]);
    }
  }
  delete g;
  return UR_OK;
}
// binds the CALLING THREAD to (group, rank); its per-context state (the encoder's side stream and events, hand-off counters) is the
// context rank + 1's from now on (common.h: g_ctx_id).  ur_loop_detach: back to the process's default context.
extern "C" int ur_loop_attach(void* group, int32_t rank) {
  LoopGroup* g = (LoopGroup*)group;
  UR_REQUIRE(g && rank >= 0 && rank < g->world, UR_ERR_ARG, "ur_loop_attach: rank %d", rank);
  t_loop = LoopCtx{g, rank};
  g_ctx_id = rank + 1;
  return UR_OK;
}
extern "C" int ur_loop_detach(void) {
  t_loop = LoopCtx{};
  g_ctx_id = 0;
  return UR_OK;
}
extern "C" int ur_loop_post(const void* send, int32_t comm, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(t_loop.g && send && (comm == 0 || comm == 1), UR_ERR_ARG, "ur_loop_post: not attached / bad argument");
  LoopRank& me = t_loop.g->rank[t_loop.rank];
  hipStream_t st = as_stream(stream);
  if (me.last_valid[comm]) UR_HIP(hipStreamWaitEvent(st, me.last[comm], 0));   // one communicator: operations in issue order
  me.send[comm] = send;
  UR_HIP(hipEventRecord(me.ready[comm], st));
  return UR_OK;
}
extern "C" int ur_loop_all_to_all_pull(void* recv, int64_t bytes_per_peer, int32_t kind, int32_t comm, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(t_loop.g && recv && bytes_per_peer > 0 && kind >= 0 && kind <= 2 && (comm == 0 || comm == 1), UR_ERR_ARG,
             "ur_loop_all_to_all_pull: not attached / bad argument");
  LoopGroup& g = *t_loop.g;
  hipStream_t st = as_stream(stream);
  const int cls = kind == 0 ? PC_A2A_IDS : kind == 1 ? PC_A2A_ROWS : PC_A2A_GRADS;
  for (int p = 0; p < g.world; ++p) UR_HIP(hipStreamWaitEvent(st, g.rank[p].ready[comm], 0));
  {
    ProfScope ps(cls, st, (double)bytes_per_peer * (g.world - 1));   // (the copies alone: what a loopback run subtracts from a rank's device time)
    for (int p = 0; p < g.world; ++p)
      UR_HIP(hipMemcpyAsync((char*)recv + (size_t)p * bytes_per_peer, (const char*)g.rank[p].send[comm] + (size_t)t_loop.rank * bytes_per_peer,
                            (size_t)bytes_per_peer, hipMemcpyDeviceToDevice, st));
  }
  UR_HIP(hipEventRecord(g.rank[t_loop.rank].done[comm], st));
  return UR_OK;
}
extern "C" int ur_loop_all_reduce_pull(int64_t n, int32_t comm, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(t_loop.g && n > 0 && (comm == 0 || comm == 1), UR_ERR_ARG, "ur_loop_all_reduce_pull: not attached / bad argument");
  LoopGroup& g = *t_loop.g;
  LoopRank& me = g.rank[t_loop.rank];
  hipStream_t st = as_stream(stream);
  if (me.ar_tmp_n[comm] < n) {   // (grown on first use: a hipMalloc synchronises the device once, not per step)
    if (me.ar_tmp[comm]) UR_HIP(hipFree(me.ar_tmp[comm]));
    UR_HIP(hipMalloc((void**)&me.ar_tmp[comm], (size_t)n * sizeof(float)));
    me.ar_tmp_n[comm] = n;
  }
  LoopPtrs src{};
  for (int p = 0; p < g.world; ++p) {
    UR_HIP(hipStreamWaitEvent(st, g.rank[p].ready[comm], 0));
    src.p[p] = (const float*)g.rank[p].send[comm];
  }
  {
    ProfScope ps(PC_ALLREDUCE, st, (double)n * 4.0 * 2.0 * (g.world - 1) / g.world);
    hipLaunchKernelGGL(loop_all_reduce_kernel, dim3((unsigned)std::min<long long>(1024, (n + 255) / 256)), dim3(256), 0, st, src, g.world, (long long)n, me.ar_tmp[comm]);
    UR_LAUNCH_CHECK();
  }
  UR_HIP(hipEventRecord(me.done[comm], st));
  return UR_OK;
}
extern "C" int ur_loop_finish(int32_t comm, float* all_reduce_out, int64_t n, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(t_loop.g && (comm == 0 || comm == 1), UR_ERR_ARG, "ur_loop_finish: not attached / bad argument");
  LoopGroup& g = *t_loop.g;
  LoopRank& me = g.rank[t_loop.rank];
  hipStream_t st = as_stream(stream);
  for (int p = 0; p < g.world; ++p) UR_HIP(hipStreamWaitEvent(st, g.rank[p].done[comm], 0));
  if (all_reduce_out) {
    UR_REQUIRE(n > 0 && n <= me.ar_tmp_n[comm], UR_ERR_ARG, "ur_loop_finish: n=%lld", (long long)n);
    UR_HIP(hipMemcpyAsync(all_reduce_out, me.ar_tmp[comm], (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  UR_HIP(hipEventRecord(me.last[comm], st));
  me.last_valid[comm] = true;
  return UR_OK;
}

extern "C" int ur_comm_unique_id(void* id_out) {
  UR_REQUIRE(id_out, UR_ERR_ARG, "ur_comm_unique_id: null pointer");
  UR_REQUIRE(rccl()->ok, UR_ERR_UNSUPPORTED, "ur_comm_unique_id: no RCCL library in this process");
  ncclUniqueId id[2];
  UR_NCCL(rccl()->GetUniqueId(&id[0]));
  UR_NCCL(rccl()->GetUniqueId(&id[1]));
  memcpy(id_out, id, sizeof(id));
  return UR_OK;
}

extern "C" int ur_comm_init(const void* id, int32_t rank, int32_t world) {
  UR_REQUIRE(id && world >= 1 && rank >= 0 && rank < world, UR_ERR_ARG, "ur_comm_init: rank %d of %d", rank, world);
  UR_REQUIRE(rccl()->ok, UR_ERR_UNSUPPORTED, "ur_comm_init: no RCCL library in this process");
  if (g_comm.comm) {
    UR_REQUIRE(g_comm.rank == rank && g_comm.world == world, UR_ERR_ARG, "ur_comm_init: already initialised as rank %d of %d",
               g_comm.rank, g_comm.world);
    return UR_OK;
  }
  ncclUniqueId uid[2];
  memcpy(uid, id, sizeof(uid));
  UR_NCCL(rccl()->CommInitRank(&g_comm.comm, world, uid[0], rank));
  g_comm.single = getenv("UR_COMM_SINGLE") && atoi(getenv("UR_COMM_SINGLE")) != 0;
  if (g_comm.single) g_comm.comm2 = g_comm.comm;
  else UR_NCCL(rccl()->CommInitRank(&g_comm.comm2, world, uid[1], rank));
  g_comm.rank = rank;
  g_comm.world = world;
  return UR_OK;
}

extern "C" int ur_comm_world(void) { return !rccl()->ok ? -1 : (g_comm.comm ? g_comm.world : 0); }

// What RCCL ITSELF says about the library's communicators (ncclCommCount / ncclCommUserRank of each): out[0..1] = ranks in the row /
// the ahead communicator, out[2..3] = this process's rank in them.  Returns the rank count both agree on, 0 when not initialised,
// < 0 on error (no RCCL, the two communicators disagree, or they disagree with what ur_comm_init was told).
extern "C" int ur_comm_count(int32_t* out4) {
  UR_REQUIRE(rccl()->ok, UR_ERR_UNSUPPORTED, "ur_comm_count: no RCCL library in this process");
  if (!g_comm.comm) return 0;
  UR_REQUIRE(rccl()->CommCount && rccl()->CommUserRank, UR_ERR_UNSUPPORTED, "ur_comm_count: this RCCL exports no ncclCommCount / ncclCommUserRank");
  int n[2] = {0, 0}, r[2] = {-1, -1};
  UR_NCCL(rccl()->CommCount(g_comm.comm, &n[0]));
  UR_NCCL(rccl()->CommCount(g_comm.comm2, &n[1]));
  UR_NCCL(rccl()->CommUserRank(g_comm.comm, &r[0]));
  UR_NCCL(rccl()->CommUserRank(g_comm.comm2, &r[1]));
  if (out4) { out4[0] = n[0]; out4[1] = n[1]; out4[2] = r[0]; out4[3] = r[1]; }
  UR_REQUIRE(n[0] == n[1] && n[0] == g_comm.world && r[0] == r[1] && r[0] == g_comm.rank, UR_ERR_ARG,
             "ur_comm_count: RCCL reports %d / %d ranks (this process %d / %d), ur_comm_init was told rank %d of %d", n[0], n[1], r[0], r[1],
             g_comm.rank, g_comm.world);
  return n[0];
}
extern "C" int ur_loop_world(void) { return t_loop.g ? t_loop.g->world : 0; }

extern "C" int ur_comm_destroy(void) {
  if (g_comm.comm) {
    UR_NCCL(rccl()->CommDestroy(g_comm.comm));
    if (g_comm.comm2 && !g_comm.single) UR_NCCL(rccl()->CommDestroy(g_comm.comm2));
    g_comm = Comm{};
  }
  return UR_OK;
}

extern "C" int ur_comm_all_reduce_sum(float* buf, int64_t n, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(buf && n > 0, UR_ERR_ARG, "ur_comm_all_reduce_sum: bad argument");
  UR_REQUIRE(g_comm.comm, UR_ERR_ARG, "ur_comm_all_reduce_sum: no communicator (ur_comm_init)");
  ProfScope ps(PC_ALLREDUCE, as_stream(stream), (double)n * 4.0 * 2.0 * (g_comm.world - 1) / g_comm.world);   // (ring all-reduce: 2 (W - 1) / W of the buffer leaves the rank)
  UR_NCCL(rccl()->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, g_comm.comm2, as_stream(stream)));
  return UR_OK;
}

// equal-split all-to-all of a packed buffer through the library's communicators: ahead != 0 = the second one (the dense all-reduce's and
// the id exchange's: work issued a step ahead on the plan stream), else the row communicator of the step's own exchanges.
extern "C" int ur_comm_all_to_all(const void* send, void* recv, int64_t bytes_per_peer, int32_t ahead, int32_t kind, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(send && recv && bytes_per_peer > 0 && kind >= 0 && kind <= 2, UR_ERR_ARG, "ur_comm_all_to_all: bad argument");
  UR_REQUIRE(g_comm.comm, UR_ERR_ARG, "ur_comm_all_to_all: no communicator (ur_comm_init)");
  const int cls = kind == 0 ? PC_A2A_IDS : kind == 1 ? PC_A2A_ROWS : PC_A2A_GRADS;   // (which per-collective timer the group is booked on)
  return a2a_bytes(send, recv, (size_t)bytes_per_peer, as_stream(stream), ahead ? g_comm.comm2 : g_comm.comm, cls);
}

extern "C" int ur_shard_fixup_plan(const int32_t* recv_ids, int32_t world, int32_t cap, const int32_t* prev_uniq,
                                   const int32_t* prev_n_uniq_dev, int64_t prev_n_max, int32_t cap2, int32_t* req2, int32_t* slot2,
                                   int32_t* counts_ws, int32_t* flags_dev, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(recv_ids && req2 && slot2 && counts_ws && flags_dev, UR_ERR_ARG, "ur_shard_fixup_plan: null pointer");
  UR_REQUIRE(world >= 1 && cap > 0 && cap2 > 0 && cap2 <= cap && (world == 1 || cap >= 64) &&
                 (prev_uniq == nullptr || (prev_n_uniq_dev && prev_n_max > 0 && prev_n_max < (1LL << 31))),
             UR_ERR_ARG, "ur_shard_fixup_plan: world=%d cap=%d cap2=%d", world, cap, cap2);
  hipStream_t st = as_stream(stream);
  const size_t n2 = (size_t)world * cap2;
  UR_HIP(hipMemsetAsync(req2, 0, n2 * sizeof(int32_t), st));
  UR_HIP(hipMemsetAsync(slot2, 0xFF, n2 * sizeof(int32_t), st));   // -1
  if (!prev_uniq) return UR_OK;                                     // (nothing in flight: every list is padding)
  UR_HIP(hipMemsetAsync(counts_ws, 0, (size_t)world * sizeof(int32_t), st));
  hipLaunchKernelGGL(shard_fixup_plan_kernel, dim3(cdiv((long long)world * cap, 256)), dim3(256), 0, st, recv_ids, world, cap, prev_uniq,
                     prev_n_uniq_dev, (int)prev_n_max, cap2, req2, slot2, counts_ws, flags_dev);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

extern "C" int ur_shard_fixup_apply(float* compact, const float* rows2, const int32_t* slot2, int32_t world, int32_t cap, int32_t cap2,
                                    int32_t d, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(compact && rows2 && slot2, UR_ERR_ARG, "ur_shard_fixup_apply: null pointer");
  UR_REQUIRE(world >= 1 && cap > 0 && cap2 > 0 && d > 0 && d % 4 == 0, UR_ERR_ARG, "ur_shard_fixup_apply: world=%d cap=%d cap2=%d d=%d", world, cap, cap2, d);
  const long long n2 = (long long)world * cap2;
  hipLaunchKernelGGL(shard_fixup_apply_kernel, dim3(cdiv(n2 * (d / 4), 256)), dim3(256), 0, as_stream(stream), (float4*)compact,
                     (const float4*)rows2, slot2, n2, cap, cap2, d / 4);
  UR_LAUNCH_CHECK();
  return UR_OK;
}

// (1) ids: pack the plan's unique keys into the fixed-capacity send block, then (communicator up) all-to-all into recv_ids.
// transport == 0: pack only -- the caller moves send_ids itself (the gloo route of the CPU-staged tests).
extern "C" int ur_shard_exchange_ids(const int32_t* uniq_key, const int32_t* n_uniq_dev, int32_t* counts_dev, int64_t n_local,
                                     int32_t world, int32_t cap, int32_t* send_ids, int32_t* slot_of_uniq, int32_t* u_of_slot,
                                     int32_t* flags_dev, int32_t* recv_ids, int32_t transport, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(uniq_key && n_uniq_dev && send_ids && slot_of_uniq && u_of_slot && flags_dev, UR_ERR_ARG,
             "ur_shard_exchange_ids: null pointer");
  UR_REQUIRE(world >= 1 && world <= 64 && cap > 0 && (long long)world * cap < (1LL << 31), UR_ERR_ARG,
             "ur_shard_exchange_ids: world=%d cap=%d", world, cap);
  hipStream_t st = as_stream(stream);
  ProfScope ps(PC_SORT, st, (double)world * cap * 12.0);
  hipLaunchKernelGGL(shard_pack_kernel, dim3(cdiv((long long)world * cap, 256)), dim3(256), 0, st, uniq_key, n_uniq_dev, counts_dev,
                     (long long)n_local, world, cap, send_ids, slot_of_uniq, u_of_slot, flags_dev);
  UR_LAUNCH_CHECK();
  if (!transport) return UR_OK;
  UR_REQUIRE(recv_ids, UR_ERR_ARG, "ur_shard_exchange_ids: null receive buffer");
  UR_REQUIRE(g_comm.comm && g_comm.world == world, UR_ERR_ARG, "ur_shard_exchange_ids: communicator of %d ranks, world=%d", g_comm.world, world);
  return a2a_bytes(send_ids, recv_ids, (size_t)cap * sizeof(int32_t), st, g_comm.comm2, PC_A2A_IDS);
}

// (2) rows: gather the requested rows of this rank's shard (req_ids: world * cap local rows, as received) into rows_ws, then
// all-to-all into compact [world * cap, d]: row q of `compact` is the row slot q of ur_shard_exchange_ids asked for.
extern "C" int ur_shard_exchange_rows(const float* table, const int32_t* req_ids, int32_t world, int32_t cap, int32_t d, float* rows_ws,
                                      float* compact, int32_t transport, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(table && req_ids && rows_ws, UR_ERR_ARG, "ur_shard_exchange_rows: null pointer");
  UR_REQUIRE(world >= 1 && cap > 0 && d > 0 && d % 4 == 0, UR_ERR_ARG, "ur_shard_exchange_rows: world=%d cap=%d d=%d", world, cap, d);
  hipStream_t st = as_stream(stream);
  int rc = gather_rows(table, req_ids, 4, (long long)world * cap, d, rows_ws, st);
  if (rc || !transport) return rc;
  UR_REQUIRE(compact, UR_ERR_ARG, "ur_shard_exchange_rows: null receive buffer");
  UR_REQUIRE(g_comm.comm && g_comm.world == world, UR_ERR_ARG, "ur_shard_exchange_rows: communicator of %d ranks, world=%d", g_comm.world, world);
  return a2a_bytes(rows_ws, compact, (size_t)cap * d * sizeof(float), st, g_comm.comm, PC_A2A_ROWS);
}

// (3) row gradients: uniq_grad [n_uniq, d] (unique order, from ur_rows_reduce) -> slot layout (padding slots: zeros) in send_ws, then
// all-to-all into grads_in [world * cap, d] on the owners (block s = what rank s sends: summed in source-rank order by the owner's plan).
extern "C" int ur_shard_exchange_grads(const float* uniq_grad, const int32_t* u_of_slot, int32_t world, int32_t cap, int32_t d,
                                       const float* loss_out, const int32_t* flags_dev, float* send_ws, float* grads_in,
                                       int32_t transport, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE((uniq_grad == nullptr || u_of_slot) && send_ws, UR_ERR_ARG, "ur_shard_exchange_grads: null pointer");
  UR_REQUIRE(world >= 1 && cap > 0 && d > 0 && d % 4 == 0, UR_ERR_ARG, "ur_shard_exchange_grads: world=%d cap=%d d=%d", world, cap, d);
  hipStream_t st = as_stream(stream);
  const long long n_slots = (long long)world * cap;
  if (!uniq_grad) {
    hipLaunchKernelGGL(shard_flag_rows_kernel, dim3(cdiv(world, 64)), dim3(64), 0, st, world, cap, d / 4, loss_out, flags_dev, (float4*)send_ws, id_guard().dev);
    UR_LAUNCH_CHECK();
  } else {
    ProfScope ps(PC_REDUCE, st, (double)n_slots * d * 8.0);
    hipLaunchKernelGGL(shard_scatter_rows_kernel, dim3(cdiv(n_slots * (d / 4), 256)), dim3(256), 0, st, (const float4*)uniq_grad, u_of_slot,
                       n_slots, cap, d / 4, loss_out, flags_dev, (float4*)send_ws, id_guard().dev);
    UR_LAUNCH_CHECK();
  }
  if (!transport) return UR_OK;
  UR_REQUIRE(grads_in, UR_ERR_ARG, "ur_shard_exchange_grads: null receive buffer");
  UR_REQUIRE(g_comm.comm && g_comm.world == world, UR_ERR_ARG, "ur_shard_exchange_grads: communicator of %d ranks, world=%d", g_comm.world, world);
  return a2a_bytes(send_ws, grads_in, (size_t)cap * d * sizeof(float), st, g_comm.comm, PC_A2A_GRADS);
}

// after (3): the flags every rank put into slot 0 of its blocks -> out4 = [gradient scale (1 / world, or -1 = skip the step), mean loss,
// ranks with a NaN loss, ranks with a capacity overflow]
extern "C" int ur_shard_step_flags(const float* grads_in, int32_t world, int32_t cap, int32_t d, float* out4, void* stream) {
  UR_TRACE_SCOPE();
  UR_REQUIRE(grads_in && out4 && world >= 1 && cap > 0 && d >= 4, UR_ERR_ARG, "ur_shard_step_flags: bad argument");
  hipLaunchKernelGGL(shard_step_flags_kernel, dim3(1), dim3(64), 0, as_stream(stream), grads_in, world, cap, d, out4);
  UR_LAUNCH_CHECK();
  return UR_OK;
}
