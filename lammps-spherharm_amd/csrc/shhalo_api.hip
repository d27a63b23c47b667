// shhalo_api.hip — the C ABI of include/shhalo.h: LAMMPS' Comm::exchange / borders / forward_comm / reverse_comm
// around PairSH::compute for a device-resident host, one rank per GPU.
//
// Host side: the brick geometry and message layout (halo_plan.cpp), the device buffers of the plan, the two
// transports, and Verlet::run over all ranks.  Per step and direction of travel there is ONE pack kernel, ONE
// grouped point-to-point exchange (ncclGroupStart .. one ncclSend + one ncclRecv per remote peer .. ncclGroupEnd)
// and ONE unpack kernel on the caller's stream; the host waits only where a count must be read back (at a
// reneighbouring) and for the rebuild decision.
//
// Transports:
//   RCCL    librccl is bound at run time (dlopen; the copy already in the process — e.g. PyTorch's — is preferred,
//           so that there is one RCCL and one HIP runtime per process).  ncclSend/ncclRecv over xGMI.
//   local   the ranks are host threads of ONE process that share a hub: a send posts (pointer, bytes, ready event),
//           the matching receive enqueues a device copy behind that event on the receiver's stream.  Same message
//           pattern, same kernels; for rehearsing N ranks on fewer than N GPUs and for a self-periodic single rank.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/shhalo.h"
#include "halo_kernels.hpp"
#include "shpair_ctx.hpp"

using namespace shp;

namespace {

// ------------------------------------------------------------------------------------------------ RCCL binding
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string error;
};

RcclApi* rccl_api()
{
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {  // the copy the process already holds (PyTorch's), if any
      api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (api.handle) break;
    }
    for (int k = 0; k < 2 && !api.handle; ++k) api.handle = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) {
      const char* e = dlerror();
      api.error = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "unknown dlopen error");
      return;
    }
    auto sym = [&](const char* s) -> void* {
      void* p = dlsym(api.handle, s);
      if (!p && api.error.empty()) api.error = std::string("librccl does not export ") + s;
      return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
  });
  return &api;
}

struct Msg {
  int peer;
  void* ptr;
  size_t bytes;
};

struct Transport {
  std::string err;
  virtual ~Transport() {}
  // one grouped exchange: at most one send and one receive per peer; zero-byte messages are left out by the caller
  virtual int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t st) = 0;
  virtual int allreduce_max_i32(int* dev, int n, hipStream_t st) = 0;   // in place
  virtual int allreduce_sum_f64(double* dev, int n, hipStream_t st) = 0;
  virtual int size() const = 0;
  virtual int kind() const = 0;  // 0 local, 1 RCCL
  virtual int version() const { return 0; }
};

struct RcclTransport : Transport {
  RcclApi* api = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 0;
  ~RcclTransport() override
  {
    if (comm && api) (void)api->CommDestroy(comm);
  }
  int fail(const char* what, ncclResult_t r)
  {
    err = std::string(what) + " failed: " + (api && api->GetErrorString ? api->GetErrorString(r) : "?");
    return SHPAIR_EHIP;
  }
  int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t st) override
  {
    if (sends.empty() && recvs.empty()) return SHPAIR_OK;
    ncclResult_t r = api->GroupStart();
    if (r != ncclSuccess) return fail("ncclGroupStart", r);
    for (const Msg& m : recvs) {
      r = api->Recv(m.ptr, m.bytes, ncclChar, m.peer, comm, st);
      if (r != ncclSuccess) break;
    }
    if (r == ncclSuccess)
      for (const Msg& m : sends) {
        r = api->Send(m.ptr, m.bytes, ncclChar, m.peer, comm, st);
        if (r != ncclSuccess) break;
      }
    const ncclResult_t r2 = api->GroupEnd();
    if (r != ncclSuccess) return fail("ncclSend/ncclRecv", r);
    if (r2 != ncclSuccess) return fail("ncclGroupEnd", r2);
    return SHPAIR_OK;
  }
  int allreduce_max_i32(int* dev, int n, hipStream_t st) override
  {
    const ncclResult_t r = api->AllReduce(dev, dev, (size_t)n, ncclInt32, ncclMax, comm, st);
    return r == ncclSuccess ? SHPAIR_OK : fail("ncclAllReduce", r);
  }
  int allreduce_sum_f64(double* dev, int n, hipStream_t st) override
  {
    const ncclResult_t r = api->AllReduce(dev, dev, (size_t)n, ncclFloat64, ncclSum, comm, st);
    return r == ncclSuccess ? SHPAIR_OK : fail("ncclAllReduce", r);
  }
  int size() const override { return nranks; }
  int kind() const override { return 1; }
  int version() const override
  {
    int v = 0;
    if (api && api->GetVersion) (void)api->GetVersion(&v);
    return v;
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------------ local hub
struct shhalo_hub {
  struct Post {
    const void* src;
    size_t bytes;
    hipEvent_t ready = nullptr, done = nullptr;
    bool consumed = false;
  };
  int nranks = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::deque<Post*>> box;  // [src * nranks + dst]
  // all-reduce
  std::vector<double> acc, result;
  int arrived = 0;
  unsigned long long generation = 0;
};

namespace {

// a rank thread that failed must not leave the others waiting for ever
std::chrono::seconds hub_timeout()
{
  static const long s = [] {
    const char* e = getenv("SHHALO_HUB_TIMEOUT_S");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 120L;
  }();
  return std::chrono::seconds(s);
}

struct LocalTransport : Transport {
  shhalo_hub* hub = nullptr;  // null: single rank
  int rank = 0, nranks = 1;
  int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t st) override
  {
    if (sends.empty() && recvs.empty()) return SHPAIR_OK;
    if (!hub) {
      err = "local transport without a hub was asked to talk to another rank";
      return SHPAIR_ESTATE;
    }
    std::vector<shhalo_hub::Post*> mine;
    for (const Msg& m : sends) {
      shhalo_hub::Post* p = new shhalo_hub::Post();
      p->src = m.ptr;
      p->bytes = m.bytes;
      if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess || hipEventRecord(p->ready, st) != hipSuccess) {
        err = "hub: event creation failed";
        return SHPAIR_EHIP;
      }
      mine.push_back(p);
      {
        std::lock_guard<std::mutex> lk(hub->mu);
        hub->box[(size_t)rank * nranks + m.peer].push_back(p);
      }
      hub->cv.notify_all();
    }
    for (const Msg& m : recvs) {
      shhalo_hub::Post* p = nullptr;
      {
        std::unique_lock<std::mutex> lk(hub->mu);
        auto& q = hub->box[(size_t)m.peer * nranks + rank];
        if (!hub->cv.wait_for(lk, hub_timeout(), [&] { return !q.empty(); })) {
          err = "hub: rank " + std::to_string(rank) + " timed out waiting for a message from rank " + std::to_string(m.peer) +
                " (did that rank fail?)";
          return SHPAIR_ESTATE;
        }
        p = q.front();
        q.pop_front();
      }
      if (p->bytes != m.bytes) {
        err = "hub: a message from rank " + std::to_string(m.peer) + " has " + std::to_string(p->bytes) + " bytes, " +
              std::to_string(m.bytes) + " expected";
        return SHPAIR_ESTATE;
      }
      if (hipStreamWaitEvent(st, p->ready, 0) != hipSuccess ||
          hipMemcpyAsync(m.ptr, p->src, m.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipEventRecord(p->done, st) != hipSuccess) {
        err = "hub: device copy failed";
        return SHPAIR_EHIP;
      }
      {
        std::lock_guard<std::mutex> lk(hub->mu);
        p->consumed = true;
      }
      hub->cv.notify_all();
    }
    // the send buffers may be rewritten only after the receivers' copies: this rank's stream waits for them
    for (shhalo_hub::Post* p : mine) {
      {
        std::unique_lock<std::mutex> lk(hub->mu);
        if (!hub->cv.wait_for(lk, hub_timeout(), [&] { return p->consumed; })) {
          err = "hub: rank " + std::to_string(rank) + " timed out waiting for a receiver (did that rank fail?)";
          return SHPAIR_ESTATE;  // the post stays with the hub: the receiver may still come for it
        }
      }
      (void)hipStreamWaitEvent(st, p->done, 0);
      (void)hipEventDestroy(p->ready);
      (void)hipEventDestroy(p->done);
      delete p;
    }
    return SHPAIR_OK;
  }
  template <typename T, typename OP>
  int allreduce(T* dev, int n, hipStream_t st, OP op)
  {
    if (!hub || nranks == 1) return SHPAIR_OK;
    std::vector<T> h((size_t)n);
    if (hipMemcpyAsync(h.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      err = "hub: all-reduce read-back failed";
      return SHPAIR_EHIP;
    }
    {
      std::unique_lock<std::mutex> lk(hub->mu);
      if (hub->arrived == 0) hub->acc.assign((size_t)n, 0.0);
      for (int k = 0; k < n; ++k) hub->acc[k] = hub->arrived == 0 ? (double)h[k] : op(hub->acc[k], (double)h[k]);
      if (++hub->arrived == nranks) {
        hub->result = hub->acc;
        hub->arrived = 0;
        ++hub->generation;
        hub->cv.notify_all();
      } else {
        const unsigned long long g = hub->generation;
        if (!hub->cv.wait_for(lk, hub_timeout(), [&] { return hub->generation != g; })) {
          err = "hub: rank " + std::to_string(rank) + " timed out in an all-reduce (did another rank fail?)";
          return SHPAIR_ESTATE;
        }
      }
      for (int k = 0; k < n; ++k) h[k] = (T)hub->result[k];
    }
    if (hipMemcpyAsync(dev, h.data(), n * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      err = "hub: all-reduce write-back failed";
      return SHPAIR_EHIP;
    }
    return SHPAIR_OK;
  }
  int allreduce_max_i32(int* dev, int n, hipStream_t st) override
  {
    return allreduce(dev, n, st, [](double a, double b) { return a > b ? a : b; });
  }
  int allreduce_sum_f64(double* dev, int n, hipStream_t st) override
  {
    return allreduce(dev, n, st, [](double a, double b) { return a + b; });
  }
  int size() const override { return nranks; }
  int kind() const override { return 0; }
};

// Host-staged transport: the caller's functions move the bytes (MPI in a LAMMPS host, gloo in bench.py); the library
// stages through page-locked host memory around them.  Every call blocks the host until its messages are complete.
struct StagedTransport : Transport {
  shhalo_exchange_fn xfn = nullptr;
  shhalo_allreduce_fn rfn = nullptr;
  void* user = nullptr;
  int rank = 0, nranks = 1;
  unsigned char* hbuf = nullptr;   // pinned: [send bytes | recv bytes]
  size_t hcap = 0;
  ~StagedTransport() override
  {
    if (hbuf) (void)hipHostFree(hbuf);
  }
  int ensure(size_t bytes)
  {
    if (bytes <= hcap) return SHPAIR_OK;
    if (hbuf) (void)hipHostFree(hbuf);
    hbuf = nullptr;
    hcap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipHostMalloc((void**)&hbuf, want) != hipSuccess) {
      (void)hipGetLastError();
      err = "staged transport: hipHostMalloc of " + std::to_string(want) + " bytes failed";
      return SHPAIR_ENOMEM;
    }
    hcap = want;
    return SHPAIR_OK;
  }
  int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, hipStream_t st) override
  {
    if (sends.empty() && recvs.empty()) return SHPAIR_OK;
    size_t sb = 0, rb = 0;
    for (const Msg& m : sends) sb += (m.bytes + 15) & ~(size_t)15;
    for (const Msg& m : recvs) rb += (m.bytes + 15) & ~(size_t)15;
    // the previous exchange's upward copies read this buffer: they are complete (this call ended with a stream wait)
    if (const int rc = ensure(sb + rb)) return rc;
    std::vector<int> sp_, rp_;
    std::vector<void*> sptr, rptr;
    std::vector<size_t> sby, rby;
    size_t off = 0;
    for (const Msg& m : sends) {
      if (hipMemcpyAsync(hbuf + off, m.ptr, m.bytes, hipMemcpyDeviceToHost, st) != hipSuccess) {
        err = "staged transport: copy of a send buffer to the host failed";
        return SHPAIR_EHIP;
      }
      sp_.push_back(m.peer); sptr.push_back(hbuf + off); sby.push_back(m.bytes);
      off += (m.bytes + 15) & ~(size_t)15;
    }
    for (const Msg& m : recvs) {
      rp_.push_back(m.peer); rptr.push_back(hbuf + off); rby.push_back(m.bytes);
      off += (m.bytes + 15) & ~(size_t)15;
    }
    if (hipStreamSynchronize(st) != hipSuccess) {
      err = "staged transport: hipStreamSynchronize failed";
      return SHPAIR_EHIP;
    }
    const int xrc = xfn(user, (int)sends.size(), sp_.data(), sptr.data(), sby.data(), (int)recvs.size(), rp_.data(), rptr.data(),
                        rby.data());
    if (xrc != 0) {
      err = "staged transport: the caller's exchange function returned " + std::to_string(xrc) + " on rank " + std::to_string(rank);
      return SHPAIR_ESTATE;
    }
    for (size_t k = 0; k < recvs.size(); ++k)
      if (hipMemcpyAsync(recvs[k].ptr, rptr[k], recvs[k].bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
        err = "staged transport: copy of a received buffer to the device failed";
        return SHPAIR_EHIP;
      }
    if (hipStreamSynchronize(st) != hipSuccess) {   // the host buffer is free again, and the caller's next call may be another exchange
      err = "staged transport: hipStreamSynchronize failed";
      return SHPAIR_EHIP;
    }
    return SHPAIR_OK;
  }
  template <typename T>
  int allreduce(T* dev, int n, int kind, hipStream_t st)
  {
    if (nranks == 1) return SHPAIR_OK;
    if (const int rc = ensure((size_t)n * sizeof(T))) return rc;
    if (hipMemcpyAsync(hbuf, dev, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      err = "staged transport: all-reduce read-back failed";
      return SHPAIR_EHIP;
    }
    const int rrc = rfn(user, hbuf, n, kind);
    if (rrc != 0) {
      err = "staged transport: the caller's all-reduce function returned " + std::to_string(rrc) + " on rank " + std::to_string(rank);
      return SHPAIR_ESTATE;
    }
    if (hipMemcpyAsync(dev, hbuf, (size_t)n * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      err = "staged transport: all-reduce write-back failed";
      return SHPAIR_EHIP;
    }
    return SHPAIR_OK;
  }
  int allreduce_max_i32(int* dev, int n, hipStream_t st) override { return allreduce(dev, n, 0, st); }
  int allreduce_sum_f64(double* dev, int n, hipStream_t st) override { return allreduce(dev, n, 1, st); }
  int size() const override { return nranks; }
  int kind() const override { return 2; }
};

inline unsigned nblk(long long n, int b) { return (unsigned)((n + b - 1) / b > 0 ? (n + b - 1) / b : 1); }

}  // namespace

// ------------------------------------------------------------------------------------------------ the context
struct shhalo_ctx {
  shpair_ctx* sp = nullptr;
  Transport* tr = nullptr;
  shhalo_geometry geo{};
  HaloGeom hg{};
  double skin = 0.0;
  std::string err;

  // static: remote peers (ascending rank) and the slot tables of the two partitions
  int npeers = 0;
  int peer_rank[26] = {};
  HaloSlots ghost_slots{}, mig_slots{};
  int ghost_peer_of_slot[kHaloMaxSlots] = {};  // index into peer_rank, -1: self

  // the current plan
  shhalo_layout lay{};
  HaloMsgTables tab{};
  int plan_nlocal = -1, nghost = 0;
  DevBuf<int> d_send_idx, d_order;
  DevBuf<unsigned char> d_send_code, d_cat;
  DevBuf<double> d_sendbuf, d_recvbuf, d_rsend, d_rrecv, d_migrows, d_migin;
  DevBuf<int> d_blockcnt, d_start, d_totals, d_msg, d_msgin, d_flags, d_peer_of_slot;
  int* h_ints = nullptr;  // pinned: totals[28] | msgin[26*27] | flags[2]
  shhalo_stats stats{};
  // option "halo_overlap" of the pair context: the forward exchange of a step runs on a stream of its own beside the
  // pair kernels of the slots that touch owned atoms only (made on first use)
  hipStream_t st2x[2] = {nullptr, nullptr};   // the exchange stream of "halo_overlap": [0] ordinary, [1] at the highest stream priority
  bool ev2 = false;
  hipEvent_t ev_ready = nullptr, ev_ghosts = nullptr, ev_bdone = nullptr, ev_rev = nullptr;
};

#define H_FAIL(h, code, ...)                \
  do {                                      \
    char _b[512];                           \
    snprintf(_b, sizeof(_b), __VA_ARGS__);  \
    (h)->err = _b;                          \
    return (code);                          \
  } while (0)
#define H_HIP(h, call)                                                                                  \
  do {                                                                                                  \
    hipError_t _e = (call);                                                                             \
    if (_e != hipSuccess)                                                                               \
      H_FAIL(h, SHPAIR_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define H_RC(h, call)                                        \
  do {                                                       \
    const int _rc = (call);                                  \
    if (_rc) {                                               \
      if ((h)->err.empty()) (h)->err = "internal error";     \
      return _rc;                                            \
    }                                                        \
  } while (0)
#define H_TR(h, call)                          \
  do {                                         \
    const int _rc = (call);                    \
    if (_rc) {                                 \
      (h)->err = (h)->tr->err;                 \
      return _rc;                              \
    }                                          \
  } while (0)
#define H_SP(h, call)                          \
  do {                                         \
    const int _rc = (call);                    \
    if (_rc) {                                 \
      (h)->err = (h)->sp->err;                 \
      return _rc;                              \
    }                                          \
  } while (0)

namespace {

constexpr int kPinTotals = 0, kPinMsgIn = kHaloMaxSlots, kPinFlags = kHaloMaxSlots + 26 * 27, kPinInts = kPinFlags + 4;

// exclusive scan of n ints into out[0..n] (out[n] = total): the three passes of step_kernels.hpp
int scan_ints(shhalo_ctx* h, const int* in, int* out, int n, hipStream_t st)
{
  H_SP(h, shstep_exclusive_scan(h->sp, in, out, n, st));
  return SHPAIR_OK;
}

int peer_index(const shhalo_ctx* h, int rank)
{
  for (int k = 0; k < h->npeers; ++k)
    if (h->peer_rank[k] == rank) return k;
  return -1;
}

// Stable partition of the n owned rows (halo_kernels.hpp): counts per (slot, workgroup), scan, per-slot totals and the
// count messages for the remote peers.  Leaves d_start (positions) for the fill pass.
template <int MODE>
int partition_count(shhalo_ctx* h, const HaloSlots& sl, int n, const double* x, hipStream_t st)
{
  const int nb = (int)nblk(n, kHaloBlock);
  const size_t cells = (size_t)sl.nslots * nb;
  H_HIP(h, h->d_blockcnt.ensure(cells));
  H_HIP(h, h->d_start.ensure(cells + 1));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_count_kernel<MODE>), dim3(nb), dim3(kHaloBlock), 0, st, n, h->hg, sl, x,
                     (const unsigned char*)h->d_cat.p, h->d_blockcnt.p, nb);
  H_HIP(h, hipGetLastError());
  H_RC(h, scan_ints(h, h->d_blockcnt.p, h->d_start.p, (int)cells, st));
  H_HIP(h, hipMemsetAsync(h->d_msg.p, 0, (size_t)26 * 27 * sizeof(int), st));
  hipLaunchKernelGGL(halo_totals_kernel, dim3(1), dim3(64), 0, st, sl, (const int*)h->d_start.p, nb, MODE, h->npeers,
                     (const int*)h->d_peer_of_slot.p, h->d_totals.p, h->d_msg.p);
  H_HIP(h, hipGetLastError());
  return SHPAIR_OK;
}

// the 27-int count vectors travel to / from every remote peer; then totals, the peers' vectors and the error flags
// come to the host in one go
int exchange_counts(shhalo_ctx* h, int nslots, hipStream_t st)
{
  std::vector<Msg> sends, recvs;
  for (int k = 0; k < h->npeers; ++k) {
    sends.push_back({h->peer_rank[k], h->d_msg.p + 27 * k, 27 * sizeof(int)});
    recvs.push_back({h->peer_rank[k], h->d_msgin.p + 27 * k, 27 * sizeof(int)});
  }
  H_TR(h, h->tr->exchange(sends, recvs, st));
  H_HIP(h, hipMemcpyAsync(h->h_ints + kPinTotals, h->d_totals.p, nslots * sizeof(int), hipMemcpyDeviceToHost, st));
  if (h->npeers)
    H_HIP(h, hipMemcpyAsync(h->h_ints + kPinMsgIn, h->d_msgin.p, (size_t)h->npeers * 27 * sizeof(int), hipMemcpyDeviceToHost, st));
  H_HIP(h, hipMemcpyAsync(h->h_ints + kPinFlags, h->d_flags.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  H_HIP(h, hipStreamSynchronize(st));
  return SHPAIR_OK;
}

// The decision to fail is COLLECTIVE wherever a rank-local condition (a lost atom, a capacity that is too small, an
// index outside a table) is found between two exchanges: a rank that simply returned would leave its peers inside
// ncclRecv for rows that are never sent — RCCL has no timeout.  Every rank contributes its code (0 or -SHPAIR_E*) to
// one max-all-reduce and all of them return an error if any did: the failing rank its own code and message, the
// others SHPAIR_ESTATE naming the code.  One rank: nothing to agree on.
int agree(shhalo_ctx* h, int local_rc, hipStream_t st)
{
  if (h->tr->size() <= 1) return local_rc;
  const std::string mine = h->err;
  h->h_ints[kPinFlags + 3] = local_rc ? -local_rc : 0;
  H_HIP(h, hipMemcpyAsync(h->d_flags.p + 3, h->h_ints + kPinFlags + 3, sizeof(int), hipMemcpyHostToDevice, st));
  H_TR(h, h->tr->allreduce_max_i32(h->d_flags.p + 3, 1, st));
  H_HIP(h, hipMemcpyAsync(h->h_ints + kPinFlags + 3, h->d_flags.p + 3, sizeof(int), hipMemcpyDeviceToHost, st));
  H_HIP(h, hipStreamSynchronize(st));
  if (local_rc) {
    h->err = mine;
    return local_rc;
  }
  const int worst = h->h_ints[kPinFlags + 3];
  if (worst > 0)
    H_FAIL(h, SHPAIR_ESTATE, "rank %d stops because another rank failed (%s); its own state was consistent", h->geo.rank,
           shpair_strerror(-worst));
  return SHPAIR_OK;
}

int check_arrays(shhalo_ctx* h, const shhalo_arrays* a)
{
  if (!a) H_FAIL(h, SHPAIR_EINVAL, "null arrays");
  if (a->nlocal < 0 || a->nmax < a->nlocal) H_FAIL(h, SHPAIR_EINVAL, "bad nlocal (%d) / nmax (%d)", a->nlocal, a->nmax);
  if (a->nmax > 0 && (!a->x || !a->v || !a->quat || !a->angmom || !a->f || !a->torque || !a->type || !a->shtype || !a->mask || !a->tag))
    H_FAIL(h, SHPAIR_EINVAL, "null array pointer");
  return SHPAIR_OK;
}

int finish_create(shhalo_ctx* h, shpair_ctx* sp, int rank, const int grid[3], const double lo[3], const double hi[3],
                  const int periodic[3], double skin)
{
  if (!(skin >= 0.0) || !std::isfinite(skin)) H_FAIL(h, SHPAIR_EINVAL, "skin %g must be finite and >= 0", skin);
  double rm = 0.0;
  for (int k = 0; k < sp->nshapes; ++k) {
    if (sp->shapes[k].lmax < 0) H_FAIL(h, SHPAIR_ESTATE, "shape %d is not set (the ghost cutoff needs every bounding radius)", k);
    rm = std::fmax(rm, sp->shapes[k].rmax);
  }
  if (!(rm > 0.0)) H_FAIL(h, SHPAIR_ESTATE, "shapes are not set");
  h->sp = sp;
  h->skin = skin;
  const double cut = 2.0 * rm + skin;
  if (shhalo_plan_geometry(grid, lo, hi, periodic, cut, rank, &h->geo))
    H_FAIL(h, SHPAIR_EINVAL, "grid %dx%dx%d does not fit the box: a decomposed brick edge is shorter than the ghost cutoff %g, an "
           "undecomposed periodic edge shorter than twice that, or the arguments are not finite", grid[0], grid[1], grid[2], cut);
  if (h->geo.nranks != h->tr->size())
    H_FAIL(h, SHPAIR_EINVAL, "grid %dx%dx%d has %d bricks but the transport has %d ranks", grid[0], grid[1], grid[2], h->geo.nranks,
           h->tr->size());
  h->hg = halo_geom_of(h->geo);
  // remote peers, ascending
  std::vector<int> pr;
  for (int c = 0; c < 27; ++c)
    if (h->geo.peer[c] >= 0 && h->geo.peer[c] != rank) pr.push_back(h->geo.peer[c]);
  std::sort(pr.begin(), pr.end());
  pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
  h->npeers = (int)pr.size();
  for (int k = 0; k < h->npeers; ++k) h->peer_rank[k] = pr[k];
  // ghost slots: directions with a peer, by (peer, code) — the order of shhalo_plan_layout
  std::vector<int> dirs;
  for (int c = 0; c < 27; ++c)
    if (c != 13 && h->geo.peer[c] >= 0) dirs.push_back(c);
  std::sort(dirs.begin(), dirs.end(), [&](int a, int b) {
    return h->geo.peer[a] != h->geo.peer[b] ? h->geo.peer[a] < h->geo.peer[b] : a < b;
  });
  h->ghost_slots.nslots = (int)dirs.size();
  for (int s = 0; s < (int)dirs.size(); ++s) {
    h->ghost_slots.code_of_slot[s] = dirs[s];
    h->ghost_peer_of_slot[s] = peer_index(h, h->geo.peer[dirs[s]]);
  }
  // migration categories: 0 stays (also where the direction wraps onto this rank), 1 + k goes to remote peer k
  h->mig_slots.nslots = 1 + h->npeers;
  for (int c = 0; c < 27; ++c) {
    const int p = h->geo.peer[c];
    h->mig_slots.cat_of_code[c] = (c == 13 || p < 0 || p == rank) ? 0 : 1 + peer_index(h, p);
  }
  H_HIP(h, hipSetDevice(sp->device));
  H_HIP(h, h->d_totals.ensure(kHaloMaxSlots));
  H_HIP(h, h->d_msg.ensure(26 * 27));
  H_HIP(h, h->d_msgin.ensure(26 * 27));
  H_HIP(h, h->d_flags.ensure(4));
  H_HIP(h, h->d_peer_of_slot.ensure(kHaloMaxSlots));
  H_HIP(h, hipMemset(h->d_flags.p, 0, 4 * sizeof(int)));
  H_HIP(h, hipMemset(h->d_msgin.p, 0, 26 * 27 * sizeof(int)));
  H_HIP(h, hipMemcpy(h->d_peer_of_slot.p, h->ghost_peer_of_slot, kHaloMaxSlots * sizeof(int), hipMemcpyHostToDevice));
  H_HIP(h, hipHostMalloc((void**)&h->h_ints, kPinInts * sizeof(int)));
  // Neighbor::build of this rank bins its brick plus the ghost shell: a non-periodic box (the periodic images
  // are ghost rows like any other here)
  double blo[3], bhi[3];
  const int nonper[3] = {0, 0, 0};
  for (int d = 0; d < 3; ++d) {
    blo[d] = h->geo.blo[d] - cut;
    bhi[d] = h->geo.bhi[d] + cut;
  }
  H_SP(h, shstep_set_box(sp, blo, bhi, nonper, skin));
  h->stats.nranks_transport = h->tr->size();
  h->stats.transport = h->tr->kind();
  h->stats.rccl_version = h->tr->version();
  return SHPAIR_OK;
}

HaloArrays dev_arrays(const shhalo_arrays* a)
{
  HaloArrays d;
  d.x = a->x; d.v = a->v; d.quat = a->quat; d.angmom = a->angmom;
  d.type = a->type; d.shtype = a->shtype; d.mask = a->mask; d.tag = a->tag;
  return d;
}

}  // namespace

extern "C" {

int shhalo_get_unique_id(unsigned char id[SHHALO_UNIQUE_ID_BYTES])
{
  if (!id) return SHPAIR_EINVAL;
  RcclApi* api = rccl_api();
  if (!api->handle || !api->error.empty()) return SHPAIR_ENODEV;
  static_assert(sizeof(ncclUniqueId) == SHHALO_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return SHPAIR_EHIP;
  std::memcpy(id, &u, sizeof(u));
  return SHPAIR_OK;
}

int shhalo_create_rccl(shhalo_ctx** out, shpair_ctx* sp, const unsigned char id[SHHALO_UNIQUE_ID_BYTES], int rank, int nranks,
                       const int grid[3], const double lo[3], const double hi[3], const int periodic[3], double skin)
{
  if (!out) return SHPAIR_EINVAL;
  *out = nullptr;
  if (!sp || !id || !grid || !lo || !hi || !periodic || rank < 0 || rank >= nranks) return SHPAIR_EINVAL;
  RcclApi* api = rccl_api();
  if (!api->handle || !api->error.empty()) CTX_FAIL(sp, SHPAIR_ENODEV, "%s", api->error.c_str());
  if (hipSetDevice(sp->device) != hipSuccess) CTX_FAIL(sp, SHPAIR_EHIP, "hipSetDevice(%d) failed", sp->device);
  shhalo_ctx* h = new (std::nothrow) shhalo_ctx();
  RcclTransport* t = new (std::nothrow) RcclTransport();
  if (!h || !t) {
    delete h;
    delete t;
    return SHPAIR_ENOMEM;
  }
  t->api = api;
  t->nranks = nranks;
  h->tr = t;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  const ncclResult_t r = api->CommInitRank(&t->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    sp->err = std::string("ncclCommInitRank failed: ") + api->GetErrorString(r);
    t->comm = nullptr;
    shhalo_destroy(h);
    return SHPAIR_EHIP;
  }
  int cnt = 0;
  if (api->CommCount(t->comm, &cnt) == ncclSuccess) t->nranks = cnt;
  const int rc = finish_create(h, sp, rank, grid, lo, hi, periodic, skin);
  if (rc) {
    sp->err = h->err;
    shhalo_destroy(h);
    return rc;
  }
  *out = h;
  return SHPAIR_OK;
}

int shhalo_hub_create(shhalo_hub** out, int nranks)
{
  if (!out || nranks < 1) return SHPAIR_EINVAL;
  shhalo_hub* hub = new (std::nothrow) shhalo_hub();
  if (!hub) return SHPAIR_ENOMEM;
  hub->nranks = nranks;
  hub->box.resize((size_t)nranks * nranks);
  *out = hub;
  return SHPAIR_OK;
}

void shhalo_hub_destroy(shhalo_hub* hub) { delete hub; }

int shhalo_create_staged(shhalo_ctx** out, shpair_ctx* sp, shhalo_exchange_fn exchange, shhalo_allreduce_fn allreduce, void* user,
                         int rank, int nranks, const int grid[3], const double lo[3], const double hi[3], const int periodic[3],
                         double skin)
{
  if (!out) return SHPAIR_EINVAL;
  *out = nullptr;
  if (!sp || !grid || !lo || !hi || !periodic || rank < 0 || rank >= nranks) return SHPAIR_EINVAL;
  if (nranks > 1 && (!exchange || !allreduce)) CTX_FAIL(sp, SHPAIR_EINVAL, "the staged transport needs an exchange and an all-reduce function");
  if (hipSetDevice(sp->device) != hipSuccess) CTX_FAIL(sp, SHPAIR_EHIP, "hipSetDevice(%d) failed", sp->device);
  shhalo_ctx* h = new (std::nothrow) shhalo_ctx();
  StagedTransport* t = new (std::nothrow) StagedTransport();
  if (!h || !t) {
    delete h;
    delete t;
    return SHPAIR_ENOMEM;
  }
  t->xfn = exchange;
  t->rfn = allreduce;
  t->user = user;
  t->rank = rank;
  t->nranks = nranks;
  h->tr = t;
  const int rc = finish_create(h, sp, rank, grid, lo, hi, periodic, skin);
  if (rc) {
    sp->err = h->err;
    shhalo_destroy(h);
    return rc;
  }
  *out = h;
  return SHPAIR_OK;
}

int shhalo_create_local(shhalo_ctx** out, shpair_ctx* sp, shhalo_hub* hub, int rank, int nranks, const int grid[3],
                        const double lo[3], const double hi[3], const int periodic[3], double skin)
{
  if (!out) return SHPAIR_EINVAL;
  *out = nullptr;
  if (!sp || !grid || !lo || !hi || !periodic || rank < 0 || rank >= nranks) return SHPAIR_EINVAL;
  if (nranks > 1 && (!hub || hub->nranks != nranks)) CTX_FAIL(sp, SHPAIR_EINVAL, "a hub created for %d ranks is needed", nranks);
  shhalo_ctx* h = new (std::nothrow) shhalo_ctx();
  LocalTransport* t = new (std::nothrow) LocalTransport();
  if (!h || !t) {
    delete h;
    delete t;
    return SHPAIR_ENOMEM;
  }
  t->hub = nranks > 1 ? hub : nullptr;
  t->rank = rank;
  t->nranks = nranks;
  h->tr = t;
  const int rc = finish_create(h, sp, rank, grid, lo, hi, periodic, skin);
  if (rc) {
    sp->err = h->err;
    shhalo_destroy(h);
    return rc;
  }
  *out = h;
  return SHPAIR_OK;
}

void shhalo_destroy(shhalo_ctx* h)
{
  if (!h) return;
  if (h->sp) (void)hipSetDevice(h->sp->device);
  (void)hipDeviceSynchronize();
  delete h->tr;
  h->d_send_idx.release(); h->d_order.release(); h->d_send_code.release(); h->d_cat.release();
  h->d_sendbuf.release(); h->d_recvbuf.release(); h->d_rsend.release(); h->d_rrecv.release();
  h->d_migrows.release(); h->d_migin.release(); h->d_blockcnt.release(); h->d_start.release();
  h->d_totals.release(); h->d_msg.release(); h->d_msgin.release(); h->d_flags.release(); h->d_peer_of_slot.release();
  if (h->h_ints) (void)hipHostFree(h->h_ints);
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  if (h->ev_ghosts) (void)hipEventDestroy(h->ev_ghosts);
  if (h->ev_bdone) (void)hipEventDestroy(h->ev_bdone);
  if (h->ev_rev) (void)hipEventDestroy(h->ev_rev);
  for (hipStream_t s2 : h->st2x)
    if (s2) (void)hipStreamDestroy(s2);
  delete h;
}

const char* shhalo_last_error(const shhalo_ctx* h) { return h ? h->err.c_str() : "null context"; }

int shhalo_get_geometry(const shhalo_ctx* h, shhalo_geometry* out)
{
  if (!h || !out) return SHPAIR_EINVAL;
  *out = h->geo;
  return SHPAIR_OK;
}

int shhalo_get_stats(const shhalo_ctx* h, shhalo_stats* out)
{
  if (!h || !out) return SHPAIR_EINVAL;
  *out = h->stats;
  return SHPAIR_OK;
}

int shhalo_exchange_device(shhalo_ctx* h, shhalo_arrays* a, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  H_RC(h, check_arrays(h, a));
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  const int n = a->nlocal;
  h->plan_nlocal = -1;  // the send lists refer to the old rows
  H_HIP(h, h->d_cat.ensure((size_t)(n > 0 ? n : 1)));
  if (n > 0) {
    hipLaunchKernelGGL(halo_wrap_dest_kernel, dim3(nblk(n, kHaloBlock)), dim3(kHaloBlock), 0, st, n, h->hg, h->mig_slots, a->x,
                       h->d_cat.p, h->d_flags.p);
    H_HIP(h, hipGetLastError());
  }
  H_RC(h, partition_count<1>(h, h->mig_slots, n, a->x, st));
  H_RC(h, exchange_counts(h, h->mig_slots.nslots, st));
  const int* tot = h->h_ints + kPinTotals;
  const int nstay = tot[0];
  int nleave = 0, narr = 0;
  for (int k = 0; k < h->npeers; ++k) {
    nleave += tot[1 + k];
    narr += h->h_ints[kPinMsgIn + 27 * k];
  }
  // rank-local failures, decided by all ranks together BEFORE the rows travel (agree() above)
  int local_rc = SHPAIR_OK;
  char why[320] = "";
  if (h->h_ints[kPinFlags] & kHaloErrLost) {
    H_HIP(h, hipMemsetAsync(h->d_flags.p, 0, sizeof(int), st));
    local_rc = SHPAIR_ESTATE;
    snprintf(why, sizeof(why), "rank %d: an owned atom left its brick and the 26 neighbouring bricks since the last exchange "
             "(lost atom: the timestep or the skin is too large)", h->geo.rank);
  } else if (nstay + nleave != n) {
    local_rc = SHPAIR_EHIP;
    snprintf(why, sizeof(why), "internal: partition of %d rows gave %d + %d", n, nstay, nleave);
  } else if ((long long)nstay + narr > a->nmax) {
    local_rc = SHPAIR_ENOMEM;
    snprintf(why, sizeof(why), "rank %d: %d owned atoms after migration exceed the capacity nmax = %d", h->geo.rank,
             nstay + narr, a->nmax);
  }
  h->err = why;
  H_RC(h, agree(h, local_rc, st));
  if (nleave == 0 && narr == 0) return SHPAIR_OK;
  const HaloArrays da = dev_arrays(a);
  std::vector<Msg> sends, recvs;
  if (nleave > 0) {
    const int nb = (int)nblk(n, kHaloBlock);
    H_HIP(h, h->d_order.ensure((size_t)n));
    H_HIP(h, h->d_migrows.ensure((size_t)n * kMigWidth));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_fill_kernel<1>), dim3(nb), dim3(kHaloBlock), 0, st, n, h->hg, h->mig_slots,
                       (const double*)a->x, (const unsigned char*)h->d_cat.p, (const int*)h->d_start.p, nb, h->d_order.p,
                       (unsigned char*)nullptr);
    hipLaunchKernelGGL(halo_mig_gather_kernel, dim3(nb), dim3(kHaloBlock), 0, st, n, (const int*)h->d_order.p, da, h->d_migrows.p);
    if (nstay > 0)
      hipLaunchKernelGGL(halo_mig_scatter_kernel, dim3(nblk(nstay, kHaloBlock)), dim3(kHaloBlock), 0, st, nstay, 0,
                         (const double*)h->d_migrows.p, da);
    H_HIP(h, hipGetLastError());
    int off = nstay;
    for (int k = 0; k < h->npeers; ++k) {
      if (tot[1 + k] > 0)
        sends.push_back({h->peer_rank[k], h->d_migrows.p + (size_t)off * kMigWidth, (size_t)tot[1 + k] * kMigWidth * sizeof(double)});
      off += tot[1 + k];
    }
  }
  if (narr > 0) {
    H_HIP(h, h->d_migin.ensure((size_t)narr * kMigWidth));
    int off = 0;
    for (int k = 0; k < h->npeers; ++k) {
      const int c = h->h_ints[kPinMsgIn + 27 * k];
      if (c > 0) recvs.push_back({h->peer_rank[k], h->d_migin.p + (size_t)off * kMigWidth, (size_t)c * kMigWidth * sizeof(double)});
      off += c;
    }
  }
  H_TR(h, h->tr->exchange(sends, recvs, st));
  if (narr > 0) {
    hipLaunchKernelGGL(halo_mig_scatter_kernel, dim3(nblk(narr, kHaloBlock)), dim3(kHaloBlock), 0, st, narr, nstay,
                       (const double*)h->d_migin.p, da);
    H_HIP(h, hipGetLastError());
  }
  a->nlocal = nstay + narr;
  h->stats.migrated_out += nleave;
  h->stats.migrated_in += narr;
  return SHPAIR_OK;
}

int shhalo_borders_device(shhalo_ctx* h, const shhalo_arrays* a, int* nghost, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  if (nghost) *nghost = 0;
  H_RC(h, check_arrays(h, a));
  if (!nghost) H_FAIL(h, SHPAIR_EINVAL, "null nghost");
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  const int n = a->nlocal;
  h->plan_nlocal = -1;
  h->nghost = 0;
  H_HIP(h, h->d_cat.ensure(1));
  H_RC(h, partition_count<0>(h, h->ghost_slots, n, a->x, st));
  H_RC(h, exchange_counts(h, h->ghost_slots.nslots, st));
  int send_cnt[27] = {0}, recv_cnt[27] = {0};
  for (int s = 0; s < h->ghost_slots.nslots; ++s) send_cnt[h->ghost_slots.code_of_slot[s]] = h->h_ints[kPinTotals + s];
  for (int c = 0; c < 27; ++c) {
    const int p = h->geo.peer[c];
    if (c == 13 || p < 0) continue;
    // what arrives through my direction c was sent with the sender's code 26 - c
    recv_cnt[c] = (p == h->geo.rank) ? send_cnt[26 - c] : h->h_ints[kPinMsgIn + 27 * peer_index(h, p) + (26 - c)];
  }
  if (shhalo_plan_layout(&h->geo, send_cnt, recv_cnt, &h->lay)) H_FAIL(h, SHPAIR_EINVAL, "internal: message layout");
  const shhalo_layout& L = h->lay;
  *nghost = L.nghost;
  {
    int local_rc = SHPAIR_OK;
    char why[256] = "";
    if ((long long)n + L.nghost > a->nmax) {
      local_rc = SHPAIR_ENOMEM;
      snprintf(why, sizeof(why), "rank %d: %d owned + %d ghost rows exceed the capacity nmax = %d", h->geo.rank, n, L.nghost,
               a->nmax);
    }
    h->err = why;
    H_RC(h, agree(h, local_rc, st));   // before the ghost rows travel: every rank returns, or none
  }
  H_HIP(h, h->d_send_idx.ensure((size_t)(L.nsend > 0 ? L.nsend : 1)));
  H_HIP(h, h->d_send_code.ensure((size_t)(L.nsend > 0 ? L.nsend : 1)));
  H_HIP(h, h->d_sendbuf.ensure((size_t)(L.nsend > 0 ? L.nsend : 1) * kBorderWidth));
  H_HIP(h, h->d_recvbuf.ensure((size_t)(L.nghost > 0 ? L.nghost : 1) * kBorderWidth));
  H_HIP(h, h->d_rsend.ensure((size_t)(L.nghost > 0 ? L.nghost : 1) * kRevWidth));
  H_HIP(h, h->d_rrecv.ensure((size_t)(L.nsend > 0 ? L.nsend : 1) * kRevWidth));
  for (int c = 0; c < 27; ++c) {
    for (int d = 0; d < 3; ++d) h->tab.shift[c][d] = h->geo.shift[c][d];
    h->tab.self[c] = (h->geo.peer[c] == h->geo.rank) ? 1 : 0;
    h->tab.send_off[c] = L.send_off[c];
    h->tab.recv_off[c] = L.recv_off[c];
    h->tab.recv_cnt[c] = L.recv_cnt[c];
  }
  if (n > 0 && L.nsend > 0) {
    const int nb = (int)nblk(n, kHaloBlock);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_fill_kernel<0>), dim3(nb), dim3(kHaloBlock), 0, st, n, h->hg, h->ghost_slots,
                       (const double*)a->x, (const unsigned char*)nullptr, (const int*)h->d_start.p, nb, h->d_send_idx.p,
                       h->d_send_code.p);
    H_HIP(h, hipGetLastError());
  }
  h->plan_nlocal = n;
  h->nghost = L.nghost;
  h->stats.npeers = L.npeers;
  h->stats.nsend_rows = L.nsend;
  h->stats.nghost_rows = L.nghost;
  long long fb = 0, rb = 0;
  for (int k = 0; k < L.npeers; ++k) {
    fb += (long long)L.peer_send_cnt[k] * kFwdWidth * 8;
    rb += (long long)L.peer_recv_cnt[k] * kRevWidth * 8;
  }
  h->stats.forward_bytes_per_step = fb;
  h->stats.reverse_bytes_per_step = rb;
  ++h->stats.rebuilds;
  // the ghost rows: positions, orientations and the per-atom constants in one wider message
  if (L.nsend > 0) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_pack_kernel<kBorderWidth>), dim3(nblk(L.nsend, kHaloBlock)), dim3(kHaloBlock), 0, st,
                       L.nsend, h->tab, (const int*)h->d_send_idx.p, (const unsigned char*)h->d_send_code.p, (const double*)a->x,
                       (const double*)a->quat, (const int*)a->tag, (const int*)a->type, (const int*)a->shtype, h->d_sendbuf.p,
                       h->d_recvbuf.p);
    H_HIP(h, hipGetLastError());
  }
  std::vector<Msg> sends, recvs;
  for (int k = 0; k < L.npeers; ++k) {
    if (L.peer_send_cnt[k] > 0)
      sends.push_back({L.peer_rank[k], h->d_sendbuf.p + (size_t)L.peer_send_off[k] * kBorderWidth,
                       (size_t)L.peer_send_cnt[k] * kBorderWidth * sizeof(double)});
    if (L.peer_recv_cnt[k] > 0)
      recvs.push_back({L.peer_rank[k], h->d_recvbuf.p + (size_t)L.peer_recv_off[k] * kBorderWidth,
                       (size_t)L.peer_recv_cnt[k] * kBorderWidth * sizeof(double)});
  }
  H_TR(h, h->tr->exchange(sends, recvs, st));
  if (L.nghost > 0) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_unpack_kernel<kBorderWidth>), dim3(nblk(L.nghost, kHaloBlock)), dim3(kHaloBlock), 0, st,
                       L.nghost, n, (const double*)h->d_recvbuf.p, a->x, a->quat, a->tag, a->type, a->shtype);
    H_HIP(h, hipGetLastError());
  }
  return SHPAIR_OK;
}

// Neighbor::build over the brick plus its ghost shell.  The list build reports shape indices outside the table (they
// may have arrived with migrated atoms or ghost rows): a rank-local failure between two exchanges, so the ranks agree
// on it before anyone posts the forward exchange.
int shhalo_neighbor_build_device(shhalo_ctx* h, const shhalo_arrays* a, int nghost, int* npairs, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  H_RC(h, check_arrays(h, a));
  if (!npairs || nghost < 0) H_FAIL(h, SHPAIR_EINVAL, "null npairs or negative nghost");
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  const int lrc = shstep_neighbor_build_device(h->sp, a->nlocal, nghost, a->x, a->shtype, a->tag, npairs, st);
  if (lrc) h->err = h->sp->err;
  else h->err.clear();
  return agree(h, lrc, st);
}

int shhalo_forward_device(shhalo_ctx* h, double* x, double* quat, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  if (h->plan_nlocal < 0) H_FAIL(h, SHPAIR_ESTATE, "forward: no plan (shhalo_borders_device first)");
  const shhalo_layout& L = h->lay;
  if (L.nsend == 0 && L.nghost == 0) return SHPAIR_OK;
  if (!x || !quat) H_FAIL(h, SHPAIR_EINVAL, "null array pointer");
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  if (L.nsend > 0) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_pack_kernel<kFwdWidth>), dim3(nblk(L.nsend, kHaloBlock)), dim3(kHaloBlock), 0, st, L.nsend,
                       h->tab, (const int*)h->d_send_idx.p, (const unsigned char*)h->d_send_code.p, (const double*)x,
                       (const double*)quat, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, h->d_sendbuf.p,
                       h->d_recvbuf.p);
    H_HIP(h, hipGetLastError());
  }
  if (L.npeers > 0) {
    std::vector<Msg> sends, recvs;
    for (int k = 0; k < L.npeers; ++k) {
      if (L.peer_send_cnt[k] > 0)
        sends.push_back({L.peer_rank[k], h->d_sendbuf.p + (size_t)L.peer_send_off[k] * kFwdWidth,
                         (size_t)L.peer_send_cnt[k] * kFwdWidth * sizeof(double)});
      if (L.peer_recv_cnt[k] > 0)
        recvs.push_back({L.peer_rank[k], h->d_recvbuf.p + (size_t)L.peer_recv_off[k] * kFwdWidth,
                         (size_t)L.peer_recv_cnt[k] * kFwdWidth * sizeof(double)});
    }
    H_TR(h, h->tr->exchange(sends, recvs, st));
  }
  if (L.nghost > 0) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(halo_unpack_kernel<kFwdWidth>), dim3(nblk(L.nghost, kHaloBlock)), dim3(kHaloBlock), 0, st,
                       L.nghost, h->plan_nlocal, (const double*)h->d_recvbuf.p, x, quat, (int*)nullptr, (int*)nullptr, (int*)nullptr);
    H_HIP(h, hipGetLastError());
  }
  return SHPAIR_OK;
}

int shhalo_reverse_device(shhalo_ctx* h, double* f, double* torque, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  if (h->plan_nlocal < 0) H_FAIL(h, SHPAIR_ESTATE, "reverse: no plan (shhalo_borders_device first)");
  const shhalo_layout& L = h->lay;
  if (L.nsend == 0 && L.nghost == 0) return SHPAIR_OK;
  if (!f || !torque) H_FAIL(h, SHPAIR_EINVAL, "null array pointer");
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  if (L.nghost > 0) {
    hipLaunchKernelGGL(halo_rpack_kernel, dim3(nblk(L.nghost, kHaloBlock)), dim3(kHaloBlock), 0, st, L.nghost, h->plan_nlocal, h->tab,
                       (const double*)f, (const double*)torque, h->d_rsend.p, h->d_rrecv.p);
    H_HIP(h, hipGetLastError());
  }
  if (L.npeers > 0) {
    std::vector<Msg> sends, recvs;
    for (int k = 0; k < L.npeers; ++k) {
      // what came in through the forward exchange goes back to where it came from
      if (L.peer_recv_cnt[k] > 0)
        sends.push_back({L.peer_rank[k], h->d_rsend.p + (size_t)L.peer_recv_off[k] * kRevWidth,
                         (size_t)L.peer_recv_cnt[k] * kRevWidth * sizeof(double)});
      if (L.peer_send_cnt[k] > 0)
        recvs.push_back({L.peer_rank[k], h->d_rrecv.p + (size_t)L.peer_send_off[k] * kRevWidth,
                         (size_t)L.peer_send_cnt[k] * kRevWidth * sizeof(double)});
    }
    H_TR(h, h->tr->exchange(sends, recvs, st));
  }
  if (L.nsend > 0) {
    if (h->sp->opt_deterministic) {
      // bitwise reproducible sums: one launch per direction (unique owners inside a block, plain adds), in code order
      for (int c = 0; c < 27; ++c)
        if (L.send_cnt[c] > 0)
          hipLaunchKernelGGL(halo_runpack_block_kernel, dim3(nblk(L.send_cnt[c], kHaloBlock)), dim3(kHaloBlock), 0, st,
                             L.send_cnt[c], L.send_off[c], (const int*)h->d_send_idx.p, (const double*)h->d_rrecv.p, f, torque);
    } else {
      hipLaunchKernelGGL(halo_runpack_kernel, dim3(nblk(L.nsend, kHaloBlock)), dim3(kHaloBlock), 0, st, L.nsend,
                         (const int*)h->d_send_idx.p, (const double*)h->d_rrecv.p, f, torque);
    }
    H_HIP(h, hipGetLastError());
  }
  return SHPAIR_OK;
}

int shhalo_check_rebuild_device(shhalo_ctx* h, int nlocal, const double* x, int* rebuild, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  if (!rebuild) H_FAIL(h, SHPAIR_EINVAL, "null rebuild pointer");
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  int* flag = nullptr;
  int forced = 0;
  H_SP(h, shstep_enqueue_check(h->sp, nlocal, x, &flag, &forced, st));
  hipLaunchKernelGGL(halo_flag_merge_kernel, dim3(1), dim3(64), 0, st, (const int*)flag, h->d_flags.p + 2);
  H_HIP(h, hipGetLastError());
  if (forced) H_HIP(h, hipMemsetAsync(h->d_flags.p + 2, 0xff, 1, st));  // low byte set: > 0
  H_TR(h, h->tr->allreduce_max_i32(h->d_flags.p + 2, 1, st));
  H_HIP(h, hipMemcpyAsync(h->h_ints + kPinFlags + 2, h->d_flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, st));
  H_HIP(h, hipStreamSynchronize(st));
  *rebuild = h->h_ints[kPinFlags + 2] > 0 ? 1 : 0;
  return SHPAIR_OK;
}

int shhalo_allreduce_sum_device(shhalo_ctx* h, double* data, int n, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  if (n < 0 || (n > 0 && !data)) H_FAIL(h, SHPAIR_EINVAL, "bad all-reduce arguments");
  if (n == 0) return SHPAIR_OK;
  H_HIP(h, hipSetDevice(h->sp->device));
  H_TR(h, h->tr->allreduce_sum_f64(data, n, (hipStream_t)stream));
  return SHPAIR_OK;
}

int shhalo_transport_selftest(shhalo_ctx* h, int nbytes, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  if (nbytes <= 0 || nbytes > (1 << 28)) H_FAIL(h, SHPAIR_EINVAL, "self-test size %d", nbytes);
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  DevBuf<unsigned char> src, dst;
  DevBuf<double> red;
  DevBuf<int> redi;
  H_HIP(h, src.ensure((size_t)nbytes));
  H_HIP(h, dst.ensure((size_t)nbytes));
  H_HIP(h, red.ensure(3));
  H_HIP(h, redi.ensure(2));
  std::vector<unsigned char> pat((size_t)nbytes), back((size_t)nbytes, 0);
  for (int k = 0; k < nbytes; ++k) pat[k] = (unsigned char)((k * 131 + 7 + 17 * h->geo.rank) & 0xff);
  H_HIP(h, hipMemcpyAsync(src.p, pat.data(), (size_t)nbytes, hipMemcpyHostToDevice, st));
  H_HIP(h, hipMemsetAsync(dst.p, 0, (size_t)nbytes, st));
  const int me = h->geo.rank;
  std::vector<Msg> sends{{me, src.p, (size_t)nbytes}}, recvs{{me, dst.p, (size_t)nbytes}};
  // (the host-staged transport's caller may not be able to send to itself — gloo cannot — and a hub-less local transport
  // has nobody to talk to: there only the all-reduce is exercised)
  const bool p2p = h->tr->kind() == 1 || (h->tr->kind() == 0 && h->tr->size() > 1);
  if (p2p) {
    H_TR(h, h->tr->exchange(sends, recvs, st));
    H_HIP(h, hipMemcpyAsync(back.data(), dst.p, (size_t)nbytes, hipMemcpyDeviceToHost, st));
  } else {
    back = pat;
  }
  const int n = h->tr->size();
  const double dv[3] = {1.0, 0.5 * (me + 1), -2.0};
  const int iv[2] = {me + 1, -me};
  H_HIP(h, hipMemcpyAsync(red.p, dv, sizeof(dv), hipMemcpyHostToDevice, st));
  H_HIP(h, hipMemcpyAsync(redi.p, iv, sizeof(iv), hipMemcpyHostToDevice, st));
  H_TR(h, h->tr->allreduce_sum_f64(red.p, 3, st));
  H_TR(h, h->tr->allreduce_max_i32(redi.p, 2, st));
  double dr[3];
  int ir[2];
  H_HIP(h, hipMemcpyAsync(dr, red.p, sizeof(dr), hipMemcpyDeviceToHost, st));
  H_HIP(h, hipMemcpyAsync(ir, redi.p, sizeof(ir), hipMemcpyDeviceToHost, st));
  H_HIP(h, hipStreamSynchronize(st));
  for (int k = 0; k < nbytes; ++k)
    if (back[k] != pat[k]) H_FAIL(h, SHPAIR_ESTATE, "transport self-test: byte %d came back as %d, sent %d", k, (int)back[k], (int)pat[k]);
  if (dr[0] != (double)n || dr[1] != 0.25 * n * (n + 1) || dr[2] != -2.0 * n || ir[0] != n || ir[1] != 0)
    H_FAIL(h, SHPAIR_ESTATE, "transport self-test: all-reduce over %d rank(s) gave sum (%g, %g, %g), max (%d, %d)", n, dr[0], dr[1], dr[2],
           ir[0], ir[1]);
  return SHPAIR_OK;
}

int shhalo_run_device(shhalo_ctx* h, shhalo_arrays* a, const shhalo_run_params* p, int nsteps, int* nghost_io, int* rebuilds,
                      double* kernel_ms, void* stream)
{
  if (!h) return SHPAIR_EINVAL;
  if (rebuilds) *rebuilds = 0;
  if (kernel_ms) *kernel_ms = 0.0;
  H_RC(h, check_arrays(h, a));
  if (!p || !nghost_io || nsteps < 0) H_FAIL(h, SHPAIR_EINVAL, "null arguments or nsteps < 0");
  if (p->check_every < 1 || !std::isfinite(p->dt)) H_FAIL(h, SHPAIR_EINVAL, "bad check_every (%d) / dt", p->check_every);
  if (p->eflag_last && !p->ev_dev) H_FAIL(h, SHPAIR_EINVAL, "eflag_last set but ev_dev is null");
  if (h->plan_nlocal != a->nlocal || *nghost_io != h->nghost || !h->sp->have_neighbors)
    H_FAIL(h, SHPAIR_ESTATE, "run: the plan, ghosts and neighbour list of the current atoms must be built first "
           "(shhalo_exchange_device + shhalo_borders_device + shstep_neighbor_build_device)");
  H_HIP(h, hipSetDevice(h->sp->device));
  hipStream_t st = (hipStream_t)stream;
  shpair_ctx* sp = h->sp;
  // "halo_overlap": the exchange stream.  Two kinds, made on first use, chosen per call by "halo_stream_priority":
  //  [0] an ordinary non-blocking stream.  HIP maps a process's streams round robin onto a few hardware queues, and one
  //      created as the fifth or later of the process (torch's, the context's two, RCCL's own come first) shares a queue
  //      with one of them — if that is the compute stream the exchange runs behind the pair kernels it is meant to run
  //      beside (measured for the host-pointer path's upload stream: +0.11 ms per call when it shared,
  //      tools/host_path_probe.py);
  //  [1] a stream at the highest stream priority: a priority level of its own is a queue of its own, and the pack /
  //      RCCL / unpack kernels — a few workgroups, latency-critical — are dispatched ahead of the pair kernels' backlog.
  //  Which is better between GPUs is unmeasured here (one GPU per box).  In the rehearsal of 8 rank threads on ONE GPU
  //  [1] costs 5 % (26.0 against 24.8 ms per timestep; [0]: 24.5 against 24.6 without overlap,
  //  profiles/r05_g_local8_priority.txt) — there every rank's high-priority kernels pre-empt every other rank's pair
  //  kernels — so [0] is the default and bench.py --gpus N times both in the run itself.
  hipStream_t st2 = nullptr;
  if (sp->opt_overlap) {
    const int kind = sp->opt_halo_prio ? 1 : 0;
    if (!h->st2x[kind]) {
      if (kind == 1) {
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) {
          (void)hipGetLastError();
          prio_least = prio_greatest = 0;
        }
        if (hipStreamCreateWithPriority(&h->st2x[1], hipStreamNonBlocking, prio_greatest) != hipSuccess) {
          (void)hipGetLastError();
          h->st2x[1] = nullptr;
        }
      }
      if (!h->st2x[kind]) H_HIP(h, hipStreamCreateWithFlags(&h->st2x[kind], hipStreamNonBlocking));
    }
    st2 = h->st2x[kind];
    if (!h->ev2) {
      H_HIP(h, hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
      H_HIP(h, hipEventCreateWithFlags(&h->ev_ghosts, hipEventDisableTiming));
      H_HIP(h, hipEventCreateWithFlags(&h->ev_bdone, hipEventDisableTiming));
      H_HIP(h, hipEventCreateWithFlags(&h->ev_rev, hipEventDisableTiming));
      h->ev2 = true;
    }
  }
  const bool body = p->gravity[0] != 0.0 || p->gravity[1] != 0.0 || p->gravity[2] != 0.0 || p->gamma_t != 0.0 || p->gamma_r != 0.0;
  int nghost = *nghost_io, nreb = 0;
  // pair-kernel time: one event pair around EACH slot range of a step — up to three with "halo_overlap" — so that the
  // waits for the exchange's events between the ranges are not counted as kernel time (bounded pool; beyond it the
  // steps are not timed)
  constexpr int kEvPerStep = 6;
  const int ntimed = kernel_ms ? (nsteps < 2048 ? nsteps : 2048) : 0;
  std::vector<hipEvent_t> ev((size_t)kEvPerStep * ntimed, nullptr);
  std::vector<unsigned char> ev_used((size_t)(kEvPerStep / 2) * ntimed, 0);
  for (auto& e : ev) H_HIP(h, hipEventCreate(&e));
  // range k (0, 1, 2) of `step`: record before / after on the caller's stream
  const auto tick = [&](const int step, const int k, const int end) {
    if (step >= ntimed) return;
    (void)hipEventRecord(ev[(size_t)kEvPerStep * step + 2 * k + end], st);
    if (end) ev_used[(size_t)(kEvPerStep / 2) * step + k] = 1;
  };
  int rc = SHPAIR_OK;
  for (int step = 0; step < nsteps && rc == SHPAIR_OK; ++step) {
    rc = shstep_nve_device(sp, 0, a->nlocal, p->dt, a->x, a->v, a->quat, a->angmom, a->f, a->torque, a->shtype, a->mask,
                           p->groupbit, st);
    if (rc) { h->err = sp->err; break; }
    if ((step + 1) % p->check_every == 0) {
      int rebuild = 0;
      rc = shhalo_check_rebuild_device(h, a->nlocal, a->x, &rebuild, st);
      if (rc) break;
      if (rebuild) {
        int np = 0;
        rc = shhalo_exchange_device(h, a, st);
        if (!rc) rc = shhalo_borders_device(h, a, &nghost, st);
        if (!rc) {
          // the list build reports shape indices outside the table (they may have arrived with migrated atoms): a
          // rank-local failure in the middle of the step, so the ranks agree on it before the forward exchange
          rc = shhalo_neighbor_build_device(h, a, nghost, &np, st);
        }
        if (rc) break;
        ++nreb;
      }
    }
    // "halo_overlap": the forward exchange (pack, ncclSend / ncclRecv per peer, unpack) goes to a second stream behind
    // this step's positions, and the pair kernels of the slots whose atoms are all owned — the front segment of the
    // partitioned list, down to a multiple of 32 slots — run beside it; the slots with a ghost wait for the exchange.
    const bool overlap = sp->opt_overlap && st2 != nullptr;
    hipStream_t sf = overlap ? st2 : st;
    if (overlap) {
      if (hipEventRecord(h->ev_ready, st) != hipSuccess || hipStreamWaitEvent(st2, h->ev_ready, 0) != hipSuccess) {
        h->err = "hipEventRecord / hipStreamWaitEvent failed (halo_overlap)";
        rc = SHPAIR_EHIP;
        break;
      }
    }
    rc = shhalo_forward_device(h, a->x, a->quat, sf);
    if (rc) break;
    if (overlap && hipEventRecord(h->ev_ghosts, st2) != hipSuccess) {
      h->err = "hipEventRecord failed (halo_overlap)";
      rc = SHPAIR_EHIP;
      break;
    }
    const size_t nall = (size_t)a->nlocal + nghost;
    rc = shstep_force_clear_device(sp, (int)nall, a->f, a->torque, st);   // one launch (two memsets are four fill kernels)
    if (rc) { h->err = sp->err; break; }
    const int ef = (p->eflag_last && step == nsteps - 1) ? 1 : 0;
    // "halo_overlap" 2 (atomic accumulation only): the REVERSE exchange is hidden too — the owned-only slots are cut in
    // two, [0, a) runs beside the forward exchange, the ghost slots follow it, and [a, split) runs beside the reverse
    // exchange, whose unpack adds into the owners' rows with the same FP64 atomics the pair kernels use.  (The
    // deterministic mode adds in a fixed order with plain stores: there the reverse exchange stays behind the kernels.)
    const bool overlap_rev = overlap && sp->opt_overlap >= 2 && !sp->opt_deterministic;
    bool reverse_done = false;
    if (overlap) {
      const int split = (sp->n_interior < sp->npairs ? sp->n_interior : sp->npairs) & ~31;   // never beyond the installed list
      const int cut = overlap_rev ? ((split / 2) & ~31) : split;   // [0, cut) beside the forward exchange
      tick(step, 0, 0);
      rc = shp_compute_range(sp, a->nlocal, nghost, a->x, a->quat, a->type, a->shtype, 1, ef, ef, a->f, a->torque,
                             ef ? p->ev_dev : nullptr, st, 0, cut, kPartPre);
      tick(step, 0, 1);
      if (!rc && hipStreamWaitEvent(st, h->ev_ghosts, 0) != hipSuccess) {
        h->err = "hipStreamWaitEvent failed (halo_overlap)";
        rc = SHPAIR_EHIP;
        break;
      }
      if (!rc) {
        tick(step, 1, 0);
        rc = shp_compute_range(sp, a->nlocal, nghost, a->x, a->quat, a->type, a->shtype, 1, ef, ef, a->f, a->torque,
                               ef ? p->ev_dev : nullptr, st, split, sp->npairs, overlap_rev ? 0 : kPartPost);
        tick(step, 1, 1);
      }
      if (!rc && overlap_rev) {
        // every contribution to a ghost row is in: the reverse exchange starts on the second stream ...
        if (hipEventRecord(h->ev_bdone, st) != hipSuccess || hipStreamWaitEvent(st2, h->ev_bdone, 0) != hipSuccess) {
          h->err = "hipEventRecord / hipStreamWaitEvent failed (halo_overlap 2)";
          rc = SHPAIR_EHIP;
          break;
        }
        rc = shhalo_reverse_device(h, a->f, a->torque, st2);
        if (rc) break;
        if (hipEventRecord(h->ev_rev, st2) != hipSuccess) {
          h->err = "hipEventRecord failed (halo_overlap 2)";
          rc = SHPAIR_EHIP;
          break;
        }
        // ... beside the second half of the owned-only slots
        tick(step, 2, 0);
        rc = shp_compute_range(sp, a->nlocal, nghost, a->x, a->quat, a->type, a->shtype, 1, ef, ef, a->f, a->torque,
                               ef ? p->ev_dev : nullptr, st, cut, split, kPartPost);
        tick(step, 2, 1);
        if (!rc && hipStreamWaitEvent(st, h->ev_rev, 0) != hipSuccess) {
          h->err = "hipStreamWaitEvent failed (halo_overlap 2)";
          rc = SHPAIR_EHIP;
          break;
        }
        reverse_done = true;
      }
    } else {
      tick(step, 0, 0);
      rc = shpair_compute_device(sp, a->nlocal, nghost, a->x, a->quat, a->type, a->shtype, 1, ef, ef, a->f, a->torque,
                                 ef ? p->ev_dev : nullptr, st);
      tick(step, 0, 1);
    }
    if (rc) { h->err = sp->err; break; }
    if (!reverse_done) {
      rc = shhalo_reverse_device(h, a->f, a->torque, st);
      if (rc) break;
    }
    if (body) {
      rc = shstep_post_force_device(sp, a->nlocal, p->gravity, p->gamma_t, p->gamma_r, a->v, a->quat, a->angmom, a->shtype, a->mask,
                                    p->groupbit, a->f, a->torque, st);
      if (rc) { h->err = sp->err; break; }
    }
    rc = shstep_nve_device(sp, 1, a->nlocal, p->dt, a->x, a->v, a->quat, a->angmom, a->f, a->torque, a->shtype, a->mask,
                           p->groupbit, st);
    if (rc) { h->err = sp->err; break; }
  }
  const hipError_t es = hipStreamSynchronize(st);
  // a step that ended early may have left an exchange in flight on the second stream, reading the caller's arrays
  if (rc != SHPAIR_OK && st2) (void)hipStreamSynchronize(st2);
  if (rc == SHPAIR_OK && es == hipSuccess && kernel_ms) {
    double sum = 0.0;
    for (int k = 0; k < ntimed; ++k)
      for (int r = 0; r < kEvPerStep / 2; ++r) {
        float ms = 0.f;
        if (ev_used[(size_t)(kEvPerStep / 2) * k + r] &&
            hipEventElapsedTime(&ms, ev[(size_t)kEvPerStep * k + 2 * r], ev[(size_t)kEvPerStep * k + 2 * r + 1]) == hipSuccess)
          sum += ms;
      }
    *kernel_ms = sum;
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  *nghost_io = nghost;
  if (rebuilds) *rebuilds = nreb;
  if (rc) return rc;
  if (es != hipSuccess) H_FAIL(h, SHPAIR_EHIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
  // the kernels' error bits (a type or shape index outside its table — such rows arrive from other ranks packed
  // into 64-bit words — makes a kernel skip the pair / particle and raise a bit instead of reading out of bounds):
  // read once per call, and agreed on by all ranks like the failures of a reneighbouring
  int local_rc = shpair_check_device_errors(sp, st);
  if (local_rc) h->err = sp->err;
  else h->err.clear();
  H_RC(h, agree(h, local_rc, st));
  return SHPAIR_OK;
}

}  // extern "C"
