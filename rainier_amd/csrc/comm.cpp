// comm.cpp -- the ONE collective of the multi-GPU path, behind the C ABI: an RCCL all-gather of the device-resident draws
// over xGMI (include/rainier_hip.h, "multi-process multi-GPU").
//
// One process per GPU (the launch the reference's users get from `torch.distributed.run`, MPI or a JVM per device): every
// rank samples its shard of the chains (seeds by GLOBAL chain id, no data-path collective: chains never interact,
// sampler/Driver.scala:13-17) and the draws are gathered once at the end.  RCCL is reached with dlopen, so that a
// single-GPU deployment does not need librccl at all; the 128-byte unique id is created by rank 0 and handed to the other
// ranks by whatever bootstrap the host program has (bench.py: a gloo broadcast; a JVM: its own RPC).
#include <hip/hip_runtime.h>

#include <cstring>
#include <atomic>
#include <chrono>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <string>

#include <dlfcn.h>

#include "../../include/rainier_hip.h"

extern "C" int rh_sampler_draws_device(rh_sampler *s, void **p);
// engine.cpp: what a gather needs to know about a sampler handle
extern "C" int rh_sampler_geometry_(rh_sampler *s, int *device, void **stream, int64_t *doubles);
extern "C" void rh_set_thread_error_(const char *msg);

namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm;
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(nccl_uid *) = nullptr;
  int (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*CommAbort)(nccl_comm) = nullptr;   // optional
  int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
  bool load() {
    if (h) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) { err = std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror(); return false; }
#define SYM(field, name) field = (decltype(field))dlsym(h, name); if (!field) { err = std::string("RCCL symbol missing: ") + name; return false; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather") SYM(AllReduce, "ncclAllReduce") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    CommAbort = (decltype(CommAbort))dlsym(h, "ncclCommAbort");
    return true;
  }
};
Rccl g_rccl;
std::mutex g_mu;
enum { kNcclFloat64 = 8, kNcclMax = 2 };

int fail(int code, const std::string &msg) { rh_set_thread_error_(msg.c_str()); return code; }
// A rank that never arrives (its rh_model_create failed, its process died) must not hang the others for ever: communicator
// set-up and every collective are bounded by RH_COMM_TIMEOUT_S seconds (default 180) and fail with RH_E_DEVICE.
double comm_timeout_s() {
  if (const char *e = std::getenv("RH_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) return v; }
  return 180.0;
}
// hipStreamSynchronize with a deadline (the collective's kernel cannot be recalled; the caller reports and gives up)
bool wait_stream(hipStream_t st, const char *what, std::string &err) {
  const auto t0 = std::chrono::steady_clock::now();
  const double limit = comm_timeout_s();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) { err = std::string(what) + ": " + hipGetErrorString(q); return false; }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
      err = std::string(what) + ": no completion within " + std::to_string((int)limit) + " s (RH_COMM_TIMEOUT_S) -- a rank is missing or stuck";
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
int nccl_fail(const char *what, int rc) {
  return fail(RH_E_DEVICE, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
}
}  // namespace

struct rh_comm {
  nccl_comm comm = nullptr;
  int world = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  void *d_gather = nullptr; size_t gather_bytes = 0;
  void *d_scalar = nullptr;
  // a collective that missed its deadline is still enqueued on `stream` and may still touch d_gather / d_scalar: the communicator
  // refuses every further call, and rh_comm_destroy aborts it (ncclCommAbort) and LEAVES the buffers and the stream alone -- the
  // process is expected to report the failure and exit
  bool poisoned = false;
};

extern "C" int rh_comm_unique_id(unsigned char id[RH_COMM_ID_BYTES]) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!id) return fail(RH_E_INVALID, "rh_comm_unique_id: NULL");
  if (!g_rccl.load()) return fail(RH_E_UNSUPPORTED, g_rccl.err);
  // (no device: say so before RCCL is touched -- its bootstrap would fail the same way, and leave a "[FATAL ERROR]" line on stderr
  //  when the process exits)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RH_E_DEVICE, "rh_comm_unique_id: no HIP device");
  nccl_uid u;
  const int rc = g_rccl.GetUniqueId(&u);
  if (rc) return nccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id, u.internal, RH_COMM_ID_BYTES);
  return RH_OK;
}

extern "C" int rh_comm_create(const unsigned char id[RH_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device, rh_comm **out) {
  if (!out) return fail(RH_E_INVALID, "rh_comm_create: out is NULL");
  *out = nullptr;
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(RH_E_INVALID, "rh_comm_create: bad arguments");
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rccl.load()) return fail(RH_E_UNSUPPORTED, g_rccl.err);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RH_E_DEVICE, "rh_comm_create: no HIP device");
  if (device < 0 || device >= ndev) return fail(RH_E_INVALID, "rh_comm_create: device ordinal out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(RH_E_DEVICE, "hipSetDevice failed");
  rh_comm *c = new rh_comm();
  c->world = world; c->rank = rank; c->device = device;
  nccl_uid u;
  std::memcpy(u.internal, id, RH_COMM_ID_BYTES);
  // ncclCommInitRank blocks until all `world` ranks have called it: run it on a helper thread and give up after the deadline
  // (the helper then stays blocked inside RCCL; the caller is about to report the failure and exit)
  struct Init { nccl_comm comm = nullptr; std::promise<int> done; std::atomic<bool> abandoned{false}; };
  auto st = std::make_shared<Init>();
  std::future<int> fut = st->done.get_future();
  std::thread([st, world, u, rank, device] {
    (void)hipSetDevice(device);
    const int rc = g_rccl.CommInitRank(&st->comm, world, u, rank);
    if (st->abandoned.load() && rc == 0 && st->comm && g_rccl.CommAbort) g_rccl.CommAbort(st->comm);   // nobody is waiting any more
    st->done.set_value(rc);
  }).detach();
  if (fut.wait_for(std::chrono::duration<double>(comm_timeout_s())) != std::future_status::ready) {
    st->abandoned.store(true);   // should the helper ever get its communicator, it aborts it instead of leaking it
    delete c;
    return fail(RH_E_DEVICE, "ncclCommInitRank: rank " + std::to_string(rank) + " of " + std::to_string(world) + " waited " +
                             std::to_string((int)comm_timeout_s()) + " s (RH_COMM_TIMEOUT_S) for the other ranks -- one of them never reached rh_comm_create");
  }
  const int rc = fut.get();
  c->comm = st->comm;
  if (rc) { c->comm = nullptr; delete c; return nccl_fail("ncclCommInitRank", rc); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->d_scalar, 2 * sizeof(double)) != hipSuccess) {
    rh_comm_destroy(c);
    return fail(RH_E_DEVICE, "rh_comm_create: stream / buffer allocation failed");
  }
  *out = c;
  return RH_OK;
}

extern "C" void rh_comm_destroy(rh_comm *c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->poisoned) {
    if (c->comm && g_rccl.CommAbort) g_rccl.CommAbort(c->comm);
    delete c;   // (the device buffers and the stream stay: work that cannot be recalled may still use them)
    return;
  }
  if (c->comm) g_rccl.CommDestroy(c->comm);
  if (c->d_gather) hipFree(c->d_gather);
  if (c->d_scalar) hipFree(c->d_scalar);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

// draws of every rank's sampler (each [chains][iterations][nvars] fp64, device-resident) -> [world * chains][...] in rank
// order == global chain id order.  host_out (may be NULL) receives the gathered draws; *dev_out (may be NULL) the device
// pointer of the gathered buffer (owned by the communicator, valid until the next gather / destroy).
extern "C" int rh_comm_allgather_draws(rh_comm *c, rh_sampler *s, double *host_out, void **dev_out) {
  if (!c || !s) return fail(RH_E_INVALID, "rh_comm_allgather_draws: NULL");
  if (c->poisoned) return fail(RH_E_DEVICE, "rh_comm_allgather_draws: an earlier collective of this communicator timed out; it accepts no further work");
  int dev = 0; void *sstream = nullptr; int64_t count = 0; void *d_draws = nullptr;
  if (rh_sampler_geometry_(s, &dev, &sstream, &count) != RH_OK || rh_sampler_draws_device(s, &d_draws) != RH_OK)
    return fail(RH_E_INVALID, "rh_comm_allgather_draws: bad sampler handle");
  if (dev != c->device) return fail(RH_E_INVALID, "rh_comm_allgather_draws: sampler and communicator are on different devices");
  if (hipSetDevice(c->device) != hipSuccess) return fail(RH_E_DEVICE, "hipSetDevice failed");
  const size_t bytes = (size_t)count * sizeof(double) * (size_t)c->world;
  if (bytes > c->gather_bytes) {
    if (c->d_gather) hipFree(c->d_gather);
    c->d_gather = nullptr; c->gather_bytes = 0;
    if (hipMalloc(&c->d_gather, bytes ? bytes : 8) != hipSuccess) return fail(RH_E_DEVICE, "rh_comm_allgather_draws: hipMalloc failed");
    c->gather_bytes = bytes;
  }
  // the sampler's launches are complete when rh_sampler_run returns (it synchronises its stream): no cross-stream event needed
  const int rc = g_rccl.AllGather(d_draws, c->d_gather, (size_t)count, kNcclFloat64, c->comm, c->stream);
  if (rc) return nccl_fail("ncclAllGather", rc);
  { std::string werr; if (!wait_stream(c->stream, "rh_comm_allgather_draws (ncclAllGather)", werr)) { c->poisoned = true; return fail(RH_E_DEVICE, werr); } }
  if (host_out && hipMemcpy(host_out, c->d_gather, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(RH_E_DEVICE, "rh_comm_allgather_draws: copy to host failed");
  if (dev_out) *dev_out = c->d_gather;
  return RH_OK;
}

// max over ranks of one double (the timing reduction of a benchmark) -- also a device-side barrier
extern "C" int rh_comm_allreduce_max(rh_comm *c, double *value) {
  if (!c || !value) return fail(RH_E_INVALID, "rh_comm_allreduce_max: NULL");
  if (c->poisoned) return fail(RH_E_DEVICE, "rh_comm_allreduce_max: an earlier collective of this communicator timed out; it accepts no further work");
  if (hipSetDevice(c->device) != hipSuccess) return fail(RH_E_DEVICE, "hipSetDevice failed");
  if (hipMemcpy(c->d_scalar, value, sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(RH_E_DEVICE, "copy failed");
  const int rc = g_rccl.AllReduce(c->d_scalar, (char *)c->d_scalar + sizeof(double), 1, kNcclFloat64, kNcclMax, c->comm, c->stream);
  if (rc) return nccl_fail("ncclAllReduce", rc);
  { std::string werr; if (!wait_stream(c->stream, "rh_comm_allreduce_max (ncclAllReduce)", werr)) { c->poisoned = true; return fail(RH_E_DEVICE, werr); } }
  if (hipMemcpy(value, (char *)c->d_scalar + sizeof(double), sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(RH_E_DEVICE, "copy failed");
  return RH_OK;
}

// hipDeviceSynchronize on `device`: the "synchronise on both sides of the timed region" of a benchmark whose host code does
// not otherwise touch the HIP runtime
extern "C" int rh_device_synchronize(int32_t device) {
  if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(RH_E_DEVICE, "rh_device_synchronize failed");
  return RH_OK;
}
