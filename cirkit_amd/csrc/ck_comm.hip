// The path's one exchange step through the C ABI: a SUM all-reduce over RCCL / xGMI, enqueued on the caller's stream.
// (the [sum log p, count] pair of a forward, the flat gradient buffer of a training step.)
// RCCL is bound at run time (dlopen: the copy already in the process -- PyTorch-ROCm ships one built against the HIP runtime it
// loaded -- else the path the caller names, else the system's): the library keeps loading on a box without RCCL, and
// ck_comm_* report CK_ERR_UNSUPPORTED there.  The reference has no distributed code at all; this replaces what a user of it
// would write with torch.distributed around `TorchCircuit.forward`.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "ck_internal.h"

namespace {

// rccl.h's ABI, restated (the enum values and the 128-byte id are fixed by NCCL's public interface)
struct UniqueId {
  char internal[128];
};
using Comm = void*;
enum { kFloat32 = 7, kFloat64 = 8, kSum = 0 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  char origin[256] = "";
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

bool bind(void* h, const char* origin) {
  Rccl r;
  r.handle = h;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) return false;
  strncpy(r.origin, origin, sizeof(r.origin) - 1);
  g_rccl = r;
  return true;
}

int load_rccl(const char* path) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle != nullptr) return CK_OK;
  // a copy that is already mapped first: two RCCL builds in one process would each bring their own kernels and bootstrap state
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) {
      if (bind(h, name)) return CK_OK;
    }
  }
  if (path != nullptr && path[0] != 0) {
    if (void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL)) {
      if (bind(h, path)) return CK_OK;
    }
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
      if (bind(h, name)) return CK_OK;
    }
  }
  return ck::fail(CK_ERR_UNSUPPORTED, "ck_comm: librccl not found (%s)", dlerror() ? dlerror() : "no loader message");
}

struct CommState {
  Comm comm = nullptr;
  int rank = 0, world = 1, device = 0;
  // the communicator's own stream: collectives that nothing on the launch stream waits for (the [sum, count] pair of an
  // evaluation step) run beside the next step's kernels instead of between them
  hipStream_t side = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
};

}  // namespace

struct ck_comm {
  CommState s;
};

extern "C" {

int ck_comm_load(const char* librccl_path) { return load_rccl(librccl_path); }

int ck_comm_unique_id(void* id128) {
  CK_REQUIRE(id128 != nullptr, "ck_comm_unique_id: null id buffer");
  if (int st = load_rccl(nullptr)) return st;
  UniqueId id;
  const int r = g_rccl.GetUniqueId(&id);
  if (r != 0) return ck::fail(CK_ERR_HIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
  memcpy(id128, id.internal, sizeof(id.internal));
  return CK_OK;
}

int ck_comm_init(const void* id128, int rank, int world, int device, ck_comm** out) {
  CK_REQUIRE(id128 != nullptr && out != nullptr, "ck_comm_init: null argument");
  CK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ck_comm_init: rank %d outside a world of %d", rank, world);
  if (int st = load_rccl(nullptr)) return st;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return ck::fail(CK_ERR_HIP, "ck_comm_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  Comm c = nullptr;
  const int r = g_rccl.CommInitRank(&c, world, id, rank);  // blocks until every rank of the world has called it
  if (r != 0) return ck::fail(CK_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
  ck_comm* cm = new ck_comm();
  if (hipStreamCreateWithFlags(&cm->s.side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&cm->s.ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&cm->s.ev_out, hipEventDisableTiming) != hipSuccess) {
    g_rccl.CommDestroy(c);
    delete cm;
    return ck::fail(CK_ERR_HIP, "ck_comm_init: stream / event creation failed");
  }
  cm->s.comm = c;
  cm->s.rank = rank;
  cm->s.world = world;
  cm->s.device = device;
  *out = cm;
  return CK_OK;
}

static int all_reduce(ck_comm* comm, void* buf, int64_t n, int dtype, void* stream, const char* what) {
  CK_REQUIRE(comm != nullptr && comm->s.comm != nullptr, "%s: no communicator", what);
  CK_REQUIRE(buf != nullptr && n > 0, "%s: empty buffer", what);
  Comm c = comm->s.comm;
  auto fn = g_rccl.AllReduce;
  // (through ck::dispatch: inside ck_program_begin / _end the collective becomes a step of the recorded launch list)
  return ck::dispatch(
      [=](hipStream_t s) {
        const int r = fn(buf, buf, static_cast<size_t>(n), dtype, kSum, c, s);  // in place
        return r == 0 ? hipSuccess : hipErrorUnknown;
      },
      stream);
}

int ck_comm_all_reduce_f64(ck_comm* comm, double* buf, int64_t n, void* stream) {
  return all_reduce(comm, buf, n, kFloat64, stream, "ck_comm_all_reduce_f64");
}

int ck_comm_all_reduce_f32(ck_comm* comm, float* buf, int64_t n, void* stream) {
  return all_reduce(comm, buf, n, kFloat32, stream, "ck_comm_all_reduce_f32");
}

// The same collective on the communicator's OWN stream, ordered behind everything enqueued on `after_stream` so far: the launch
// stream goes on with the next step at once.  ck_comm_wait makes a stream wait for the LAST such collective (and, through the
// side stream's order, for all before it).
static int all_reduce_async(ck_comm* comm, void* buf, int64_t n, int dtype, void* after_stream, const char* what) {
  CK_REQUIRE(comm != nullptr && comm->s.comm != nullptr, "%s: no communicator", what);
  CK_REQUIRE(buf != nullptr && n > 0, "%s: empty buffer", what);
  CommState* st = &comm->s;
  auto fn = g_rccl.AllReduce;
  return ck::dispatch(
      [=](hipStream_t s) {
        hipError_t e = hipEventRecord(st->ev_in, s);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(st->side, st->ev_in, 0);
        if (e != hipSuccess) return e;
        if (fn(buf, buf, static_cast<size_t>(n), dtype, kSum, st->comm, st->side) != 0) return hipErrorUnknown;
        return hipEventRecord(st->ev_out, st->side);
      },
      after_stream);
}

int ck_comm_all_reduce_async_f64(ck_comm* comm, double* buf, int64_t n, void* after_stream) {
  return all_reduce_async(comm, buf, n, kFloat64, after_stream, "ck_comm_all_reduce_async_f64");
}

int ck_comm_wait(ck_comm* comm, void* stream) {
  CK_REQUIRE(comm != nullptr && comm->s.comm != nullptr, "ck_comm_wait: no communicator");
  CommState* st = &comm->s;
  return ck::dispatch([=](hipStream_t s) { return hipStreamWaitEvent(s, st->ev_out, 0); }, stream);
}

int ck_comm_info(const ck_comm* comm, int32_t out[3], char* origin, int origin_len) {
  CK_REQUIRE(comm != nullptr && out != nullptr, "ck_comm_info: null argument");
  out[0] = comm->s.rank;
  out[1] = comm->s.world;
  out[2] = comm->s.device;
  if (origin != nullptr && origin_len > 0) {
    strncpy(origin, g_rccl.origin, static_cast<size_t>(origin_len) - 1);
    origin[origin_len - 1] = 0;
  }
  return CK_OK;
}

int ck_comm_destroy(ck_comm* comm) {
  if (comm == nullptr) return CK_OK;
  int r = 0;
  if (comm->s.side != nullptr) {
    (void)hipStreamSynchronize(comm->s.side);
    (void)hipStreamDestroy(comm->s.side);
  }
  if (comm->s.ev_in != nullptr) (void)hipEventDestroy(comm->s.ev_in);
  if (comm->s.ev_out != nullptr) (void)hipEventDestroy(comm->s.ev_out);
  if (comm->s.comm != nullptr) r = g_rccl.CommDestroy(comm->s.comm);
  delete comm;
  if (r != 0) return ck::fail(CK_ERR_HIP, "ncclCommDestroy: %s", g_rccl.GetErrorString(r));
  return CK_OK;
}

}  // extern "C"
