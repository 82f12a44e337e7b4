// RCCL entry points of the C ABI: the gradient exchange of data-parallel training
// (reference: DistributedDataParallel in promptttspp/trainers/tts.py:52-55,117 -- init_process_group("nccl")
// + the bucketed all-reduce DDP issues during backward).
//
// librccl is bound at run time with dlopen, not at link time: inside a PyTorch process this resolves to the
// librccl.so torch has already loaded (one RCCL instance per process), a plain C/C++ host gets ROCm's, and
// libptpp_hip.so still loads on a box without RCCL (the calls then fail with PTPP_ENOTSUP).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/ptpp.h"

void ptpp_set_error(const char* fmt, ...);

namespace {

// the slice of rccl.h this file needs (opaque handle, 128-byte id passed BY VALUE, enum values)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*ErrStrFn)(int);
constexpr int kFloat32 = 7, kBfloat16 = 9, kSum = 0, kAvg = 4;

struct Rccl {
  void* h = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init_rank = nullptr;
  CommDestroyFn destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  BroadcastFn broadcast = nullptr;
  ErrStrFn err = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
      if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // the instance the process already has (torch's)
    for (const char* n : names)
      if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.h) {
      r.get_id = (GetUniqueIdFn)dlsym(r.h, "ncclGetUniqueId");
      r.init_rank = (CommInitRankFn)dlsym(r.h, "ncclCommInitRank");
      r.destroy = (CommDestroyFn)dlsym(r.h, "ncclCommDestroy");
      r.all_reduce = (AllReduceFn)dlsym(r.h, "ncclAllReduce");
      r.broadcast = (BroadcastFn)dlsym(r.h, "ncclBroadcast");
      r.err = (ErrStrFn)dlsym(r.h, "ncclGetErrorString");
    }
  }
  if (!r.h || !r.get_id || !r.init_rank || !r.destroy || !r.all_reduce || !r.broadcast) {
    ptpp_set_error("RCCL is not available (dlopen librccl.so: %s)", r.h ? "missing symbols" : dlerror());
    return nullptr;
  }
  return &r;
}

int fail(Rccl* r, const char* what, int rc) {
  ptpp_set_error("%s failed: %s (ncclResult %d)", what, r->err ? r->err(rc) : "?", rc);
  return PTPP_ELAUNCH;
}

}  // namespace

extern "C" int ptpp_comm_unique_id(void* id_out) {
  if (!id_out) { ptpp_set_error("comm_unique_id: null pointer"); return PTPP_EINVAL; }
  Rccl* r = rccl();
  if (!r) return PTPP_ENOTSUP;
  UniqueId id;
  const int rc = r->get_id(&id);
  if (rc != 0) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id_out, &id, sizeof(id));
  return PTPP_OK;
}

extern "C" int ptpp_comm_init(int rank, int world, const void* unique_id, void** comm_out) {
  if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) {
    ptpp_set_error("comm_init: bad arguments (rank %d of %d)", rank, world);
    return PTPP_EINVAL;
  }
  Rccl* r = rccl();
  if (!r) return PTPP_ENOTSUP;
  UniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  Comm c = nullptr;
  const int rc = r->init_rank(&c, world, id, rank);
  if (rc != 0) return fail(r, "ncclCommInitRank", rc);
  *comm_out = c;
  return PTPP_OK;
}

extern "C" int ptpp_comm_destroy(void* comm) {
  if (!comm) return PTPP_OK;
  Rccl* r = rccl();
  if (!r) return PTPP_ENOTSUP;
  const int rc = r->destroy(comm);
  return rc == 0 ? PTPP_OK : fail(r, "ncclCommDestroy", rc);
}

extern "C" int ptpp_allreduce_mean(void* buf, int64_t n, int dtype, void* comm, void* stream) {
  if (!buf || !comm || n < 0 || (dtype != PTPP_F32 && dtype != PTPP_BF16)) {
    ptpp_set_error("allreduce_mean: bad arguments (n %lld, dtype %d)", (long long)n, dtype);
    return PTPP_EINVAL;
  }
  if (n == 0) return PTPP_OK;
  Rccl* r = rccl();
  if (!r) return PTPP_ENOTSUP;
  const int rc = r->all_reduce(buf, buf, (size_t)n, dtype == PTPP_F32 ? kFloat32 : kBfloat16, kAvg, comm,
                               reinterpret_cast<hipStream_t>(stream));
  return rc == 0 ? PTPP_OK : fail(r, "ncclAllReduce", rc);
}

extern "C" int ptpp_broadcast(void* buf, int64_t n, int dtype, int root, void* comm, void* stream) {
  if (!buf || !comm || n < 0 || (dtype != PTPP_F32 && dtype != PTPP_BF16)) {
    ptpp_set_error("broadcast: bad arguments (n %lld, dtype %d)", (long long)n, dtype);
    return PTPP_EINVAL;
  }
  if (n == 0) return PTPP_OK;
  Rccl* r = rccl();
  if (!r) return PTPP_ENOTSUP;
  const int rc = r->broadcast(buf, buf, (size_t)n, dtype == PTPP_F32 ? kFloat32 : kBfloat16, root, comm,
                              reinterpret_cast<hipStream_t>(stream));
  return rc == 0 ? PTPP_OK : fail(r, "ncclBroadcast", rc);
}
