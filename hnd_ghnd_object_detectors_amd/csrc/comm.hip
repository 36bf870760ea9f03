// Gradient exchange of the data-parallel distillation step: one RCCL communicator per process (one process per
// GPU), one in-place averaging all-reduce of the flat gradient arena per step over xGMI.
//
// Replaces what torch DistributedDataParallel does for the reference (src/mimic_runner.py:141-143 wraps the student,
// src/utils/main_util.py:43-62 init_distributed_mode creates the process group): DDP buckets the 25 trainable
// tensors and averages them across ranks; here they already sit in ONE flat arena (586 566 floats = 2.35 MB), so the
// exchange is a single ncclAllReduce(ncclAvg) -- ring all-reduce over the point-to-point xGMI links, far below the
// size where more than one bucket would pay.
//
// RCCL is resolved at run time (dlopen) instead of at link time: the process normally already holds the RCCL that
// PyTorch-ROCm loaded (torch/lib/librccl.so), and two RCCL copies in one process must be avoided; it also keeps
// libhnd_hip.so loadable on hosts without RCCL (single-GPU use never touches this file's entry points).
#include "common.h"

#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>      // types and prototypes only; no symbol of it is linked

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  char why[256] = "";
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // 1) whatever RCCL the process already holds (PyTorch's), 2) HND_RCCL_PATH, 3) the system library
    const char* env = getenv("HND_RCCL_PATH");
    const char* names[] = {"librccl.so", "librccl.so.1", env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (int i = 0; i < 6 && !r.handle; ++i) {
      if (!names[i]) continue;
      r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | (i < 2 ? RTLD_NOLOAD : 0));
    }
    if (!r.handle) {
      snprintf(r.why, sizeof(r.why), "RCCL not found (dlopen librccl.so: %s)", dlerror());
      return;
    }
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.handle, "ncclAllReduce");
    r.error_string = (decltype(r.error_string))dlsym(r.handle, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.error_string) {
      snprintf(r.why, sizeof(r.why), "RCCL library lacks a required symbol");
      r.handle = nullptr;
    }
  });
  return r;
}

struct Comm {
  ncclComm_t nccl;
  int rank, world;
};

}  // namespace

extern "C" {

int hnd_comm_unique_id(void* id_out, size_t bytes) {
  HND_REQUIRE(id_out && bytes >= sizeof(ncclUniqueId), "hnd_comm_unique_id: need a %zu-byte buffer", sizeof(ncclUniqueId));
  Rccl& r = rccl();
  HND_REQUIRE(r.handle, "hnd_comm_unique_id: %s", r.why);
  ncclUniqueId id;
  ncclResult_t e = r.get_unique_id(&id);
  if (e != ncclSuccess) {
    hnd::set_error("hnd_comm_unique_id: ncclGetUniqueId: %s", r.error_string(e));
    return HND_ERR_LAUNCH;
  }
  memcpy(id_out, &id, sizeof(id));
  return HND_OK;
}

int hnd_comm_init(int rank, int world, const void* unique_id, size_t bytes, void** comm_out) {
  HND_REQUIRE(comm_out && unique_id && bytes >= sizeof(ncclUniqueId) && world >= 1 && rank >= 0 && rank < world,
              "hnd_comm_init: bad arguments (rank %d of %d)", rank, world);
  Rccl& r = rccl();
  HND_REQUIRE(r.handle, "hnd_comm_init: %s", r.why);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t e = r.comm_init_rank(&c, world, id, rank);      // binds to the calling thread's current HIP device
  if (e != ncclSuccess) {
    hnd::set_error("hnd_comm_init: ncclCommInitRank(rank %d / %d): %s", rank, world, r.error_string(e));
    return HND_ERR_LAUNCH;
  }
  *comm_out = new Comm{c, rank, world};
  return HND_OK;
}

int hnd_allreduce_avg_flat(void* comm, float* flat, int64_t n, void* stream) {
  HND_REQUIRE(comm && flat && n > 0, "hnd_allreduce_avg_flat: bad arguments");
  Comm* c = (Comm*)comm;
  Rccl& r = rccl();
  ncclResult_t e = r.all_reduce(flat, flat, (size_t)n, ncclFloat32, ncclAvg, c->nccl, hnd::as_stream(stream));
  if (e != ncclSuccess) {
    hnd::set_error("hnd_allreduce_avg_flat: ncclAllReduce(%lld floats): %s", (long long)n, r.error_string(e));
    return HND_ERR_LAUNCH;
  }
  return HND_OK;
}

int hnd_comm_destroy(void* comm) {
  if (!comm) return HND_OK;
  Comm* c = (Comm*)comm;
  Rccl& r = rccl();
  ncclResult_t e = r.handle ? r.comm_destroy(c->nccl) : ncclSuccess;
  delete c;
  if (e != ncclSuccess) {
    hnd::set_error("hnd_comm_destroy: %s", r.error_string(e));
    return HND_ERR_LAUNCH;
  }
  return HND_OK;
}

}  // extern "C"
