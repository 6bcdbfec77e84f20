// The gradient exchange of data-parallel training as C-ABI entry points (SURVEY.md 8b export list: evmi_allreduce_bucket): a
// thin wrapper over RCCL's all-reduce (xGMI between the GPUs of a node) for hosts that do not go through torch.distributed --
// the Python trainers do (train/hifigan.py: BucketReducer), a C / C++ host calls these.  One process per GPU:
//
//     rank 0:  evmi_comm_unique_id(id)  --(any side channel)-->  every rank:  evmi_comm_init_rank(&comm, world, id, rank)
//     per bucket, as backward finishes it:  evmi_allreduce_bucket(comm, grad + lo, hi - lo, 1.f / world, stream)
//
// sums the contiguous fp32 slice in place over the ranks (ring / tree chosen by RCCL; buckets of tens of MB keep the per-link
// xGMI rings busy) and scales by `scale` (the 1 / world of a gradient mean) on the same stream, so the optimiser kernel queued
// behind it sees averaged gradients without a host round trip.
//
// RCCL is resolved at run time (dlopen / dlsym) instead of being linked: a process that already holds RCCL (PyTorch bundles
// its own copy) must not get a second one, and hosts that never call these functions need no RCCL at all.
#include <dlfcn.h>

#include <cstring>

#include "common.h"

namespace evmi {

typedef int (*fn_get_unique_id)(void*);
struct UniqueIdBytes { char internal[128]; };  // ncclUniqueId is passed BY VALUE to ncclCommInitRank
typedef int (*fn_comm_init_rank_t)(void**, int, UniqueIdBytes, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct RcclApi {
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank_t comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_error_string error_string = nullptr;
  bool ok = false;
};

static RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    void* h = nullptr;
    // a copy already mapped into the process first (torch's bundled one), then the system library
    if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) h = RTLD_DEFAULT;
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return a;
    a.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    a.comm_init_rank = (fn_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
    a.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    a.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    a.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_reduce;
    return a;
  }();
  return api;
}

static int rccl_fail(const char* what, int rc) {
  RcclApi& a = rccl();
  return fail(EVMI_ERR_HIP, std::string(what) + ": " + (a.error_string ? a.error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

__global__ void scale_inplace_kernel(float* x, long long n, float s) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

int evmi_comm_unique_id(void* id_out_128_bytes) {
  if (!id_out_128_bytes) return fail(EVMI_ERR_INVALID_ARG, "comm_unique_id: null pointer");
  if (!rccl().ok) return fail(EVMI_ERR_UNSUPPORTED, "comm_unique_id: RCCL (librccl.so) could not be loaded");
  const int rc = rccl().get_unique_id(id_out_128_bytes);
  return rc ? rccl_fail("ncclGetUniqueId", rc) : EVMI_OK;
}

int evmi_comm_init_rank(void** comm_out, int world, const void* id_128_bytes, int rank) {
  if (!comm_out || !id_128_bytes || world <= 0 || rank < 0 || rank >= world) return fail(EVMI_ERR_INVALID_ARG, "comm_init_rank: arguments");
  if (!rccl().ok) return fail(EVMI_ERR_UNSUPPORTED, "comm_init_rank: RCCL (librccl.so) could not be loaded");
  UniqueIdBytes id;
  memcpy(id.internal, id_128_bytes, sizeof(id.internal));
  const int rc = rccl().comm_init_rank(comm_out, world, id, rank);
  return rc ? rccl_fail("ncclCommInitRank", rc) : EVMI_OK;
}

int evmi_comm_destroy(void* comm) {
  if (!comm) return EVMI_OK;
  if (!rccl().ok) return fail(EVMI_ERR_UNSUPPORTED, "comm_destroy: RCCL (librccl.so) could not be loaded");
  const int rc = rccl().comm_destroy(comm);
  return rc ? rccl_fail("ncclCommDestroy", rc) : EVMI_OK;
}

int evmi_allreduce_bucket(void* comm, float* grad_dev, long long n, float scale, void* stream) {
  if (!comm || !grad_dev || n < 0) return fail(EVMI_ERR_INVALID_ARG, "allreduce_bucket: arguments");
  if (n == 0) return EVMI_OK;
  if (!rccl().ok) return fail(EVMI_ERR_UNSUPPORTED, "allreduce_bucket: RCCL (librccl.so) could not be loaded");
  const int rc = rccl().all_reduce(grad_dev, grad_dev, (size_t)n, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, (hipStream_t)stream);
  if (rc) return rccl_fail("ncclAllReduce", rc);
  if (scale != 1.f) {
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_dev, n, scale);
    EVMI_LAUNCH_CHECK("allreduce_bucket scale");
  }
  return EVMI_OK;
}

}  // extern "C"
