// Data-parallel exchange entry points of the C ABI (coati_comm_init / allgather_rows / reducescatter_rows /
// allreduce_bucket / comm_destroy) for hosts that do not bring torch.distributed: thin, stream-ordered calls into RCCL
// (librccl, the xGMI collectives library of ROCm).  The Python host of this repository keeps using torch.distributed -- the
// reference binds it too (coati/models/autograd_funs/autograd_funs.py:10-21, coati/training/train_coati.py:71-76) -- and its
// "nccl" backend IS this library; these entries give a C / C++ host the same three collectives of the step:
//   all-gather of the [B, E] embedding rows (AllGatherFunction.forward), reduce-scatter(sum) of their gradients
//   (AllGatherFunction.backward), all-reduce(average) of a gradient bucket (DistributedDataParallel).
// RCCL is resolved with dlopen at the first coati_comm_* call: the library has no link-time dependency on it, a single-GPU host
// never loads it, and inside a torch process the already loaded librccl is the one that answers.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include "common.h"

// The handful of RCCL (= NCCL API) types and enumerators these entries use, declared here: the library builds on a ROCm install
// without the rccl development headers (a single-GPU host), and the values are the stable NCCL 2.x ABI (nccl.h: ncclResult_t 0 =
// success; ncclDataType_t float32 = 7, bfloat16 = 9; ncclRedOp_t sum = 0, avg = 4; a 128-byte opaque unique id).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclAvg = 4 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
}

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

template <typename F>
bool sym(void* h, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(h, name));
  return out != nullptr;
}
void load_rccl() {
  const char* names[] = {getenv("COATI_RCCL_LIB"), "librccl.so.1", "librccl.so"};
  for (const char* n : names) {
    if (n == nullptr || *n == 0) continue;
    void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy the process already holds (torch's), if any
    if (h == nullptr) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) continue;
    Rccl r;
    r.handle = h;
    if (sym(h, "ncclGetUniqueId", r.GetUniqueId) && sym(h, "ncclCommInitRank", r.CommInitRank) && sym(h, "ncclCommDestroy", r.CommDestroy) &&
        sym(h, "ncclAllGather", r.AllGather) && sym(h, "ncclReduceScatter", r.ReduceScatter) && sym(h, "ncclAllReduce", r.AllReduce) &&
        sym(h, "ncclGetErrorString", r.GetErrorString)) {
      r.ok = true;
      g_rccl = r;
      return;
    }
    dlclose(h);
  }
}
int need_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.ok) {
    const char* why = dlerror();
    coati_set_error("coati_comm: librccl not found (tried $COATI_RCCL_LIB, librccl.so.1, librccl.so): %s", why ? why : "no loader message");
    return COATI_EHIP;
  }
  return COATI_OK;
}
int rc_of(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return COATI_OK;
  coati_set_error("%s: RCCL error %d: %s", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return COATI_EHIP;
}
// element types of the ABI (include/coati_hip.h): 0 = f32, 1 = bf16
bool dtype_of(int dtype, ncclDataType_t& t, size_t& bytes) {
  if (dtype == 0) { t = ncclFloat32; bytes = 4; return true; }
  if (dtype == 1) { t = ncclBfloat16; bytes = 2; return true; }
  return false;
}
}  // namespace

struct coati_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" {

int coati_comm_unique_id(void* id_out, int id_bytes) {
  COATI_CHECK_ARG(id_out != nullptr && id_bytes >= (int)NCCL_UNIQUE_ID_BYTES, "coati_comm_unique_id: need a buffer of %d bytes", (int)NCCL_UNIQUE_ID_BYTES);
  COATI_TRY(need_rccl());
  ncclUniqueId id;
  COATI_TRY(rc_of(g_rccl.GetUniqueId(&id), "coati_comm_unique_id"));
  std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return COATI_OK;
}

int coati_comm_init(const void* unique_id, int rank, int world, coati_comm** out) {
  COATI_CHECK_ARG(unique_id != nullptr && out != nullptr, "coati_comm_init: null argument");
  COATI_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "coati_comm_init: rank %d of %d", rank, world);
  COATI_TRY(need_rccl());
  ncclUniqueId id;
  std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  coati_comm* c = new coati_comm;
  c->rank = rank;
  c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    coati_set_error("coati_comm_init: no current HIP device");
    return COATI_EHIP;
  }
  const int rc = rc_of(g_rccl.CommInitRank(&c->comm, world, id, rank), "coati_comm_init");   // one communicator per process = per GPU
  if (rc != COATI_OK) {
    delete c;
    return rc;
  }
  *out = c;
  return COATI_OK;
}

int coati_comm_rank(const coati_comm* c) { return c ? c->rank : -1; }
int coati_comm_world(const coati_comm* c) { return c ? c->world : -1; }

int coati_comm_destroy(coati_comm* c) {
  if (c == nullptr) return COATI_OK;
  int rc = COATI_OK;
  if (c->comm != nullptr) rc = rc_of(g_rccl.CommDestroy(c->comm), "coati_comm_destroy");
  delete c;
  return rc;
}

// the communicator belongs to the device that was current at coati_comm_init: a call from another device would enqueue on a stream
// of the wrong GPU
static int check_device(const coati_comm* c, const char* what) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != c->device) {
    coati_set_error("%s: current device %d, the communicator was created on device %d", what, dev, c->device);
    return COATI_EARG;
  }
  return COATI_OK;
}

// recv[world * rows, cols] <- concatenation over ranks of send[rows, cols] (rank-major)
int coati_allgather_rows(coati_comm* c, const void* send, void* recv, int64_t rows, int64_t cols, int dtype, void* stream) {
  COATI_CHECK_ARG(c && send && recv && rows >= 0 && cols >= 0, "coati_allgather_rows: bad argument");
  ncclDataType_t t;
  size_t b;
  COATI_CHECK_ARG(dtype_of(dtype, t, b), "coati_allgather_rows: dtype %d (0 = f32, 1 = bf16)", dtype);
  COATI_TRY(check_device(c, "coati_allgather_rows"));
  return rc_of(g_rccl.AllGather(send, recv, (size_t)(rows * cols), t, c->comm, (hipStream_t)stream), "coati_allgather_rows");
}

// recv[rows, cols] <- this rank's row block of the sum over ranks of send[world * rows, cols]
int coati_reducescatter_rows(coati_comm* c, const void* send, void* recv, int64_t rows, int64_t cols, int dtype, void* stream) {
  COATI_CHECK_ARG(c && send && recv && rows >= 0 && cols >= 0, "coati_reducescatter_rows: bad argument");
  ncclDataType_t t;
  size_t b;
  COATI_CHECK_ARG(dtype_of(dtype, t, b), "coati_reducescatter_rows: dtype %d (0 = f32, 1 = bf16)", dtype);
  COATI_TRY(check_device(c, "coati_reducescatter_rows"));
  return rc_of(g_rccl.ReduceScatter(send, recv, (size_t)(rows * cols), t, ncclSum, c->comm, (hipStream_t)stream), "coati_reducescatter_rows");
}

// buf[n] <- sum (average != 0: mean) over ranks, in place
int coati_allreduce_bucket(coati_comm* c, void* buf, int64_t n, int dtype, int average, void* stream) {
  COATI_CHECK_ARG(c && buf && n >= 0, "coati_allreduce_bucket: bad argument");
  ncclDataType_t t;
  size_t b;
  COATI_CHECK_ARG(dtype_of(dtype, t, b), "coati_allreduce_bucket: dtype %d (0 = f32, 1 = bf16)", dtype);
  COATI_TRY(check_device(c, "coati_allreduce_bucket"));
  return rc_of(g_rccl.AllReduce(buf, buf, (size_t)n, t, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream), "coati_allreduce_bucket");
}

}  // extern "C"
