// The one collective of the path, behind the C ABI: an all-gather of the per-rank logits over RCCL / xGMI
// (the batch axis of `jax.vmap(net)(images)` -- README.md:37-40 -- shards across the GPUs of a node,
// rank r owns images[r*B/W : (r+1)*B/W], one ncclAllGather of f32[B/W, classes] rebuilds the (B, classes) result).
//
// librccl is ~570 MB, so it is NOT a link-time dependency: it is dlopen'ed by the first mv_comm_* call.  Search order:
// $EQV_RCCL_LIB, a librccl already mapped into the process (e.g. PyTorch's bundled copy), librccl.so.1 on the loader
// path, /opt/rocm/lib/librccl.so.1.  One communicator per process (one process per GPU).
#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace mv {
namespace {

typedef struct { char internal[128]; } nccl_uid_t;            // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
typedef int (*fn_get_uid)(nccl_uid_t*);
typedef int (*fn_init_rank)(nccl_comm_t*, int, nccl_uid_t, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int /*ncclDataType_t*/, nccl_comm_t, hipStream_t);
typedef int (*fn_allreduce)(const void*, void*, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, nccl_comm_t, hipStream_t);
typedef int (*fn_destroy)(nccl_comm_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* h = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allgather allgather = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
};
std::mutex g_cmu;
Rccl g_rccl;
nccl_comm_t g_comm = nullptr;
int g_rank = 0, g_nranks = 0;

int load_rccl() {
    if (g_rccl.h) return MV_OK;
    const char* env = getenv("EQV_RCCL_LIB");
    void* h = nullptr;
    if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error("mv_comm: cannot load librccl (set EQV_RCCL_LIB): %s", dlerror());
        return MV_E_UNSUPPORTED;
    }
    Rccl r;
    r.h = h;
    r.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    r.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    r.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
    r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!r.get_uid || !r.init_rank || !r.allgather || !r.destroy) {
        set_error("mv_comm: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
        return MV_E_UNSUPPORTED;
    }
    g_rccl = r;
    return MV_OK;
}

int nccl_fail(const char* what, int rc) {
    set_error("mv_comm: %s failed: ncclResult %d (%s)", what, rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    return rc > 0 ? rc : MV_E_INVALID;
}

}  // namespace
}  // namespace mv

extern "C" {

int mv_comm_unique_id(void* out, size_t out_bytes) {
    MV_CHECK_ARG(out && out_bytes >= MV_COMM_ID_BYTES, "mv_comm_unique_id: need a %d-byte buffer", MV_COMM_ID_BYTES);
    std::lock_guard<std::mutex> lk(mv::g_cmu);
    const int rc = mv::load_rccl();
    if (rc != MV_OK) return rc;
    mv::nccl_uid_t id;
    const int n = mv::g_rccl.get_uid(&id);
    if (n != 0) return mv::nccl_fail("ncclGetUniqueId", n);
    memcpy(out, &id, sizeof(id));
    return MV_OK;
}

int mv_comm_init(int rank, int nranks, const void* unique_id) {
    MV_CHECK_ARG(unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "mv_comm_init: bad rank %d / %d", rank, nranks);
    std::lock_guard<std::mutex> lk(mv::g_cmu);
    if (mv::g_comm) {
        mv::set_error("mv_comm_init: a communicator already exists (one per process); call mv_comm_destroy first");
        return MV_E_INVALID;
    }
    const int rc = mv::load_rccl();
    if (rc != MV_OK) return rc;
    mv::nccl_uid_t id;
    memcpy(&id, unique_id, sizeof(id));
    mv::nccl_comm_t c = nullptr;
    const int n = mv::g_rccl.init_rank(&c, nranks, id, rank);      // binds the CURRENT HIP device, like every mv_* call
    if (n != 0) return mv::nccl_fail("ncclCommInitRank", n);
    mv::g_comm = c;
    mv::g_rank = rank;
    mv::g_nranks = nranks;
    return MV_OK;
}

int mv_comm_size(void) { return mv::g_comm ? mv::g_nranks : 0; }
int mv_comm_rank(void) { return mv::g_comm ? mv::g_rank : -1; }

int mv_allgather(const void* send, void* recv, size_t bytes_per_rank, mv_stream_t stream) {
    MV_CHECK_ARG(send && recv, "mv_allgather: NULL buffer");
    if (!mv::g_comm) {
        mv::set_error("mv_allgather: no communicator (mv_comm_init first)");
        return MV_E_INVALID;
    }
    if (bytes_per_rank == 0) return MV_OK;
    const int n = mv::g_rccl.allgather(send, recv, bytes_per_rank, /*ncclInt8*/ 0, mv::g_comm, (hipStream_t)stream);
    if (n != 0) return mv::nccl_fail("ncclAllGather", n);
    return MV_OK;
}

// In-place sum over ranks of `count` fp32 values: BatchNorm's training-mode `pmean` of the per-channel batch moments
// (2 x C floats per layer: latency-bound, a direct exchange over xGMI).
int mv_allreduce_sum_f32(void* buf, size_t count, mv_stream_t stream) {
    MV_CHECK_ARG(buf, "mv_allreduce_sum_f32: NULL buffer");
    if (!mv::g_comm) {
        mv::set_error("mv_allreduce_sum_f32: no communicator (mv_comm_init first)");
        return MV_E_INVALID;
    }
    if (!mv::g_rccl.allreduce) {
        mv::set_error("mv_allreduce_sum_f32: librccl lacks ncclAllReduce");
        return MV_E_UNSUPPORTED;
    }
    if (count == 0) return MV_OK;
    const int n = mv::g_rccl.allreduce(buf, buf, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, mv::g_comm, (hipStream_t)stream);
    if (n != 0) return mv::nccl_fail("ncclAllReduce", n);
    return MV_OK;
}

int mv_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(mv::g_cmu);
    if (!mv::g_comm) return MV_OK;
    const int n = mv::g_rccl.destroy(mv::g_comm);
    mv::g_comm = nullptr;
    mv::g_nranks = 0;
    if (n != 0) return mv::nccl_fail("ncclCommDestroy", n);
    return MV_OK;
}

}  // extern "C"
