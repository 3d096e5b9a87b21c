// Multi-GPU exchange of the path: the SUM all-reduce of the histograms over the ranks of one node,
// RCCL over xGMI, behind the C-ABI so that a binder of include/vbq.h needs nothing but this library.
//
// RCCL is bound at RUN time (dlopen / dlsym): libvbq_hip.so carries no link-time dependency on it, loads on hosts
// without RCCL, and shares the copy a host framework (PyTorch ships its own librccl.so) has already mapped instead
// of bringing a second one into the process.   gfx950 / ROCm only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "vbq_common.h"

namespace vbq {
namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static const Rccl r = [] {
        Rccl x;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {           // a copy that is already mapped (the host framework's) wins
            x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (x.handle) break;
        }
        for (const char *n : names) {
            if (x.handle) break;
            x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!x.handle) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.handle, "ncclAllReduce"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy && x.GetErrorString;
        return x;
    }();
    return r;
}

#define VBQ_RCCL_OR_FAIL(who)                                                                                  \
    const Rccl &R = rccl();                                                                                    \
    VBQ_REQUIRE(R.ok, VBQ_ERR_UNSUPPORTED, "%s: librccl.so could not be loaded (%s)", who,                    \
                R.handle ? "missing symbols" : "dlopen failed")

#define VBQ_RCCL_CHECK(who, call)                                                       \
    do {                                                                                \
        ncclResult_t r__ = (call);                                                      \
        if (r__ != ncclSuccess) {                                                       \
            set_error("%s: %s", who, R.GetErrorString(r__));                            \
            return VBQ_ERR_LAUNCH;                                                      \
        }                                                                               \
    } while (0)

}  // namespace
}  // namespace vbq

static_assert(sizeof(ncclUniqueId) == VBQ_COMM_ID_BYTES, "VBQ_COMM_ID_BYTES must be the size of ncclUniqueId");

extern "C" int vbq_comm_unique_id(void *h_id) {
    using namespace vbq;
    VBQ_REQUIRE(h_id, VBQ_ERR_INVALID_ARGUMENT, "vbq_comm_unique_id: null pointer");
    VBQ_RCCL_OR_FAIL("vbq_comm_unique_id");
    VBQ_RCCL_CHECK("vbq_comm_unique_id", R.GetUniqueId(reinterpret_cast<ncclUniqueId *>(h_id)));
    return VBQ_OK;
}

extern "C" int vbq_comm_init(void **comm, int32_t n_ranks, const void *h_id, int32_t rank) {
    using namespace vbq;
    VBQ_REQUIRE(comm && h_id && n_ranks >= 1 && rank >= 0 && rank < n_ranks, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_comm_init: bad arguments (n_ranks=%d rank=%d)", n_ranks, rank);
    VBQ_RCCL_OR_FAIL("vbq_comm_init");
    ncclComm_t c = nullptr;
    VBQ_RCCL_CHECK("vbq_comm_init", R.CommInitRank(&c, n_ranks, *reinterpret_cast<const ncclUniqueId *>(h_id), rank));
    *comm = c;
    return VBQ_OK;
}

extern "C" int vbq_allreduce_hist(void *comm, void *d_counts, int64_t n, int32_t counts_are_i32, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_allreduce_hist: n < 0");
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(comm && d_counts, VBQ_ERR_INVALID_ARGUMENT, "vbq_allreduce_hist: null pointer");
    VBQ_RCCL_OR_FAIL("vbq_allreduce_hist");
    VBQ_RCCL_CHECK("vbq_allreduce_hist",
                   R.AllReduce(d_counts, d_counts, (size_t)n, counts_are_i32 ? ncclInt32 : ncclInt64, ncclSum,
                               reinterpret_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream)));
    return VBQ_OK;
}

extern "C" int vbq_comm_destroy(void *comm) {
    using namespace vbq;
    if (!comm) return VBQ_OK;
    VBQ_RCCL_OR_FAIL("vbq_comm_destroy");
    VBQ_RCCL_CHECK("vbq_comm_destroy", R.CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
    return VBQ_OK;
}
