// Data-parallel training inside the library: the gradient all-reduce between bamd_fwd_bwd and bamd_adam_step as ONE ncclAllReduce on
// the caller's stream (include/baler_amd.h, "data-parallel training inside the library").  gfx950 only; RCCL over xGMI.
//
// RCCL is NOT a link-time dependency: a PyTorch process already carries a librccl (torch/lib/librccl.so) and a second copy in the same
// process would be a second set of IPC / proxy threads.  The entry points are resolved on first use: the library the process has
// already loaded (its SONAME librccl.so.1 resolves to the loaded object), else the system one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "bamd_internal.hpp"

namespace bamd {
namespace {

struct Rccl {
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = nullptr;
        const char *override_path = getenv("BALER_AMD_RCCL_LIB");
        const char *names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !n[0]) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { r.why = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char *name) {
            void *p = dlsym(lib, name);
            if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + name;
            return p;
        };
        r.get_unique_id = (decltype(r.get_unique_id))sym("ncclGetUniqueId");
        r.comm_init_rank = (decltype(r.comm_init_rank))sym("ncclCommInitRank");
        r.comm_destroy = (decltype(r.comm_destroy))sym("ncclCommDestroy");
        r.comm_count = (decltype(r.comm_count))sym("ncclCommCount");
        r.all_reduce = (decltype(r.all_reduce))sym("ncclAllReduce");
        r.error_string = (decltype(r.error_string))sym("ncclGetErrorString");
        r.ok = r.why.empty();
    });
    return r;
}

int need_rccl(const Rccl *&out) {
    const Rccl &r = rccl();
    if (!r.ok) { set_error("data-parallel entry point: " + r.why); return BAMD_ERR_UNSUPPORTED; }
    out = &r;
    return BAMD_OK;
}

#define BAMD_NCCL(r, call)                                                                               \
    do {                                                                                                 \
        const ncclResult_t e_ = (call);                                                                  \
        if (e_ != ncclSuccess) {                                                                         \
            set_error(std::string(#call) + ": " + ((r)->error_string ? (r)->error_string(e_) : "?"));    \
            return BAMD_ERR_HIP;                                                                         \
        }                                                                                                \
    } while (0)

struct DevGuard {      // as api.hip's: the communicator lives on the handle's device
    int prev = -1, rc = hipSuccess;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) rc = (int)hipSetDevice(dev); else prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

int comm_allreduce_sum(bamd_handle *h, void *buf, int dtype, int64_t count, hipStream_t s) {
    const Rccl *r = nullptr;
    if (int rc = need_rccl(r)) return rc;
    if (!h->comm) { set_error("no communicator attached to the handle"); return BAMD_ERR_INVALID; }
    BAMD_NCCL(r, r->all_reduce(buf, buf, (size_t)count, dtype == BAMD_F64 ? ncclFloat64 : ncclFloat32, ncclSum, (ncclComm_t)h->comm, s));
    return BAMD_OK;
}

void comm_teardown(bamd_handle *h) {
    if (h->comm && h->comm_owned && rccl().ok) (void)rccl().comm_destroy((ncclComm_t)h->comm);
    h->comm = nullptr;
    h->comm_owned = false;
    h->comm_world = 0;
}

}  // namespace bamd

using namespace bamd;

extern "C" {

int bamd_comm_unique_id(void *id128) {
    BAMD_REQUIRE(id128, "null argument");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(r)) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    ncclUniqueId id;
    BAMD_NCCL(r, r->get_unique_id(&id));
    memcpy(id128, &id, sizeof(id));
    return BAMD_OK;
}

int bamd_comm_init(bamd_handle *h, const void *id128, int rank, int world) {
    BAMD_REQUIRE(h && id128, "null argument");
    BAMD_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(r)) return rc;
    DevGuard guard(h->device);
    BAMD_REQUIRE(guard.rc == hipSuccess, "cannot select the handle's device");
    comm_teardown(h);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    BAMD_NCCL(r, r->comm_init_rank(&comm, world, id, rank));
    h->comm = comm;
    h->comm_owned = true;
    h->comm_world = world;
    return BAMD_OK;
}

int bamd_comm_attach(bamd_handle *h, void *comm, int world) {
    BAMD_REQUIRE(h && comm, "null argument");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(r)) return rc;
    comm_teardown(h);
    int n = world;
    if (n <= 0) BAMD_NCCL(r, r->comm_count((ncclComm_t)comm, &n));
    h->comm = comm;
    h->comm_owned = false;
    h->comm_world = n;
    return BAMD_OK;
}

int bamd_comm_release(bamd_handle *h) {
    BAMD_REQUIRE(h, "null handle");
    DevGuard guard(h->device);
    comm_teardown(h);
    return BAMD_OK;
}

int bamd_comm_world(const bamd_handle *h) { return h ? h->comm_world : 0; }

int bamd_allreduce_sum(bamd_handle *h, void *buf, int dtype, int64_t count, void *stream) {
    BAMD_REQUIRE(h && buf && count >= 0, "bad arguments");
    BAMD_REQUIRE(dtype == BAMD_F32 || dtype == BAMD_F64, "bad dtype");
    DevGuard guard(h->device);
    BAMD_REQUIRE(guard.rc == hipSuccess, "cannot select the handle's device");
    return comm_allreduce_sum(h, buf, dtype, count, (hipStream_t)stream);
}

}  // extern "C"
