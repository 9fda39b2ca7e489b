// rccl_exchange.cpp -- the collectives of the sharded path over RCCL (xGMI between the GPUs of a node, the network across nodes),
// for hosts that are not Python: advantra_cli --ranks N --exchange rccl.  (pnr_amd/multigpu.py reaches RCCL through torch.distributed;
// the default transport between the ranks of ONE host stays the shared-memory all-gather of shm_exchange.cpp, because the records
// that are exchanged live in pinned host memory at both ends.)
//
//   pnr_rccl_allgather          ncclAllGather of one fixed-size block per rank: the pnr_allgather_fn of pnr_trace_replay_sharded (the
//                               per-poll exchange of finished trace records) and of the seed gather
//   pnr_rccl_allreduce_minmax   ncclAllReduce(ncclMax) over (-min, max): the two floats Jmin / Jmax of the z-slab Frangi
//
// The payloads are host data, so every call stages through pinned host memory and a device buffer on the exchange's own stream:
// host -> pinned -> device -> RCCL -> device -> pinned -> host, one stream synchronisation per call.
// librccl is opened at RUN time (dlopen), only when an exchange is opened: libpnr_hip.so has no link-time dependency on it, and a
// process that already holds an RCCL (torch) shares that copy instead of loading a second one.
#include "ctx.h"
#include <dlfcn.h>
#include <cstring>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define PNR_HAVE_RCCL 1
#else
#define PNR_HAVE_RCCL 0
#endif

namespace pnr { void set_error(const char *fmt, ...); }

#if PNR_HAVE_RCCL
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// 0 on success; the message names what is missing
int load_rccl(RcclApi &A)
{
    static RcclApi cached;
    static bool tried = false, ok = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            cached.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (cached.lib) break;
        }
        if (cached.lib) {
            cached.GetUniqueId = (decltype(cached.GetUniqueId))dlsym(cached.lib, "ncclGetUniqueId");
            cached.CommInitRank = (decltype(cached.CommInitRank))dlsym(cached.lib, "ncclCommInitRank");
            cached.CommDestroy = (decltype(cached.CommDestroy))dlsym(cached.lib, "ncclCommDestroy");
            cached.AllGather = (decltype(cached.AllGather))dlsym(cached.lib, "ncclAllGather");
            cached.AllReduce = (decltype(cached.AllReduce))dlsym(cached.lib, "ncclAllReduce");
            cached.GetErrorString = (decltype(cached.GetErrorString))dlsym(cached.lib, "ncclGetErrorString");
            ok = cached.GetUniqueId && cached.CommInitRank && cached.CommDestroy && cached.AllGather && cached.AllReduce && cached.GetErrorString;
        }
    }
    if (!ok) {
        pnr::set_error("RCCL is not available: %s", cached.lib ? "librccl lacks an expected symbol" : dlerror());
        return PNR_E_STATE;
    }
    A = cached;
    return PNR_OK;
}
} // namespace

struct pnr_rccl_exchange {
    RcclApi A;
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    int rank = 0, world = 1, device = 0;
    size_t cap = 0;                // bytes per rank and call
    unsigned char *h_send = nullptr, *h_recv = nullptr; // pinned
    unsigned char *d_send = nullptr, *d_recv = nullptr;
};

#define RX_HIP(call, x)                                                                                                  \
    do {                                                                                                                 \
        hipError_t e_ = (call);                                                                                          \
        if (e_ != hipSuccess) { pnr::set_error("%s failed: %s", #call, hipGetErrorString(e_)); x; return PNR_E_HIP; }    \
    } while (0)
#define RX_NCCL(call, x)                                                                                                 \
    do {                                                                                                                 \
        ncclResult_t r_ = (call);                                                                                        \
        if (r_ != ncclSuccess) { pnr::set_error("%s failed: %s", #call, X->A.GetErrorString(r_)); x; return PNR_E_STATE; } \
    } while (0)

extern "C" {

int pnr_rccl_unique_id(void *id128)
{
    PNR_REQUIRE(id128, PNR_E_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is the 128-byte token the ranks pass around");
    RcclApi A;
    { const int rc = load_rccl(A); if (rc) return rc; }
    ncclUniqueId id;
    const ncclResult_t r = A.GetUniqueId(&id);
    if (r != ncclSuccess) { pnr::set_error("ncclGetUniqueId failed: %s", A.GetErrorString(r)); return PNR_E_STATE; }
    std::memcpy(id128, &id, 128);
    return PNR_OK;
}

void pnr_rccl_exchange_close(pnr_rccl_exchange *X)
{
    if (!X) return;
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess) (void)hipSetDevice(X->device);
    if (X->st) (void)hipStreamSynchronize(X->st);
    if (X->comm) (void)X->A.CommDestroy(X->comm);
    if (X->h_send) (void)hipHostFree(X->h_send);
    if (X->h_recv) (void)hipHostFree(X->h_recv);
    (void)hipFree(X->d_send);
    (void)hipFree(X->d_recv);
    if (X->st) (void)hipStreamDestroy(X->st);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete X;
}

int pnr_rccl_exchange_open(const void *id128, int rank, int world, int device, int64_t capacity_bytes, pnr_rccl_exchange **out)
{
    PNR_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world && capacity_bytes >= 16, PNR_E_ARG, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        pnr::set_error("no HIP device %d: the RCCL exchange needs the rank's GPU", device);
        return PNR_E_NODEVICE;
    }
    pnr_rccl_exchange *X = new pnr_rccl_exchange();
    { const int rc = load_rccl(X->A); if (rc) { delete X; return rc; } }
    X->rank = rank; X->world = world; X->device = device;
    X->cap = (size_t)((capacity_bytes + 15) / 16 * 16);
    RX_HIP(hipSetDevice(device), pnr_rccl_exchange_close(X));
    RX_HIP(hipStreamCreateWithFlags(&X->st, hipStreamNonBlocking), pnr_rccl_exchange_close(X));
    RX_HIP(hipHostMalloc(&X->h_send, X->cap), pnr_rccl_exchange_close(X));
    RX_HIP(hipHostMalloc(&X->h_recv, X->cap * (size_t)world), pnr_rccl_exchange_close(X));
    RX_HIP(hipMalloc(&X->d_send, X->cap), pnr_rccl_exchange_close(X));
    RX_HIP(hipMalloc(&X->d_recv, X->cap * (size_t)world), pnr_rccl_exchange_close(X));
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    RX_NCCL(X->A.CommInitRank(&X->comm, world, id, rank), pnr_rccl_exchange_close(X));
    *out = X;
    return PNR_OK;
}

// pnr_allgather_fn: recv = world x bytes_per_rank, rank order
int pnr_rccl_allgather(void *user, const void *send, void *recv, int64_t bytes_per_rank)
{
    pnr_rccl_exchange *X = (pnr_rccl_exchange *)user;
    PNR_REQUIRE(X && send && recv && bytes_per_rank >= 0, PNR_E_ARG, "bad argument");
    PNR_REQUIRE((size_t)bytes_per_rank <= X->cap, PNR_E_ARG, "RCCL exchange opened for %zu bytes per rank, asked for %lld", X->cap, (long long)bytes_per_rank);
    if (bytes_per_rank == 0) return PNR_OK;
    const size_t n = (size_t)bytes_per_rank;
    std::memcpy(X->h_send, send, n);
    RX_HIP(hipMemcpyAsync(X->d_send, X->h_send, n, hipMemcpyHostToDevice, X->st), );
    RX_NCCL(X->A.AllGather(X->d_send, X->d_recv, n, ncclUint8, X->comm, X->st), );
    RX_HIP(hipMemcpyAsync(X->h_recv, X->d_recv, n * (size_t)X->world, hipMemcpyDeviceToHost, X->st), );
    RX_HIP(hipStreamSynchronize(X->st), );
    std::memcpy(recv, X->h_recv, n * (size_t)X->world);
    return PNR_OK;
}

// (*mn, *mx) <- (min over the ranks, max over the ranks): one ncclAllReduce(ncclMax) over (-min, max)
int pnr_rccl_allreduce_minmax(pnr_rccl_exchange *X, float *mn, float *mx)
{
    PNR_REQUIRE(X && mn && mx, PNR_E_ARG, "null argument");
    float v[2] = {-*mn, *mx};
    std::memcpy(X->h_send, v, sizeof(v));
    RX_HIP(hipMemcpyAsync(X->d_send, X->h_send, sizeof(v), hipMemcpyHostToDevice, X->st), );
    RX_NCCL(X->A.AllReduce(X->d_send, X->d_recv, 2, ncclFloat32, ncclMax, X->comm, X->st), );
    RX_HIP(hipMemcpyAsync(X->h_recv, X->d_recv, sizeof(v), hipMemcpyDeviceToHost, X->st), );
    RX_HIP(hipStreamSynchronize(X->st), );
    std::memcpy(v, X->h_recv, sizeof(v));
    *mn = -v[0];
    *mx = v[1];
    return PNR_OK;
}

} // extern "C"

#else // no rccl.h at build time: the entry points exist and say so

struct pnr_rccl_exchange { int unused; };
extern "C" {
int pnr_rccl_unique_id(void *) { pnr::set_error("built without rccl.h: no RCCL exchange"); return PNR_E_STATE; }
int pnr_rccl_exchange_open(const void *, int, int, int, int64_t, pnr_rccl_exchange **out) { if (out) *out = nullptr; pnr::set_error("built without rccl.h: no RCCL exchange"); return PNR_E_STATE; }
int pnr_rccl_allgather(void *, const void *, void *, int64_t) { pnr::set_error("built without rccl.h: no RCCL exchange"); return PNR_E_STATE; }
int pnr_rccl_allreduce_minmax(pnr_rccl_exchange *, float *, float *) { pnr::set_error("built without rccl.h: no RCCL exchange"); return PNR_E_STATE; }
void pnr_rccl_exchange_close(pnr_rccl_exchange *) {}
}
#endif
