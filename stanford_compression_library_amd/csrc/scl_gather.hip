// scl_gather.hip -- the one exchange step of the sharded path (BASELINE.json configs[4]): a
// variable-length gather of every rank's compacted streams to one rank, over RCCL.
//
// No reference counterpart (the reference has no communication of any kind).  Pattern: the
// ranks' byte counts travel first (ncclAllGather of one u64 per rank), then one grouped ncclSend / ncclRecv per
// non-root rank straight into the root's buffer at the prefix offsets.  xGMI is a full mesh of point-to-point links, so
// the root receives on all its links at once (7 x ~153 GB/s on an 8-GPU node); no ring, no reduction.
//
// RCCL is loaded at run time (dlopen, preferring a librccl that is already mapped -- e.g. the one PyTorch ships -- so
// that a process never ends up with two RCCL instances talking to the same devices); libscl_hip.so itself has no
// link-time dependency on it, and single-GPU users never touch it.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "scl_common.h"

typedef int ncclResult_t;  // rccl.h: ncclSuccess == 0
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;  // NCCL_UNIQUE_ID_BYTES
enum { scl_ncclUint8 = 1, scl_ncclUint64 = 5 };  // rccl.h ncclDataType_t

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_lock;

static int rccl_load() {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    if (g_rccl.handle) return SCL_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names)  // one that is already in the process first
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (const char *n : names) {
        if (h) break;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        scl_set_error("rccl: librccl.so not found (%s)", dlerror());
        return SCL_E_NODEVICE;
    }
#define SCL_RCCL_SYM(field, name)                                      \
    *(void **)(&g_rccl.field) = dlsym(h, name);                        \
    if (!g_rccl.field) {                                               \
        scl_set_error("rccl: symbol %s missing from librccl.so", name); \
        return SCL_E_NODEVICE;                                         \
    }
    SCL_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    SCL_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    SCL_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    SCL_RCCL_SYM(AllGather, "ncclAllGather")
    SCL_RCCL_SYM(Send, "ncclSend")
    SCL_RCCL_SYM(Recv, "ncclRecv")
    SCL_RCCL_SYM(GroupStart, "ncclGroupStart")
    SCL_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    SCL_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef SCL_RCCL_SYM
    g_rccl.handle = h;
    return SCL_OK;
}

#define SCL_RCCL_TRY(expr)                                                                            \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != 0) {                                                                                \
            scl_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return SCL_E_HIP;                                                                         \
        }                                                                                             \
    } while (0)

struct scl_comm {
    ncclComm_t comm;
    int rank, world, device;
    u64 *d_sizes;  // [world + 1]: entry `world` is this rank's own count (send buffer of the all-gather)
    u64 *h_sizes;  // pinned, [world + 1]: entry `world` stages this rank's own count
};

extern "C" int scl_rccl_unique_id(uint8_t *id128) {
    SCL_REQUIRE(id128, "rccl_unique_id: null pointer");
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    SCL_RCCL_TRY(g_rccl.GetUniqueId(&id));
    ::memcpy(id128, id.internal, 128);
    return SCL_OK;
}

extern "C" int scl_rccl_comm_create(const uint8_t *id128, int rank, int world, scl_comm **out) {
    SCL_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "rccl_comm_create: bad arguments");
    *out = nullptr;
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    ::memcpy(id.internal, id128, 128);
    scl_comm *c = new scl_comm();
    c->rank = rank;
    c->world = world;
    c->device = scl_current_device();
    c->d_sizes = nullptr;
    c->h_sizes = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) {
        scl_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return SCL_E_HIP;
    }
    hipError_t e = hipMalloc((void **)&c->d_sizes, (world + 1) * sizeof(u64));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_sizes, (world + 1) * sizeof(u64), hipHostMallocDefault);
    if (e != hipSuccess) {
        scl_set_error("rccl_comm_create: buffer allocation failed: %s", hipGetErrorString(e));
        (void)g_rccl.CommDestroy(c->comm);
        if (c->d_sizes) (void)hipFree(c->d_sizes);
        delete c;
        return SCL_E_ALLOC;
    }
    *out = c;
    return SCL_OK;
}

extern "C" void scl_rccl_comm_destroy(scl_comm *c) {
    if (!c) return;
    if (g_rccl.handle) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_sizes) (void)hipFree(c->d_sizes);
    if (c->h_sizes) (void)hipHostFree(c->h_sizes);
    delete c;
}

// Collective: every rank contributes one u64; h_out[world] (host) receives all of them in rank order on every rank.
// Synchronises `stream` (the values must reach the host).  Used for the byte counts of a gather, so that the root can
// size its buffer before any transfer is posted.
extern "C" int scl_rccl_allgather_u64(scl_comm *c, uint64_t value, uint64_t *h_out, void *stream) {
    SCL_REQUIRE(c && h_out, "rccl_allgather_u64: bad arguments");
    if (int rc = scl_check_device(c->device, "rccl_allgather_u64")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int W = c->world;
    c->h_sizes[W] = value;
    SCL_HIP_TRY(hipMemcpyAsync(c->d_sizes + W, c->h_sizes + W, sizeof(u64), hipMemcpyHostToDevice, st));
    SCL_RCCL_TRY(g_rccl.AllGather(c->d_sizes + W, c->d_sizes, 1, scl_ncclUint64, c->comm, st));
    SCL_HIP_TRY(hipMemcpyAsync(c->h_sizes, c->d_sizes, W * sizeof(u64), hipMemcpyDeviceToHost, st));
    SCL_HIP_TRY(hipStreamSynchronize(st));
    ::memcpy(h_out, c->h_sizes, W * sizeof(u64));
    return SCL_OK;
}

// Collective, asynchronous on `stream`: rank r's send_bytes bytes at d_send arrive at the root's
// d_recv + h_rank_offsets[r].  h_rank_offsets[world + 1] (host) is the layout every rank agreed on beforehand
// (exclusive prefix sum of the counts from scl_rccl_allgather_u64, last entry = total): h_rank_offsets[r + 1] -
// h_rank_offsets[r] must equal rank r's send_bytes.  d_recv matters on the root only.
extern "C" int scl_streams_gather_rccl(scl_comm *c, int root, const uint8_t *d_send, uint64_t send_bytes, uint8_t *d_recv,
                                       const uint64_t *h_rank_offsets, void *stream) {
    SCL_REQUIRE(c && h_rank_offsets && root >= 0 && root < c->world && (d_send || send_bytes == 0),
                "streams_gather_rccl: bad arguments");
    if (int rc = scl_check_device(c->device, "streams_gather_rccl")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int W = c->world;
    SCL_REQUIRE(h_rank_offsets[c->rank + 1] - h_rank_offsets[c->rank] == send_bytes,
                "streams_gather_rccl: rank %d sends %llu bytes but the agreed layout gives it %llu", c->rank,
                (unsigned long long)send_bytes,
                (unsigned long long)(h_rank_offsets[c->rank + 1] - h_rank_offsets[c->rank]));
    SCL_REQUIRE(c->rank != root || d_recv || h_rank_offsets[W] == 0, "streams_gather_rccl: the root needs a receive buffer");
    // one grouped exchange: every non-root rank sends, the root posts one receive per sender
    SCL_RCCL_TRY(g_rccl.GroupStart());
    if (c->rank == root) {
        for (int r = 0; r < W; ++r) {
            const u64 nb = h_rank_offsets[r + 1] - h_rank_offsets[r];
            if (r != root && nb)
                SCL_RCCL_TRY(g_rccl.Recv(d_recv + h_rank_offsets[r], nb, scl_ncclUint8, r, c->comm, st));
        }
    } else if (send_bytes) {
        SCL_RCCL_TRY(g_rccl.Send(d_send, send_bytes, scl_ncclUint8, root, c->comm, st));
    }
    SCL_RCCL_TRY(g_rccl.GroupEnd());
    if (c->rank == root && send_bytes)  // the root's own share: a device copy on the same stream
        SCL_HIP_TRY(hipMemcpyAsync(d_recv + h_rank_offsets[root], d_send, send_bytes, hipMemcpyDeviceToDevice, st));
    return SCL_OK;
}
