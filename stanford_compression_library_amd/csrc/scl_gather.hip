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

// The RCCL entry points as librccl.so exports them ...
struct RcclReal {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclReal g_real;
// ... and the table the code below calls through: the same operations with plain-C signatures (include/scl_hip.h,
// scl_rccl_api), filled with adapters over g_real by rccl_load() -- or with a caller's stand-in by scl_rccl_inject_api(),
// which is how the tests drive the W > 1 branches of the exchange without a second GPU.
static scl_rccl_api g_rccl;
static bool g_rccl_ready = false, g_rccl_injected = false;
static std::mutex g_rccl_lock;
static inline bool rccl_host_mode() { return g_rccl_injected && g_rccl.host_memory != 0; }

static int ad_get_unique_id(uint8_t *id128) {
    ncclUniqueId id;
    const int r = g_real.GetUniqueId(&id);
    if (r == 0) ::memcpy(id128, id.internal, 128);
    return r;
}
static int ad_comm_init_rank(void **comm, int world, const uint8_t *id128, int rank) {
    ncclUniqueId id;
    ::memcpy(id.internal, id128, 128);
    return g_real.CommInitRank((ncclComm_t *)comm, world, id, rank);
}
static int ad_comm_destroy(void *comm) { return g_real.CommDestroy((ncclComm_t)comm); }
static int ad_comm_count(void *comm, int *n) { return g_real.CommCount((ncclComm_t)comm, n); }
static int ad_comm_user_rank(void *comm, int *r) { return g_real.CommUserRank((ncclComm_t)comm, r); }
static int ad_all_gather(const void *s, void *r, uint64_t n, int dt, void *comm, void *st) {
    return g_real.AllGather(s, r, (size_t)n, dt, (ncclComm_t)comm, (hipStream_t)st);
}
static int ad_send(const void *b, uint64_t n, int dt, int peer, void *comm, void *st) {
    return g_real.Send(b, (size_t)n, dt, peer, (ncclComm_t)comm, (hipStream_t)st);
}
static int ad_recv(void *b, uint64_t n, int dt, int peer, void *comm, void *st) {
    return g_real.Recv(b, (size_t)n, dt, peer, (ncclComm_t)comm, (hipStream_t)st);
}
static int ad_group_start(void) { return g_real.GroupStart(); }
static int ad_group_end(void) { return g_real.GroupEnd(); }
static const char *ad_error_string(int r) { return g_real.GetErrorString(r); }

static int rccl_load() {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    if (g_rccl_ready) return SCL_OK;
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
    // resolved into a local table and published only when complete: a half-filled table is never visible
    RcclReal api;
#define SCL_RCCL_SYM(field, name)                                      \
    *(void **)(&api.field) = dlsym(h, name);                           \
    if (!api.field) {                                                  \
        scl_set_error("rccl: symbol %s missing from librccl.so", name); \
        dlclose(h);                                                    \
        return SCL_E_NODEVICE;                                         \
    }
    SCL_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    SCL_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    SCL_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    SCL_RCCL_SYM(CommCount, "ncclCommCount")
    SCL_RCCL_SYM(CommUserRank, "ncclCommUserRank")
    SCL_RCCL_SYM(AllGather, "ncclAllGather")
    SCL_RCCL_SYM(Send, "ncclSend")
    SCL_RCCL_SYM(Recv, "ncclRecv")
    SCL_RCCL_SYM(GroupStart, "ncclGroupStart")
    SCL_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    SCL_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef SCL_RCCL_SYM
    api.handle = h;
    g_real = api;
    scl_rccl_api t;
    ::memset(&t, 0, sizeof(t));
    t.get_unique_id = ad_get_unique_id;
    t.comm_init_rank = ad_comm_init_rank;
    t.comm_destroy = ad_comm_destroy;
    t.comm_count = ad_comm_count;
    t.comm_user_rank = ad_comm_user_rank;
    t.all_gather = ad_all_gather;
    t.send = ad_send;
    t.recv = ad_recv;
    t.group_start = ad_group_start;
    t.group_end = ad_group_end;
    t.error_string = ad_error_string;
    t.host_memory = 0;
    g_rccl = t;
    g_rccl_ready = true;
    return SCL_OK;
}

// Test hook (include/scl_hip.h): replaces the table -- every later scl_rccl_* / scl_streams_gather*_rccl call goes through
// `api` instead of librccl.so; NULL restores the real library (loaded again on next use).  With api->host_memory != 0
// the "device" buffers of those calls are host memory and no HIP call is made: the library's own copies become memcpy
// and the root's offset fix-up a host loop, so the layout logic of a W-rank exchange can run on a machine without a GPU.
// Communicators created under one table must be destroyed under the same one.
extern "C" int scl_rccl_inject_api(const scl_rccl_api *api) {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    if (!api) {
        g_rccl_injected = false;
        g_rccl_ready = false;
        return SCL_OK;
    }
    SCL_REQUIRE(api->get_unique_id && api->comm_init_rank && api->comm_destroy && api->comm_count &&
                    api->comm_user_rank && api->all_gather && api->send && api->recv && api->group_start &&
                    api->group_end && api->error_string,
                "rccl_inject_api: every entry of the table must be set");
    g_rccl = *api;
    g_rccl_injected = true;
    g_rccl_ready = true;
    return SCL_OK;
}

#define SCL_RCCL_TRY(expr)                                                                          \
    do {                                                                                            \
        int _r = (expr);                                                                            \
        if (_r != 0) {                                                                              \
            scl_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.error_string(_r), __FILE__, __LINE__); \
            return SCL_E_HIP;                                                                       \
        }                                                                                           \
    } while (0)

// copies the library itself makes around the exchange: HIP copies on `st`, or memcpy under an injected host-memory table
static int gx_copy(void *dst, const void *src, u64 n, hipMemcpyKind kind, hipStream_t st) {
    if (rccl_host_mode()) {
        ::memcpy(dst, src, n);
        return SCL_OK;
    }
    SCL_HIP_TRY(hipMemcpyAsync(dst, src, n, kind, st));
    return SCL_OK;
}
static int gx_check_device(int device, const char *what) { return rccl_host_mode() ? SCL_OK : scl_check_device(device, what); }

struct scl_comm {
    void *comm;
    int rank, world, device;
    u64 *d_sizes;  // [world + 1]: entry `world` is this rank's own count (send buffer of the all-gather)
    u64 *h_sizes;  // pinned, [world + 1]: entry `world` stages this rank's own count
    int host;      // created under an injected host-memory table (scl_rccl_inject_api): buffers are malloc'ed
};

extern "C" int scl_rccl_unique_id(uint8_t *id128) {
    SCL_REQUIRE(id128, "rccl_unique_id: null pointer");
    if (int rc = rccl_load()) return rc;
    SCL_RCCL_TRY(g_rccl.get_unique_id(id128));
    return SCL_OK;
}

extern "C" int scl_rccl_comm_create(const uint8_t *id128, int rank, int world, scl_comm **out) {
    SCL_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "rccl_comm_create: bad arguments");
    *out = nullptr;
    if (int rc = rccl_load()) return rc;
    const bool host = rccl_host_mode();
    scl_comm *c = new scl_comm();
    c->rank = rank;
    c->world = world;
    c->device = host ? -1 : scl_current_device();
    c->d_sizes = nullptr;
    c->h_sizes = nullptr;
    c->comm = nullptr;
    const int r = g_rccl.comm_init_rank(&c->comm, world, id128, rank);
    if (r != 0) {
        scl_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.error_string(r));
        delete c;
        return SCL_E_HIP;
    }
    if (host) {
        c->d_sizes = (u64 *)::calloc(world + 1, sizeof(u64));
        c->h_sizes = (u64 *)::calloc(world + 1, sizeof(u64));
        if (!c->d_sizes || !c->h_sizes) {
            scl_set_error("rccl_comm_create: buffer allocation failed");
            (void)g_rccl.comm_destroy(c->comm);
            ::free(c->d_sizes);
            ::free(c->h_sizes);
            delete c;
            return SCL_E_ALLOC;
        }
        c->host = 1;
        *out = c;
        return SCL_OK;
    }
    hipError_t e = hipMalloc((void **)&c->d_sizes, (world + 1) * sizeof(u64));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_sizes, (world + 1) * sizeof(u64), hipHostMallocDefault);
    if (e != hipSuccess) {
        scl_set_error("rccl_comm_create: buffer allocation failed: %s", hipGetErrorString(e));
        (void)g_rccl.comm_destroy(c->comm);
        if (c->d_sizes) (void)hipFree(c->d_sizes);
        delete c;
        return SCL_E_ALLOC;
    }
    *out = c;
    return SCL_OK;
}

extern "C" void scl_rccl_comm_destroy(scl_comm *c) {
    if (!c) return;
    if (g_rccl_ready) (void)g_rccl.comm_destroy(c->comm);
    if (c->host) {
        ::free(c->d_sizes);
        ::free(c->h_sizes);
    } else {
        if (c->d_sizes) (void)hipFree(c->d_sizes);
        if (c->h_sizes) (void)hipHostFree(c->h_sizes);
    }
    delete c;
}

// What the COMMUNICATOR says about itself (ncclCommUserRank / ncclCommCount -- not the numbers it was created with) and the
// device it belongs to; bench.py prints these so that a multi-GPU line shows how many ranks really took part.
extern "C" int scl_rccl_comm_info(scl_comm *c, int *rank, int *nranks, int *device) {
    SCL_REQUIRE(c, "rccl_comm_info: null communicator");
    int r = -1, n = -1;
    SCL_RCCL_TRY(g_rccl.comm_user_rank(c->comm, &r));
    SCL_RCCL_TRY(g_rccl.comm_count(c->comm, &n));
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    if (device) *device = c->device;
    return SCL_OK;
}

// Collective: every rank contributes one u64; h_out[world] (host) receives all of them in rank order on every rank.
// Synchronises `stream` (the values must reach the host).  Used for the byte counts of a gather, so that the root can
// size its buffer before any transfer is posted.
extern "C" int scl_rccl_allgather_u64(scl_comm *c, uint64_t value, uint64_t *h_out, void *stream) {
    SCL_REQUIRE(c && h_out, "rccl_allgather_u64: bad arguments");
    if (int rc = gx_check_device(c->device, "rccl_allgather_u64")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int W = c->world;
    c->h_sizes[W] = value;
    if (int rc = gx_copy(c->d_sizes + W, c->h_sizes + W, sizeof(u64), hipMemcpyHostToDevice, st)) return rc;
    SCL_RCCL_TRY(g_rccl.all_gather(c->d_sizes + W, c->d_sizes, 1, scl_ncclUint64, c->comm, stream));
    if (int rc = gx_copy(c->h_sizes, c->d_sizes, W * sizeof(u64), hipMemcpyDeviceToHost, st)) return rc;
    if (!c->host) SCL_HIP_TRY(hipStreamSynchronize(st));
    ::memcpy(h_out, c->h_sizes, W * sizeof(u64));
    return SCL_OK;
}

// Collective, asynchronous on `stream`, device to device: every rank contributes n_u64 values at d_in; d_out
// [world * n_u64] receives all of them in rank order on every rank.  Nothing here waits for the host: the overlapped
// pipeline of configs[4] queues it behind a sub-batch's compaction and reads the sizes back with an event.
extern "C" int scl_rccl_allgather_async(scl_comm *c, const uint64_t *d_in, uint64_t *d_out, uint64_t n_u64, void *stream) {
    SCL_REQUIRE(c && d_in && d_out && n_u64 >= 1, "rccl_allgather_async: bad arguments");
    if (int rc = gx_check_device(c->device, "rccl_allgather_async")) return rc;
    if (c->world == 1) {  // a one-rank all-gather is a copy: no collective kernel for it
        if (d_in != d_out)
            if (int rc = gx_copy(d_out, d_in, n_u64 * sizeof(u64), hipMemcpyDeviceToDevice, (hipStream_t)stream)) return rc;
        return SCL_OK;
    }
    SCL_RCCL_TRY(g_rccl.all_gather(d_in, d_out, n_u64, scl_ncclUint64, c->comm, stream));
    return SCL_OK;
}

// n_parts variable-length gathers in ONE grouped exchange, asynchronous on `stream`: for part p, rank r's
// h_send_bytes[p] bytes at d_send[p] arrive at the root's d_recv[p] + h_rank_offsets[p * (world + 1) + r].
// h_rank_offsets holds, per part, the layout every rank agreed on beforehand (exclusive prefix sum of the ranks'
// counts, last entry = total); d_recv matters on the root only.  A payload and its per-chunk offset table travel as
// two parts of one call.
//
// Failure behaviour: every argument is validated BEFORE anything is posted; a layout that disagrees with this rank's
// own count is refused on this rank only -- its peers then wait in the exchange (a collective cannot be refused
// unilaterally), so callers must derive the layout from exchanged counts (scl_rccl_allgather_*), never from guesses.
// Once ncclGroupStart has succeeded, ncclGroupEnd is called on every path: the first error is recorded, the group
// is closed, then the error is returned -- the thread never stays inside an open group.
extern "C" int scl_streams_gatherv_rccl(scl_comm *c, int root, uint32_t n_parts, const uint8_t *const *d_send,
                                        const uint64_t *h_send_bytes, uint8_t *const *d_recv,
                                        const uint64_t *h_rank_offsets, void *stream) {
    SCL_REQUIRE(c && h_rank_offsets && d_send && h_send_bytes && root >= 0 && root < c->world && n_parts >= 1 &&
                    n_parts <= 64,
                "streams_gatherv_rccl: bad arguments");
    if (int rc = gx_check_device(c->device, "streams_gatherv_rccl")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int W = c->world;
    for (u32 p = 0; p < n_parts; ++p) {
        const u64 *offs = h_rank_offsets + (u64)p * (W + 1);
        SCL_REQUIRE(d_send[p] || h_send_bytes[p] == 0, "streams_gatherv_rccl: part %u has bytes but no send buffer", p);
        SCL_REQUIRE(offs[c->rank + 1] - offs[c->rank] == h_send_bytes[p],
                    "streams_gatherv_rccl: part %u: rank %d sends %llu bytes but the agreed layout gives it %llu", p,
                    c->rank, (unsigned long long)h_send_bytes[p], (unsigned long long)(offs[c->rank + 1] - offs[c->rank]));
        SCL_REQUIRE(c->rank != root || offs[W] == 0 || (d_recv && d_recv[p]),
                    "streams_gatherv_rccl: the root needs a receive buffer for part %u", p);
    }
    // one grouped exchange: every non-root rank sends its parts, the root posts one receive per sender and part
    // (a one-rank communicator has nothing to post: only the root's own copies below)
    int first = 0;
    const char *what = "";
    if (W > 1) SCL_RCCL_TRY(g_rccl.group_start());
    for (u32 p = 0; p < n_parts && first == 0 && W > 1; ++p) {
        const u64 *offs = h_rank_offsets + (u64)p * (W + 1);
        if (c->rank == root) {
            for (int r = 0; r < W && first == 0; ++r) {
                const u64 nb = offs[r + 1] - offs[r];
                if (r != root && nb) {
                    first = g_rccl.recv(d_recv[p] + offs[r], nb, scl_ncclUint8, r, c->comm, stream);
                    what = "ncclRecv";
                }
            }
        } else if (h_send_bytes[p]) {
            first = g_rccl.send(d_send[p], h_send_bytes[p], scl_ncclUint8, root, c->comm, stream);
            what = "ncclSend";
        }
    }
    const int end = W > 1 ? g_rccl.group_end() : 0;  // always: never leave the thread's group open
    if (first != 0 || end != 0) {
        scl_set_error("streams_gatherv_rccl: %s failed: %s", first != 0 ? what : "ncclGroupEnd",
                      g_rccl.error_string(first != 0 ? first : end));
        return SCL_E_HIP;
    }
    if (c->rank == root)  // the root's own share: device copies on the same stream
        for (u32 p = 0; p < n_parts; ++p)
            if (h_send_bytes[p])
                if (int rc = gx_copy(d_recv[p] + h_rank_offsets[(u64)p * (W + 1) + root], d_send[p], h_send_bytes[p],
                                     hipMemcpyDeviceToDevice, st))
                    return rc;
    return SCL_OK;
}

// configs[4]'s exchange as ONE call: a sub-batch's dense payload AND its per-chunk offset table travel to the root in
// one grouped exchange, and the root turns the per-rank offset tables into the GLOBAL one (rank r's entries shifted by
// the bytes of the ranks before it, last entry = grand total) with one small kernel on the same stream -- so the
// root ends up with exactly what a single process would have produced for the ranks' chunks in rank order.
//   d_payload / payload_bytes : this rank's dense streams (scl_streams_compact output)
//   d_offsets                 : this rank's n_chunks + 1 record offsets (u64, the compaction's offset table)
//   h_bytes_by_rank / h_chunks_by_rank [world] : every rank's counts, as exchanged by scl_rccl_allgather_*
//   d_recv_payload [sum bytes], d_recv_offsets [sum chunks + 1] : on the root only
struct GatherFix {
    u64 cbase[65];  // exclusive prefix of the chunk counts
    u64 bbase[65];  // exclusive prefix of the byte counts
    u32 world;
};
__global__ void gather_fix_offsets(u64 *__restrict__ goffs, GatherFix f) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n = f.cbase[f.world];
    if (idx > n) return;
    if (idx == n) {
        goffs[idx] = f.bbase[f.world];
        return;
    }
    u32 r = 0;
    while (r + 1 < f.world && idx >= f.cbase[r + 1]) ++r;
    if (r) goffs[idx] += f.bbase[r];
}

extern "C" int scl_streams_gather_blocks_rccl(scl_comm *c, int root, const uint8_t *d_payload, uint64_t payload_bytes,
                                              const uint64_t *d_offsets, uint64_t n_chunks, uint8_t *d_recv_payload,
                                              uint64_t *d_recv_offsets, const uint64_t *h_bytes_by_rank,
                                              const uint64_t *h_chunks_by_rank, void *stream) {
    SCL_REQUIRE(c && h_bytes_by_rank && h_chunks_by_rank && root >= 0 && root < c->world && c->world <= 64,
                "streams_gather_blocks_rccl: bad arguments (at most 64 ranks)");
    const int W = c->world;
    SCL_REQUIRE(h_bytes_by_rank[c->rank] == payload_bytes && h_chunks_by_rank[c->rank] == n_chunks,
                "streams_gather_blocks_rccl: rank %d's own counts (%llu bytes, %llu chunks) differ from the exchanged ones",
                c->rank, (unsigned long long)payload_bytes, (unsigned long long)n_chunks);
    SCL_REQUIRE(d_offsets || n_chunks == 0, "streams_gather_blocks_rccl: null offset table");
    GatherFix f;
    u64 offs[2][65];
    f.world = (u32)W;
    f.cbase[0] = f.bbase[0] = 0;
    for (int r = 0; r < W; ++r) {
        f.cbase[r + 1] = f.cbase[r] + h_chunks_by_rank[r];
        f.bbase[r + 1] = f.bbase[r] + h_bytes_by_rank[r];
    }
    for (int r = 0; r <= W; ++r) {
        offs[0][r] = f.bbase[r];
        offs[1][r] = 8 * f.cbase[r];
    }
    if (W == 1) {  // nothing to exchange: two device copies (the offset table is already global)
        hipStream_t st = (hipStream_t)stream;
        if (int rc = gx_check_device(c->device, "streams_gather_blocks_rccl")) return rc;
        SCL_REQUIRE((d_recv_payload || payload_bytes == 0) && d_recv_offsets, "streams_gather_blocks_rccl: the root needs receive buffers");
        if (payload_bytes)
            if (int rc = gx_copy(d_recv_payload, d_payload, payload_bytes, hipMemcpyDeviceToDevice, st)) return rc;
        return gx_copy(d_recv_offsets, d_offsets, 8 * (n_chunks + 1), hipMemcpyDeviceToDevice, st);
    }
    // two parts, laid out [part][world + 1]
    u64 flat[2 * 65];
    for (int r = 0; r <= W; ++r) {
        flat[r] = offs[0][r];
        flat[(W + 1) + r] = offs[1][r];
    }
    const uint8_t *send[2] = {d_payload, (const uint8_t *)d_offsets};
    const u64 nbytes[2] = {payload_bytes, 8 * n_chunks};
    uint8_t *recv[2] = {d_recv_payload, (uint8_t *)d_recv_offsets};
    SCL_REQUIRE(c->rank != root || d_recv_offsets, "streams_gather_blocks_rccl: the root needs receive buffers");
    if (int rc = scl_streams_gatherv_rccl(c, root, 2, send, nbytes, recv, flat, stream)) return rc;
    if (c->rank == root) {
        const u64 n = f.cbase[W] + 1;
        if (c->host) {  // injected host-memory table: what gather_fix_offsets does, as a loop
            for (int r = 1; r < W; ++r)
                for (u64 i = f.cbase[r]; i < f.cbase[r + 1]; ++i) d_recv_offsets[i] += f.bbase[r];
            d_recv_offsets[f.cbase[W]] = f.bbase[W];
        } else {
            hipLaunchKernelGGL(gather_fix_offsets, dim3((u32)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_recv_offsets, f);
            SCL_HIP_TRY(hipGetLastError());
        }
    }
    return SCL_OK;
}

// The one-part form (ABI version 2 signature): rank r's send_bytes bytes at d_send arrive at the root's
// d_recv + h_rank_offsets[r].
extern "C" int scl_streams_gather_rccl(scl_comm *c, int root, const uint8_t *d_send, uint64_t send_bytes, uint8_t *d_recv,
                                       const uint64_t *h_rank_offsets, void *stream) {
    return scl_streams_gatherv_rccl(c, root, 1, &d_send, &send_bytes, &d_recv, h_rank_offsets, stream);
}
