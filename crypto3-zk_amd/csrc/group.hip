// Device group (include/zkhip.h, "device group"): N contexts -- one per GPU -- behind one caller and ONE host thread, with the path's one
// exchange inside the library.  What the reference does with `chunks = omp_get_max_threads()` inside process()
// (r1cs_gg_ppzksnark/prover.hpp:94-99) this does with GPUs: a member's point range is a chunk, the partial sums meet on member 0.
//
// Transports of the exchange (all-gather of <= 864 bytes per member): RCCL single-process communicators (ncclCommInitAll +
// a grouped ncclAllGather on the members' own streams; librccl.so is dlopen'ed at the first exchange that wants it, libzkhip.so does not
// link it), stream-ordered peer copies, or a page-locked host buffer.  No kernel here: host orchestration of the kernels of msm.hip / ntt.hip.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only: every RCCL function is reached through dlsym

#include <algorithm>
#include <memory>

#include "ctx.hpp"
#include "zk_defs.hpp"

using namespace zkhip;

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString; }
};

// process-wide: the library is loaded once and never unloaded (its communicators belong to the groups)
RcclApi *rccl_api(std::string &err) {
    static std::mutex m;
    static RcclApi api;
    static bool tried = false;
    static std::string load_error;
    std::lock_guard<std::mutex> g(m);
    if (!tried) {
        tried = true;
        // The RCCL that belongs to the HIP runtime THIS library is bound to: a process may hold two ROCm installations (PyTorch wheels ship their
        // own libamdhip64 / librccl next to /opt/rocm's; whichever libamdhip64 was loaded first serves everybody), and an RCCL built against the
        // other one fails inside ncclCommInitAll ("unhandled cuda error": found by tests/fuzz_gpu.py, which loads libzkhip.so before torch).  So:
        // the directory of the libamdhip64 that hipGetDeviceCount resolves to first, then the loader's own search.
        std::vector<std::string> names;
        Dl_info where;
        if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &where) && where.dli_fname) {
            const std::string path(where.dli_fname);
            const size_t slash = path.rfind('/');
            if (slash != std::string::npos) {
                names.push_back(path.substr(0, slash) + "/librccl.so.1");
                names.push_back(path.substr(0, slash) + "/librccl.so");
            }
        }
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(n);
        for (const std::string &name : names) {
            api.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) {
            const char *e = dlerror();
            load_error = std::string("dlopen(librccl.so.1): ") + (e ? e : "not found");
        } else {
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(api.lib, "ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.lib, "ncclAllGather"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
            if (!api.ok()) load_error = "librccl.so.1 lacks one of ncclCommInitAll / ncclCommDestroy / ncclAllGather / ncclGroupStart / ncclGroupEnd";
        }
    }
    if (!api.lib || !api.ok()) {
        err = load_error;
        return nullptr;
    }
    return &api;
}

}  // namespace

struct zkhip_device_group {
    std::vector<zkhip_ctx *> members;
    std::vector<int> devices;
    bool distinct = true;  // pairwise distinct devices (what RCCL needs)
    int transport_req = ZKHIP_GROUP_AUTO, transport = ZKHIP_GROUP_AUTO;
    std::string last_error;
    // RCCL
    RcclApi *rccl = nullptr;
    std::vector<ncclComm_t> comms;
    std::vector<void *> rccl_recv;  // per member: where an all-gather lands when the caller gave no receive buffer (RCCL needs one on every rank)
    size_t rccl_recv_cap = 0;
    // PEER: one event per member marks "this member's stream up to here"
    std::vector<hipEvent_t> ev;
    // STAGED
    void *h_stage = nullptr;
    size_t h_stage_cap = 0;
    // zkhip_group_msm: partial sums (member k: 512 B at d_part[k]), the gathered sums and the fold on member 0
    std::vector<void *> d_part;
    void *d_all = nullptr;
};

struct zkhip_group_bases {
    int curve = 0, group = 0;
    size_t n = 0;
    std::vector<zkhip_bases *> member;  // member k's slice (nullptr: empty)
    std::vector<size_t> first;          // index of its first point; first[size] = n
};

#define ZK_GROUP_HIP(g, expr)                                                     \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            (g)->last_error = std::string(#expr) + ": " + hipGetErrorString(e__); \
            return e__ == hipErrorOutOfMemory ? ZKHIP_ERR_OOM : ZKHIP_ERR_HIP;    \
        }                                                                         \
    } while (0)

static int group_fail(zkhip_device_group *g, int rc, const zkhip_ctx *ctx, const char *what) {
    g->last_error = std::string(what) + ": " + zkhip_strerror(rc) + (ctx && !ctx->last_error.empty() ? " [" + ctx->last_error + "]" : std::string());
    return rc;
}
#define ZK_GROUP_TRY(g, ctx, expr)                              \
    do {                                                        \
        int rc__ = (expr);                                      \
        if (rc__ != 0) return group_fail(g, rc__, ctx, #expr);  \
    } while (0)

// balanced contiguous split of [0, n) over `world` (the first n % world parts hold one more): the same cut the shim's shard_range makes
static inline size_t part_lo(size_t n, size_t k, size_t world) { return k * (n / world) + std::min(k, n % world); }

static int rccl_ensure(zkhip_device_group *g) {
    if (!g->comms.empty()) return ZKHIP_OK;
    if (!g->distinct) {
        g->last_error = "RCCL transport: two members of the group share a device (RCCL refuses two ranks on one GPU)";
        return ZKHIP_ERR_INVALID;
    }
    std::string err;
    g->rccl = rccl_api(err);
    if (!g->rccl) {
        g->last_error = err;
        return ZKHIP_ERR_HIP;
    }
    g->comms.assign(g->members.size(), nullptr);
    ncclResult_t r = g->rccl->CommInitAll(g->comms.data(), (int)g->members.size(), g->devices.data());
    if (r != ncclSuccess) {
        g->last_error = std::string("ncclCommInitAll: ") + g->rccl->GetErrorString(r);
        g->comms.clear();
        return ZKHIP_ERR_HIP;
    }
    return ZKHIP_OK;
}

// what AUTO means for this group, decided at the first exchange
static int resolve_transport(zkhip_device_group *g) {
    if (g->transport != ZKHIP_GROUP_AUTO) return ZKHIP_OK;
    if (g->transport_req != ZKHIP_GROUP_AUTO) {
        if (g->transport_req == ZKHIP_GROUP_RCCL) ZK_TRY(rccl_ensure(g));
        g->transport = g->transport_req;
        return ZKHIP_OK;
    }
    if (g->members.size() > 1 && g->distinct && rccl_ensure(g) == ZKHIP_OK) g->transport = ZKHIP_GROUP_RCCL;
    else g->transport = ZKHIP_GROUP_PEER;
    return ZKHIP_OK;
}

static int stage_reserve(zkhip_device_group *g, size_t bytes) {
    if (bytes <= g->h_stage_cap) return ZKHIP_OK;
    if (g->h_stage) (void)hipHostFree(g->h_stage);
    g->h_stage = nullptr;
    g->h_stage_cap = 0;
    ZK_GROUP_HIP(g, hipHostMalloc(&g->h_stage, bytes, hipHostMallocPortable));
    g->h_stage_cap = bytes;
    return ZKHIP_OK;
}

extern "C" {

int zkhip_group_init(const int *device_ids, int n_dev, zkhip_device_group **out) {
    if (!out) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n_dev < 1 || n_dev > 64) return ZKHIP_ERR_INVALID;
    std::unique_ptr<zkhip_device_group> g(new zkhip_device_group());
    int rc = ZKHIP_OK;
    for (int k = 0; k < n_dev && rc == ZKHIP_OK; ++k) {
        zkhip_ctx *c = nullptr;
        rc = zkhip_init(device_ids[k], &c);
        if (rc != ZKHIP_OK) break;
        g->members.push_back(c);
        g->devices.push_back(device_ids[k]);
        for (int j = 0; j < k; ++j)
            if (device_ids[j] == device_ids[k]) g->distinct = false;
    }
    for (size_t k = 0; k < g->members.size() && rc == ZKHIP_OK; ++k) {
        hipEvent_t e = nullptr;
        void *p = nullptr;
        if (hipSetDevice(g->devices[k]) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = ZKHIP_ERR_HIP;
        else if (hipMalloc(&p, 512) != hipSuccess) rc = ZKHIP_ERR_OOM;
        if (e) g->ev.push_back(e);
        if (p) g->d_part.push_back(p);
    }
    if (rc == ZKHIP_OK) {
        if (hipSetDevice(g->devices[0]) != hipSuccess || hipMalloc(&g->d_all, 512 * (size_t)n_dev + 512) != hipSuccess) rc = ZKHIP_ERR_OOM;
    }
    if (rc == ZKHIP_OK && g->distinct && n_dev > 1) {
        // direct xGMI copies between the members' memories (hipMemcpyPeerAsync works without it, through the host); "already enabled" and
        // "not supported" are both fine here
        for (int a = 0; a < n_dev; ++a) {
            (void)hipSetDevice(g->devices[a]);
            for (int b = 0; b < n_dev; ++b)
                if (a != b) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, g->devices[a], g->devices[b]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(g->devices[b], 0);
                }
        }
        (void)hipGetLastError();
    }
    if (rc != ZKHIP_OK) {
        zkhip_group_destroy(g.release());
        return rc;
    }
    *out = g.release();
    return ZKHIP_OK;
}

void zkhip_group_destroy(zkhip_device_group *g) {
    if (!g) return;
    for (size_t k = 0; k < g->members.size(); ++k) {
        (void)hipSetDevice(g->devices[k]);
        (void)hipStreamSynchronize(g->members[k]->stream);
    }
    if (g->rccl)
        for (ncclComm_t c : g->comms)
            if (c) (void)g->rccl->CommDestroy(c);
    for (size_t k = 0; k < g->members.size(); ++k) {
        (void)hipSetDevice(g->devices[k]);
        if (k < g->ev.size()) (void)hipEventDestroy(g->ev[k]);
        if (k < g->d_part.size()) (void)hipFree(g->d_part[k]);
        if (k < g->rccl_recv.size() && g->rccl_recv[k]) (void)hipFree(g->rccl_recv[k]);
    }
    if (g->d_all) (void)hipFree(g->d_all);
    if (g->h_stage) (void)hipHostFree(g->h_stage);
    for (zkhip_ctx *c : g->members) zkhip_destroy(c);
    delete g;
}

int zkhip_group_size(const zkhip_device_group *g) { return g ? (int)g->members.size() : 0; }
zkhip_ctx *zkhip_group_ctx(const zkhip_device_group *g, int member) {
    return g && member >= 0 && (size_t)member < g->members.size() ? g->members[member] : nullptr;
}
const char *zkhip_group_last_error(const zkhip_device_group *g) { return g ? g->last_error.c_str() : ""; }

int zkhip_group_set_transport(zkhip_device_group *g, int transport) {
    if (!g || transport < ZKHIP_GROUP_AUTO || transport > ZKHIP_GROUP_STAGED) return ZKHIP_ERR_INVALID;
    if (transport == ZKHIP_GROUP_RCCL) ZK_TRY(rccl_ensure(g));
    g->transport_req = transport;
    g->transport = transport;  // AUTO: resolved again at the next exchange
    return ZKHIP_OK;
}
int zkhip_group_transport(const zkhip_device_group *g) { return g ? g->transport : ZKHIP_GROUP_AUTO; }

int zkhip_group_sync(zkhip_device_group *g) {
    if (!g) return ZKHIP_ERR_INVALID;
    for (zkhip_ctx *c : g->members) ZK_GROUP_TRY(g, c, zkhip_sync(c));
    return ZKHIP_OK;
}

int zkhip_group_all_gather(zkhip_device_group *g, const void *const *d_send, void *const *d_recv, size_t bytes) {
    if (!g || !d_send || !d_recv) return ZKHIP_ERR_INVALID;
    const size_t n = g->members.size();
    for (size_t k = 0; k < n; ++k)
        if (!d_send[k]) return ZKHIP_ERR_INVALID;
    if (bytes == 0) return ZKHIP_OK;
    ZK_TRY(resolve_transport(g));
    if (g->transport == ZKHIP_GROUP_RCCL) {
        // every rank of an RCCL all-gather receives: members without a buffer of the caller's land in one the group keeps
        bool need_own = false;
        for (size_t k = 0; k < n; ++k) need_own = need_own || !d_recv[k];
        if (need_own && g->rccl_recv_cap < n * bytes) {
            ZK_TRY(zkhip_group_sync(g));
            g->rccl_recv.resize(n, nullptr);
            for (size_t k = 0; k < n; ++k) {
                ZK_GROUP_HIP(g, hipSetDevice(g->devices[k]));
                if (g->rccl_recv[k]) (void)hipFree(g->rccl_recv[k]);
                g->rccl_recv[k] = nullptr;
                ZK_GROUP_HIP(g, hipMalloc(&g->rccl_recv[k], n * bytes));
            }
            g->rccl_recv_cap = n * bytes;
        }
        ncclResult_t r = g->rccl->GroupStart();
        for (size_t k = 0; k < n && r == ncclSuccess; ++k)
            r = g->rccl->AllGather(d_send[k], d_recv[k] ? d_recv[k] : g->rccl_recv[k], bytes, ncclUint8, g->comms[k], g->members[k]->stream);
        const ncclResult_t r2 = g->rccl->GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) {
            g->last_error = std::string("ncclAllGather: ") + g->rccl->GetErrorString(r != ncclSuccess ? r : r2);
            return ZKHIP_ERR_HIP;
        }
        return ZKHIP_OK;
    }
    if (g->transport == ZKHIP_GROUP_STAGED) {
        ZK_TRY(stage_reserve(g, n * bytes));
        for (size_t k = 0; k < n; ++k) {
            ZK_GROUP_HIP(g, hipSetDevice(g->devices[k]));
            ZK_GROUP_HIP(g, hipMemcpyAsync(static_cast<char *>(g->h_stage) + k * bytes, d_send[k], bytes, hipMemcpyDeviceToHost, g->members[k]->stream));
        }
        for (size_t k = 0; k < n; ++k) ZK_GROUP_HIP(g, hipStreamSynchronize(g->members[k]->stream));
        for (size_t k = 0; k < n; ++k)
            if (d_recv[k]) {
                ZK_GROUP_HIP(g, hipSetDevice(g->devices[k]));
                ZK_GROUP_HIP(g, hipMemcpyAsync(d_recv[k], g->h_stage, n * bytes, hipMemcpyHostToDevice, g->members[k]->stream));
            }
        // the staging buffer is the group's: the copies out of it have finished before the next exchange may fill it again
        for (size_t k = 0; k < n; ++k)
            if (d_recv[k]) ZK_GROUP_HIP(g, hipStreamSynchronize(g->members[k]->stream));
        return ZKHIP_OK;
    }
    // PEER: every source stream is marked, every receiving stream waits for the marks and pulls
    for (size_t k = 0; k < n; ++k) {
        ZK_GROUP_HIP(g, hipSetDevice(g->devices[k]));
        ZK_GROUP_HIP(g, hipEventRecord(g->ev[k], g->members[k]->stream));
    }
    for (size_t j = 0; j < n; ++j) {
        if (!d_recv[j]) continue;
        ZK_GROUP_HIP(g, hipSetDevice(g->devices[j]));
        hipStream_t s = g->members[j]->stream;
        for (size_t k = 0; k < n; ++k) {
            if (k != j) ZK_GROUP_HIP(g, hipStreamWaitEvent(s, g->ev[k], 0));
            char *dst = static_cast<char *>(d_recv[j]) + k * bytes;
            // the peer form also between members that share a GPU: the call a multi-GPU box makes is the call the one-GPU tests make
            ZK_GROUP_HIP(g, hipMemcpyPeerAsync(dst, g->devices[j], d_send[k], g->devices[k], bytes, s));
        }
    }
    return ZKHIP_OK;
}

int zkhip_group_copy(zkhip_device_group *g, int dst_member, void *d_dst, int src_member, const void *d_src, size_t bytes) {
    if (!g || dst_member < 0 || src_member < 0 || (size_t)dst_member >= g->members.size() || (size_t)src_member >= g->members.size()) return ZKHIP_ERR_INVALID;
    if (bytes == 0) return ZKHIP_OK;
    if (!d_dst || !d_src) return ZKHIP_ERR_INVALID;
    ZK_TRY(resolve_transport(g));
    zkhip_ctx *src = g->members[src_member], *dst = g->members[dst_member];
    if (g->transport == ZKHIP_GROUP_STAGED && src != dst) {  // through the host whatever the devices: the fallback is the same code on every box (and testable on one GPU)
        ZK_TRY(stage_reserve(g, bytes));
        ZK_GROUP_HIP(g, hipSetDevice(src->device));
        ZK_GROUP_HIP(g, hipMemcpyAsync(g->h_stage, d_src, bytes, hipMemcpyDeviceToHost, src->stream));
        ZK_GROUP_HIP(g, hipStreamSynchronize(src->stream));
        ZK_GROUP_HIP(g, hipSetDevice(dst->device));
        ZK_GROUP_HIP(g, hipMemcpyAsync(d_dst, g->h_stage, bytes, hipMemcpyHostToDevice, dst->stream));
        ZK_GROUP_HIP(g, hipStreamSynchronize(dst->stream));
        return ZKHIP_OK;
    }
    if (src != dst) {
        ZK_GROUP_HIP(g, hipSetDevice(src->device));
        ZK_GROUP_HIP(g, hipEventRecord(g->ev[src_member], src->stream));
    }
    ZK_GROUP_HIP(g, hipSetDevice(dst->device));
    if (src != dst) ZK_GROUP_HIP(g, hipStreamWaitEvent(dst->stream, g->ev[src_member], 0));
    ZK_GROUP_HIP(g, hipMemcpyPeerAsync(d_dst, dst->device, d_src, src->device, bytes, dst->stream));
    return ZKHIP_OK;
}

// ---- bases cut by point range ---------------------------------------------------------------------------------------------------------
static int group_bases_make(zkhip_device_group *g, int curve, int group, const uint64_t *xy, const uint8_t *inf, const uint64_t *base_xy, const uint64_t *scalars,
                            size_t n, zkhip_group_bases **out) {
    if (!g || !out) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    if ((curve != CURVE_BLS12_381 && curve != CURVE_BN254) || (group != GROUP_G1 && group != GROUP_G2)) return ZKHIP_ERR_INVALID;
    const size_t world = g->members.size(), words = 2 * zk_coord_limbs64(curve, group);
    std::unique_ptr<zkhip_group_bases> b(new zkhip_group_bases());
    b->curve = curve;
    b->group = group;
    b->n = n;
    b->member.assign(world, nullptr);
    for (size_t k = 0; k <= world; ++k) b->first.push_back(part_lo(n, k, world));
    for (size_t k = 0; k < world; ++k) {
        const size_t lo = b->first[k], cnt = b->first[k + 1] - lo;
        int rc;
        if (scalars) rc = zkhip_bases_from_scalars(g->members[k], curve, group, base_xy, scalars + 4 * lo, cnt, &b->member[k]);
        else rc = zkhip_bases_upload(g->members[k], curve, group, xy ? xy + words * lo : nullptr, inf ? inf + lo : nullptr, cnt, &b->member[k]);
        if (rc != ZKHIP_OK) {
            group_fail(g, rc, g->members[k], scalars ? "zkhip_bases_from_scalars" : "zkhip_bases_upload");
            zkhip_group_bases_free(g, b.release());
            return rc;
        }
    }
    *out = b.release();
    return ZKHIP_OK;
}

int zkhip_group_bases_upload(zkhip_device_group *g, int curve, int group, const uint64_t *affine_xy, const uint8_t *is_infinity, size_t n, zkhip_group_bases **out) {
    if (n && !affine_xy) return ZKHIP_ERR_INVALID;
    return group_bases_make(g, curve, group, affine_xy, is_infinity, nullptr, nullptr, n, out);
}
int zkhip_group_bases_from_scalars(zkhip_device_group *g, int curve, int group, const uint64_t *base_affine_xy, const uint64_t *scalars, size_t n,
                                   zkhip_group_bases **out) {
    if (!scalars) return ZKHIP_ERR_INVALID;
    return group_bases_make(g, curve, group, nullptr, nullptr, base_affine_xy, scalars, n, out);
}
void zkhip_group_bases_free(zkhip_device_group *g, zkhip_group_bases *b) {
    if (!b) return;
    for (size_t k = 0; k < b->member.size(); ++k)
        if (b->member[k]) zkhip_bases_free(g && k < g->members.size() ? g->members[k] : nullptr, b->member[k]);
    delete b;
}
size_t zkhip_group_bases_size(const zkhip_group_bases *b) { return b ? b->n : 0; }
const zkhip_bases *zkhip_group_bases_member(const zkhip_group_bases *b, int member, size_t *first) {
    if (!b || member < 0 || (size_t)member >= b->member.size()) return nullptr;
    if (first) *first = b->first[member];
    return b->member[member];
}

// ---- multiexp over the group --------------------------------------------------------------------------------------------------------
int zkhip_group_msm(zkhip_device_group *g, const zkhip_group_bases *bases, size_t offset, size_t n, const uint64_t *scalars, uint64_t *out_jacobian) {
    if (!g || !bases || !out_jacobian || (n && !scalars)) return ZKHIP_ERR_INVALID;
    const size_t world = g->members.size();
    if (bases->member.size() != world) return ZKHIP_ERR_INVALID;
    if (offset > bases->n || n > bases->n - offset) return ZKHIP_ERR_RANGE;
    const size_t obytes = 3 * zk_coord_limbs64(bases->curve, bases->group) * 8;  // <= 288
    std::vector<const void *> send(world);
    std::vector<void *> recv(world, nullptr);
    for (size_t k = 0; k < world; ++k) {
        zkhip_ctx *c = g->members[k];
        // this member's points [first[k], first[k + 1]) cut with the call's range [offset, offset + n)
        const size_t lo = std::max(bases->first[k], offset), hi = std::min(bases->first[k + 1], offset + n), cnt = hi > lo ? hi - lo : 0;
        ZK_GROUP_HIP(g, hipSetDevice(c->device));
        ZK_GROUP_TRY(g, c, zk_msm_host_reserve(c, cnt));
        uint32_t *d_s = c->msm_host_buf + 128;
        if (cnt) ZK_GROUP_HIP(g, hipMemcpyAsync(d_s, scalars + 4 * (lo - offset), cnt * 32, hipMemcpyHostToDevice, c->stream));
        // an empty slice yields the point at infinity (Z = 0) like any empty multiexp
        ZK_GROUP_TRY(g, c, zk_msm_run(c, bases->member[k], cnt ? lo - bases->first[k] : 0, cnt, d_s, static_cast<uint32_t *>(g->d_part[k])));
        send[k] = g->d_part[k];
    }
    zkhip_ctx *root = g->members[0];
    if (world == 1) {
        ZK_GROUP_HIP(g, hipMemcpyAsync(out_jacobian, g->d_part[0], obytes, hipMemcpyDeviceToHost, root->stream));
        ZK_GROUP_HIP(g, hipStreamSynchronize(root->stream));
        return ZKHIP_OK;
    }
    recv[0] = g->d_all;
    ZK_TRY(zkhip_group_all_gather(g, send.data(), recv.data(), obytes));
    ZK_GROUP_HIP(g, hipSetDevice(root->device));
    uint32_t *d_sum = reinterpret_cast<uint32_t *>(static_cast<char *>(g->d_all) + 512 * world);
    ZK_GROUP_TRY(g, root, zk_jac_sum(root, bases->curve, bases->group, static_cast<const uint32_t *>(g->d_all), world, d_sum));
    ZK_GROUP_HIP(g, hipMemcpyAsync(out_jacobian, d_sum, obytes, hipMemcpyDeviceToHost, root->stream));
    // every member's stream drains: its d_part may be overwritten by the next call, and a failed member must not go unnoticed
    for (size_t k = 0; k < world; ++k) ZK_GROUP_HIP(g, hipStreamSynchronize(g->members[k]->stream));
    return ZKHIP_OK;
}

// ---- NTT batch dealt over the group ---------------------------------------------------------------------------------------------------
int zkhip_group_ntt(zkhip_device_group *g, int curve, uint64_t *data, size_t log_m, size_t batch, const uint64_t *omega, int inverse, const uint64_t *coset_gen) {
    if (!g || !omega || (batch && !data)) return ZKHIP_ERR_INVALID;
    if (log_m > 32) return ZKHIP_ERR_RANGE;
    if (batch == 0) return ZKHIP_OK;
    const size_t world = g->members.size(), vec_bytes = ((size_t)1 << log_m) * 32;
    std::vector<void *> d(world, nullptr);
    int rc = ZKHIP_OK;
    for (size_t k = 0; k < world && rc == ZKHIP_OK; ++k) {
        const size_t lo = part_lo(batch, k, world), cnt = part_lo(batch, k + 1, world) - lo;
        if (cnt == 0) continue;
        zkhip_ctx *c = g->members[k];
        rc = zkhip_malloc(c, cnt * vec_bytes, &d[k]);
        if (rc != ZKHIP_OK) {
            group_fail(g, rc, c, "zkhip_malloc");
            break;
        }
        char *h = reinterpret_cast<char *>(data) + lo * vec_bytes;
        if (hipMemcpyAsync(d[k], h, cnt * vec_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            rc = group_fail(g, ZKHIP_ERR_HIP, c, "hipMemcpyAsync(H2D)");
            break;
        }
        rc = zk_ntt_run(c, curve, static_cast<uint32_t *>(d[k]), log_m, cnt, omega, inverse ? 1 : 0, coset_gen);
        if (rc != ZKHIP_OK) {
            group_fail(g, rc, c, "zk_ntt_run");
            break;
        }
        if (hipMemcpyAsync(h, d[k], cnt * vec_bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = group_fail(g, ZKHIP_ERR_HIP, c, "hipMemcpyAsync(D2H)");
    }
    for (size_t k = 0; k < world; ++k) {
        if (!d[k]) continue;
        zkhip_ctx *c = g->members[k];
        (void)hipSetDevice(c->device);
        if (hipStreamSynchronize(c->stream) != hipSuccess && rc == ZKHIP_OK) rc = group_fail(g, ZKHIP_ERR_HIP, c, "hipStreamSynchronize");
        (void)zkhip_free(c, d[k]);
    }
    return rc;
}

}  // extern "C"
