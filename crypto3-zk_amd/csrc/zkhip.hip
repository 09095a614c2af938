// extern "C" surface of libzkhip.so (declared in include/zkhip.h).  No torch types, no CPU fallback:
// if no HIP device is usable, zkhip_init fails and nothing else can be called.
#include <algorithm>
#include <cstdlib>

#include "ctx.hpp"
#include "msm_recode.hpp"
#include "zk_defs.hpp"

using namespace zkhip;

static int check_device(zkhip_ctx *ctx) {
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return 0;
}

extern "C" {

const char *zkhip_strerror(int status) {
    switch (status) {
        case ZKHIP_OK: return "ok";
        case ZKHIP_ERR_NO_DEVICE: return "no usable HIP device (zkhip has no CPU fallback)";
        case ZKHIP_ERR_INVALID: return "invalid argument";
        case ZKHIP_ERR_HIP: return "HIP runtime error (see zkhip_last_error)";
        case ZKHIP_ERR_OOM: return "out of device memory";
        case ZKHIP_ERR_RANGE: return "size or offset out of range";
        default: return "unknown status";
    }
}

// Process-wide view of the block caches: which context a live block of zkhip_malloc belongs to -- so that zkhip_free through
// ANOTHER context of the process settles the block with its owner instead of leaving a stale entry there --, and which contexts exist,
// so that an out-of-memory in one of them can empty its siblings' caches on the same device.  Lock order: g_alloc_mutex, then a
// context's alloc_mutex; never the other way round.
static std::mutex g_alloc_mutex;
static std::unordered_map<void *, zkhip_ctx *> g_block_owner;
static std::unordered_set<zkhip_ctx *> g_contexts;

int zkhip_init(int device_id, zkhip_ctx **out) {
    if (!out) return ZKHIP_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZKHIP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return ZKHIP_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return ZKHIP_ERR_NO_DEVICE;
    zkhip_ctx *ctx = new zkhip_ctx();
    ctx->device = device_id;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ZKHIP_ERR_HIP;
    }
    ctx->own_stream = true;
    if (hipMalloc((void **)&ctx->d_status, 4) != hipSuccess || hipMemset(ctx->d_status, 0, 4) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return ZKHIP_ERR_OOM;
    }
    *out = ctx;
    {
        std::lock_guard<std::mutex> g(g_alloc_mutex);
        g_contexts.insert(ctx);
    }
    // A/B runs of programs that create their own contexts (the bench library, the shim's default context): ZKHIP_OPTIONS="name=value,name=value"
    // is applied to every new context; an unknown name or a malformed entry is ignored (the variable is for measurements, not for deployments)
    if (const char *env = getenv("ZKHIP_OPTIONS")) {
        std::string all(env);
        size_t at = 0;
        while (at < all.size()) {
            const size_t end = all.find(',', at);
            const std::string item = all.substr(at, end == std::string::npos ? std::string::npos : end - at);
            const size_t eq = item.find('=');
            if (eq != std::string::npos && eq > 0) (void)zkhip_set_option(ctx, item.substr(0, eq).c_str(), (int64_t)atoll(item.c_str() + eq + 1));
            if (end == std::string::npos) break;
            at = end + 1;
        }
    }
    return ZKHIP_OK;
}

static void alloc_cache_flush(zkhip_ctx *ctx);

void zkhip_destroy(zkhip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    zk_ntt_free_tables(ctx);
    zk_dom_free_tables(ctx);
    for (auto &p : ctx->prof.pending) {
        (void)hipEventDestroy(p.second.a);
        (void)hipEventDestroy(p.second.b);
    }
    for (auto &e : ctx->prof.pool) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    zk_graphs_clear(ctx);
    if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
    {
        /* the cached blocks go back to the driver; blocks the caller still HOLDS (a shared_ptr handle of the shim that outlives its context)
           are left alone and lose their owner: a later zkhip_free through any context hands them to hipFree (they used to be
           freed here, under the handle) */
        std::lock_guard<std::mutex> g(g_alloc_mutex);
        std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
        alloc_cache_flush(ctx);
        for (auto &e : ctx->alloc_live) g_block_owner.erase(e.first);
        ctx->alloc_live.clear();
        g_contexts.erase(ctx);
    }
    if (ctx->msm_host_buf) (void)hipFree(ctx->msm_host_buf);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int zkhip_device(const zkhip_ctx *ctx) { return ctx ? ctx->device : -1; }

int zkhip_stream_wait(zkhip_ctx *waiter, zkhip_ctx *signal) {
    if (!waiter || !signal || waiter->device != signal->device) return ZKHIP_ERR_INVALID;
    if (waiter == signal) return ZKHIP_OK;
    ZK_HIP_CHECK(waiter, hipSetDevice(waiter->device));
    if (!signal->order_event) ZK_HIP_CHECK(waiter, hipEventCreateWithFlags(&signal->order_event, hipEventDisableTiming));
    ZK_HIP_CHECK(waiter, hipEventRecord(signal->order_event, signal->stream));
    ZK_HIP_CHECK(waiter, hipStreamWaitEvent(waiter->stream, signal->order_event, 0));
    return ZKHIP_OK;
}

const char *zkhip_last_error(const zkhip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int zkhip_set_stream(zkhip_ctx *ctx, void *hip_stream) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return ZKHIP_OK;
}

int zkhip_sync(zkhip_ctx *ctx) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_device_status(zkhip_ctx *ctx, uint32_t *flags) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    uint32_t f = 0;
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(&f, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (f) ZK_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_status, 0, 4, ctx->stream));
    if (flags) *flags = f;
    if (f) {
        ctx->last_error = std::string("device status:") + ((f & ZK_STATUS_GATHER_RANGE) ? " gather index out of range;" : "") +
                          ((f & ZK_STATUS_MSM_PLAN_OVERFLOW) ? " MSM large-bucket plan overflow;" : "");
        return ZKHIP_ERR_RANGE;
    }
    return ZKHIP_OK;
}

int zkhip_set_option(zkhip_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return ZKHIP_ERR_INVALID;
    std::string n(name);
    if (n == "msm_window_bits") ctx->opt_msm_window_bits = (int)value;
    else if (n == "msm_segment_log") ctx->opt_msm_segment_log = (int)value;
    else if (n == "msm_sets") ctx->opt_msm_sets = (int)value;
    else if (n == "msm_tail_quads") ctx->opt_msm_tail_quads = (int)value;
    else if (n == "msm_tail_fold") ctx->opt_msm_tail_fold = (int)value;
    else if (n == "msm_fold_run") ctx->opt_msm_fold_run = (int)value;
    else if (n == "msm_tail_fold_g2") ctx->opt_msm_tail_fold_g2 = (int)value;
    else if (n == "msm_share_sort") ctx->opt_msm_share_sort = (int)value;
    else if (n == "ec_ntt_table_lanes") ctx->opt_ec_ntt_table_lanes = value < 0 ? 0 : (uint32_t)std::min<int64_t>(value, 1 << 24);
    else if (n == "msm_sort_tile_log") ctx->opt_msm_sort_tile_log = (int)value;
    else if (n == "ntt_radix_log") ctx->opt_ntt_radix_log = (int)value;
    else if (n == "ntt_tile_log") ctx->opt_ntt_tile_log = (int)value;
    else if (n == "ntt_pair") ctx->opt_ntt_pair = (int)value;
    else if (n == "poly_coset_extend") ctx->opt_poly_coset_extend = value != 0;
    else if (n == "msm_precompute") ctx->opt_msm_precompute = (int)value;
    else if (n == "msm_precompute_min") ctx->opt_msm_precompute_min = (int)value;
    else if (n == "msm_shard_world") {
        if (value < 1 || value > 64) return ZKHIP_ERR_RANGE;
        ctx->opt_msm_shard_world = (int)value;
        if (ctx->opt_msm_shard_rank >= ctx->opt_msm_shard_world) ctx->opt_msm_shard_rank = 0;
    } else if (n == "msm_shard_rank") {
        if (value < 0 || value >= ctx->opt_msm_shard_world) return ZKHIP_ERR_RANGE;
        ctx->opt_msm_shard_rank = (int)value;
    }
    else if (n == "msm_graphs") ctx->opt_msm_graphs = (int)value;
    else if (n == "alloc_cache_mb") {
        if (value < 0) return ZKHIP_ERR_RANGE;
        std::lock_guard<std::mutex> g(g_alloc_mutex);
        std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
        ctx->opt_alloc_cache_bytes = (size_t)value << 20;
        if (ctx->alloc_cached_bytes > ctx->opt_alloc_cache_bytes) alloc_cache_flush(ctx);
    }
    else if (n == "stream_priority") {
        // the context's OWN stream is recreated with a scheduling priority: < 0 the highest the device offers, > 0 the lowest, 0 the
        // default.  Two contexts on one GPU (the Groth16 shim's main and G2 streams) can so decide whose workgroups go first.
        if (!ctx->own_stream) return ZKHIP_ERR_INVALID;
        ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        int least = 0, greatest = 0;
        ZK_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        hipStream_t s = nullptr;
        ZK_HIP_CHECK(ctx, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, value < 0 ? greatest : (value > 0 ? least : 0)));
        (void)hipStreamDestroy(ctx->stream);
        ctx->stream = s;
        ctx->opt_stream_priority = (int)value;
        zk_graphs_clear(ctx);
    }
    else return ZKHIP_ERR_INVALID;
    return ZKHIP_OK;
}

int zkhip_get_option(const zkhip_ctx *ctx, const char *name, int64_t *value) {
    if (!ctx || !name || !value) return ZKHIP_ERR_INVALID;
    std::string n(name);
    if (n == "msm_window_bits") *value = ctx->opt_msm_window_bits;
    else if (n == "alloc_cache_mb") *value = (int64_t)(ctx->opt_alloc_cache_bytes >> 20);
    else if (n == "msm_segment_log") *value = ctx->opt_msm_segment_log;
    else if (n == "msm_sets") *value = ctx->opt_msm_sets;
    else if (n == "msm_tail_quads") *value = ctx->opt_msm_tail_quads;
    else if (n == "msm_tail_fold") *value = ctx->opt_msm_tail_fold;
    else if (n == "msm_fold_run") *value = ctx->opt_msm_fold_run;
    else if (n == "msm_tail_fold_g2") *value = ctx->opt_msm_tail_fold_g2;
    else if (n == "msm_share_sort") *value = ctx->opt_msm_share_sort;
    else if (n == "ec_ntt_table_lanes") *value = ctx->opt_ec_ntt_table_lanes;
    else if (n == "msm_sort_tile_log") *value = ctx->opt_msm_sort_tile_log;
    else if (n == "ntt_radix_log") *value = ctx->opt_ntt_radix_log;
    else if (n == "ntt_tile_log") *value = ctx->opt_ntt_tile_log;
    else if (n == "ntt_pair") *value = ctx->opt_ntt_pair;
    else if (n == "poly_coset_extend") *value = ctx->opt_poly_coset_extend;
    else if (n == "msm_precompute") *value = ctx->opt_msm_precompute;
    else if (n == "msm_precompute_min") *value = ctx->opt_msm_precompute_min;
    else if (n == "msm_shard_world") *value = ctx->opt_msm_shard_world;
    else if (n == "msm_shard_rank") *value = ctx->opt_msm_shard_rank;
    else if (n == "msm_graphs") *value = ctx->opt_msm_graphs;
    else if (n == "stream_priority") *value = ctx->opt_stream_priority;
    else return ZKHIP_ERR_INVALID;
    return ZKHIP_OK;
}

// give every cached block back to the driver; the caller holds g_alloc_mutex and ctx->alloc_mutex
static void alloc_cache_flush(zkhip_ctx *ctx) {
    for (auto &e : ctx->alloc_free) {
        g_block_owner.erase(e.second);
        (void)hipFree(e.second);
    }
    ctx->alloc_free.clear();
    ctx->alloc_cached_bytes = 0;
}
int zkhip_malloc(zkhip_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    const size_t want = ((bytes ? bytes : 1) + 0xFFFF) & ~(size_t)0xFFFF;
    {
        std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
        auto it = ctx->alloc_free.lower_bound(want);
        if (it != ctx->alloc_free.end() && it->first <= want + (want >> 3)) {  // a cached block of this size class (blocks enter the cache after a device sync)
            *dptr = it->second;
            ctx->alloc_cached_bytes -= it->first;
            ctx->alloc_live[*dptr] = it->first;
            ctx->alloc_free.erase(it);
            return ZKHIP_OK;  // its owner entry was kept while it sat in the cache
        }
    }
    hipError_t e = hipMalloc(dptr, want);
    if (e != hipSuccess) {  // out of memory: the caches of EVERY context on this device go back to the driver (a shim process holds several), then once more
        (void)hipGetLastError();
        std::lock_guard<std::mutex> g(g_alloc_mutex);
        for (zkhip_ctx *c : g_contexts)
            if (c->device == ctx->device) {
                std::lock_guard<std::mutex> lock(c->alloc_mutex);
                alloc_cache_flush(c);
            }
        e = hipMalloc(dptr, want);
    }
    ZK_HIP_CHECK(ctx, e);
    std::lock_guard<std::mutex> g(g_alloc_mutex);
    std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
    ctx->alloc_live[*dptr] = want;
    g_block_owner[*dptr] = ctx;
    return ZKHIP_OK;
}
int zkhip_free(zkhip_ctx *ctx, void *dptr) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (!dptr) return ZKHIP_OK;
    // hipFree used to synchronise the whole device before the block could be handed out again; a block entering the cache gets the same
    // guarantee -- no stream of ANY context (a scheme's upload stream, the prover's G2 stream) still reads or writes it.  The device
    // that is drained is the one the BLOCK lives on (its owner's; ADVICE r5: a free through a context on another GPU used to drain that
    // GPU instead), ctx's own for a block no context owns.
    int block_device = ctx->device;
    {
        std::lock_guard<std::mutex> g(g_alloc_mutex);
        auto own = g_block_owner.find(dptr);
        if (own != g_block_owner.end()) block_device = own->second->device;
    }
    ZK_HIP_CHECK(ctx, hipSetDevice(block_device));
    ZK_HIP_CHECK(ctx, hipDeviceSynchronize());
    std::lock_guard<std::mutex> g(g_alloc_mutex);
    auto own = g_block_owner.find(dptr);
    if (own == g_block_owner.end()) {  // not a live block of any context's zkhip_malloc (or its context is gone): the driver's
        ZK_HIP_CHECK(ctx, hipFree(dptr));
        return ZKHIP_OK;
    }
    zkhip_ctx *owner = own->second;  // the block is settled with the context that allocated it, whichever context it is freed through
    std::lock_guard<std::mutex> lock(owner->alloc_mutex);
    auto it = owner->alloc_live.find(dptr);
    if (it == owner->alloc_live.end()) return ZKHIP_ERR_INVALID;  // a double free: the block already sits in the owner's cache
    const size_t sz = it->second;
    owner->alloc_live.erase(it);
    if (sz <= owner->opt_alloc_cache_bytes && owner->alloc_cached_bytes + sz <= owner->opt_alloc_cache_bytes) {
        owner->alloc_free.emplace(sz, dptr);
        owner->alloc_cached_bytes += sz;
        return ZKHIP_OK;
    }
    g_block_owner.erase(own);
    ZK_HIP_CHECK(ctx, hipFree(dptr));
    return ZKHIP_OK;
}
int zkhip_memcpy_h2d(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}
int zkhip_memcpy_h2d_async(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_host_alloc(zkhip_ctx *ctx, size_t bytes, void **hptr) {
    if (!ctx || !hptr) return ZKHIP_ERR_INVALID;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ZK_HIP_CHECK(ctx, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocPortable));  // page-locked for EVERY device: a group's members upload from one staging buffer
    return ZKHIP_OK;
}
int zkhip_host_free(zkhip_ctx *ctx, void *hptr) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP_CHECK(ctx, hipHostFree(hptr));
    return ZKHIP_OK;
}
int zkhip_memcpy_d2h(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_memcpy_d2h_async(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_memcpy_d2d_async(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_memcpy_2d_d2d_async(zkhip_ctx *ctx, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width, size_t rows) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    if (width == 0 || rows == 0) return ZKHIP_OK;
    if (!dst || !src || width > dst_pitch || width > src_pitch) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    if (dst_pitch >= ((size_t)1 << 31) || src_pitch >= ((size_t)1 << 31)) {  // beyond the 2D engine's pitch range (a 2^27-point domain at step 1): row by row
        for (size_t r = 0; r < rows; ++r)
            ZK_HIP_CHECK(ctx, hipMemcpyAsync(static_cast<char *>(dst) + r * dst_pitch, static_cast<const char *>(src) + r * src_pitch, width, hipMemcpyDeviceToDevice,
                                             ctx->stream));
        return ZKHIP_OK;
    }
    ZK_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, rows, hipMemcpyDeviceToDevice, ctx->stream));
    return ZKHIP_OK;
}

// ---- bases ------------------------------------------------------------------------------------------
static int bases_alloc(zkhip_ctx *ctx, int curve, int group, size_t n, zkhip_bases **out) {
    if ((curve != CURVE_BLS12_381 && curve != CURVE_BN254) || (group != GROUP_G1 && group != GROUP_G2)) return ZKHIP_ERR_INVALID;
    zkhip_bases *b = new zkhip_bases();
    b->curve = curve;
    b->group = group;
    b->n = n;
    b->stride_u32 = zk_point_words(curve, group);
    b->d = nullptr;
    b->ntab = 1;
    b->c_tab = 0;
    b->nslots = 1;
    const bool shard = ctx->opt_msm_shard_world > 1;
    if (ctx->opt_msm_precompute && n >= (size_t)ctx->opt_msm_precompute_min) {
        b->c_tab = ctx->opt_msm_window_bits > 0 ? std::max(2, std::min(ZK_MSM_MAX_C, ctx->opt_msm_window_bits)) : zk_msm_auto_window(n);
        b->ntab = msm_windows(zk_scalar_bits(curve), b->c_tab);
        if (shard) {
            b->win_rank = ctx->opt_msm_shard_rank;
            b->win_world = ctx->opt_msm_shard_world;
        }
        b->nslots = shard ? b->local_windows() + (b->win_rank != 0 ? 1 : 0) : b->ntab;
        if (b->nslots < 1) b->nslots = 1;
    } else if (shard) {
        // too few points for tables: the whole (tiny) MSM is rank 0's, the other ranks contribute the point at infinity
        b->win_rank = ctx->opt_msm_shard_rank;
        b->win_world = ctx->opt_msm_shard_world;
    }
    hipError_t e = hipMalloc((void **)&b->d, std::max<size_t>(1, n) * b->nslots * b->stride_u32 * 4);
    if (e != hipSuccess && b->tables() && !shard) {
        // the window tables do not fit (ntab x n points): keep the points alone; the MSM then folds the windows by the
        // Horner pass of msm_final instead of bucket-wise
        (void)hipGetLastError();
        b->ntab = 1;
        b->c_tab = 0;
        b->nslots = 1;
        e = hipMalloc((void **)&b->d, std::max<size_t>(1, n) * b->stride_u32 * 4);
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipMalloc(bases): ") + hipGetErrorString(e);
        delete b;
        return ZKHIP_ERR_OOM;
    }
    *out = b;
    return 0;
}

int zkhip_bases_upload(zkhip_ctx *ctx, int curve, int group, const uint64_t *affine_xy, const uint8_t *is_infinity, size_t n,
                       zkhip_bases **out) {
    if (!ctx || !out || (n && !affine_xy)) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    zkhip_bases *b = nullptr;
    ZK_TRY(bases_alloc(ctx, curve, group, n, &b));
    uint8_t *d_inf = nullptr;
    uint32_t *d_canon = nullptr;
    const size_t cbytes = n * 2 * zk_coord_limbs64(curve, group) * 8;
    int rc = 0;
    do {
        if (n == 0) break;
        if (hipMalloc((void **)&d_canon, cbytes) != hipSuccess) {
            rc = ZKHIP_ERR_OOM;
            break;
        }
        if (hipMemcpyAsync(d_canon, affine_xy, cbytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = ZKHIP_ERR_HIP;
            break;
        }
        if (is_infinity) {
            if (hipMalloc((void **)&d_inf, n) != hipSuccess) {
                rc = ZKHIP_ERR_OOM;
                break;
            }
            if (hipMemcpyAsync(d_inf, is_infinity, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
                rc = ZKHIP_ERR_HIP;
                break;
            }
        }
        rc = zk_bases_to_mont(ctx, b, d_canon, d_inf);
        if (rc) break;
        rc = zk_bases_precompute(ctx, b);
        if (rc) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    } while (0);
    if (d_inf) (void)hipFree(d_inf);
    if (d_canon) (void)hipFree(d_canon);
    if (rc) {
        (void)hipFree(b->d);
        delete b;
        return rc;
    }
    *out = b;
    return ZKHIP_OK;
}

int zkhip_bases_upload_compressed(zkhip_ctx *ctx, int curve, int group, const uint8_t *octets, size_t n, zkhip_bases **out) {
    if (!ctx || !out || (n && !octets)) return ZKHIP_ERR_INVALID;
    if (curve != CURVE_BLS12_381) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    zkhip_bases *b = nullptr;
    ZK_TRY(bases_alloc(ctx, curve, group, n, &b));
    const size_t bytes = n * (group == GROUP_G2 ? 96 : 48);
    uint8_t *d_oct = nullptr;
    uint32_t rejected = 0;
    int rc = 0;
    do {
        if (n == 0) break;
        if (hipMalloc((void **)&d_oct, bytes + 16) != hipSuccess) {
            rc = ZKHIP_ERR_OOM;
            break;
        }
        uint32_t *d_err = reinterpret_cast<uint32_t *>(d_oct + ((bytes + 3) & ~(size_t)3));
        if (hipMemcpyAsync(d_oct, octets, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemsetAsync(d_err, 0, 4, ctx->stream) != hipSuccess) {
            rc = ZKHIP_ERR_HIP;
            break;
        }
        rc = zk_bases_decompress(ctx, b, d_oct, d_err);
        if (rc) break;
        if (hipMemcpyAsync(&rejected, d_err, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = ZKHIP_ERR_HIP;
            break;
        }
        if (rejected) {
            ctx->last_error = std::to_string(rejected) + " point encoding(s) rejected (flags, x >= p, or x not on the curve)";
            rc = ZKHIP_ERR_INVALID;
            break;
        }
        rc = zk_bases_precompute(ctx, b);
        if (rc) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    } while (0);
    if (d_oct) (void)hipFree(d_oct);
    if (rc) {
        (void)hipFree(b->d);
        delete b;
        return rc;
    }
    *out = b;
    return ZKHIP_OK;
}

// standard generators, canonical u32 limbs (x | y)
static const uint64_t GEN_BLS_G1[12] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL,
                                        0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL, 0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL,
                                        0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
static const uint64_t GEN_BLS_G2[24] = {
    0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL,
    0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL,
    0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL,
    0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};
static const uint64_t GEN_BN_G1[8] = {1, 0, 0, 0, 2, 0, 0, 0};
static const uint64_t GEN_BN_G2[16] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL,
                                       0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL,
                                       0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL,
                                       0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};

int zkhip_bases_from_scalars(zkhip_ctx *ctx, int curve, int group, const uint64_t *base_affine_xy, const uint64_t *scalars, size_t n,
                             zkhip_bases **out) {
    if (!ctx || !out || (n && !scalars)) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    zkhip_bases *b = nullptr;
    ZK_TRY(bases_alloc(ctx, curve, group, n, &b));
    const uint64_t *gen = base_affine_xy;
    if (!gen) gen = curve == CURVE_BLS12_381 ? (group == GROUP_G1 ? GEN_BLS_G1 : GEN_BLS_G2) : (group == GROUP_G1 ? GEN_BN_G1 : GEN_BN_G2);
    uint32_t *d_s = nullptr, *d_g = nullptr;
    int rc = 0;
    do {
        if (n == 0) break;
        if (hipMalloc((void **)&d_s, n * 32) != hipSuccess || hipMalloc((void **)&d_g, 2 * zk_coord_limbs64(curve, group) * 8) != hipSuccess) {
            rc = ZKHIP_ERR_OOM;
            break;
        }
        if (hipMemcpyAsync(d_s, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(d_g, gen, 2 * zk_coord_limbs64(curve, group) * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = ZKHIP_ERR_HIP;
            break;
        }
        rc = zk_bases_mul(ctx, b, d_g, d_s);
        if (rc) break;
        rc = zk_bases_precompute(ctx, b);
        if (rc) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    } while (0);
    if (d_s) (void)hipFree(d_s);
    if (d_g) (void)hipFree(d_g);
    if (rc) {
        (void)hipFree(b->d);
        delete b;
        return rc;
    }
    *out = b;
    return ZKHIP_OK;
}

// out row d_rows[j] (or first + j) <- src row j; a row is `stride` words
__global__ void bases_spread_rows(const uint32_t *__restrict__ src, size_t count, uint32_t stride, const uint32_t *__restrict__ d_rows, uint32_t first,
                                  uint32_t n_total, uint32_t *__restrict__ out, uint32_t *__restrict__ status) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count * stride) return;
    const size_t j = g / stride;
    const uint32_t w = (uint32_t)(g % stride), row = d_rows ? d_rows[j] : first + (uint32_t)j;
    // rows must be in range AND strictly increasing (the sparse B query's index list, r1cs_gg_ppzksnark.hpp check_b_indices): a
    // repeated row would silently overwrite a point, so it raises the same sticky flag as a row beyond the end
    if (row >= n_total || (d_rows && j + 1 < count && d_rows[j + 1] <= row)) {
        if (w == 0) atomicOr(status, ZK_STATUS_GATHER_RANGE);
        if (row >= n_total) return;
    }
    out[(size_t)row * stride + w] = src[g];
}

int zkhip_bases_spread(zkhip_ctx *ctx, const zkhip_bases *src, const uint32_t *d_rows, size_t first, size_t n_total, zkhip_bases **out) {
    if (!ctx || !src || !out) return ZKHIP_ERR_INVALID;
    if (n_total >= ((size_t)1 << 32) || (!d_rows && first + src->n > n_total)) return ZKHIP_ERR_RANGE;
    if (src->n * src->stride_u32 >= ((size_t)1 << 39)) return ZKHIP_ERR_RANGE;  // one lane per word: the grid must fit 32 bits
    ZK_TRY(check_device(ctx));
    zkhip_bases *b = nullptr;
    ZK_TRY(bases_alloc(ctx, src->curve, src->group, n_total, &b));
    int rc = 0;
    do {
        if (n_total == 0) break;
        const size_t words = n_total * b->stride_u32;
        if (hipMemsetAsync(b->d, 0, words * 4, ctx->stream) != hipSuccess) {  // (0, 0): the point at infinity
            rc = ZKHIP_ERR_HIP;
            break;
        }
        if (src->n) {
            const size_t lanes = src->n * src->stride_u32;
            ctx->prof_begin("bases_spread_rows");
            hipLaunchKernelGGL(bases_spread_rows, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, src->d, src->n, (uint32_t)src->stride_u32,
                               d_rows, (uint32_t)first, (uint32_t)n_total, b->d, ctx->d_status);
            ctx->prof_end();
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                ctx->last_error = std::string("bases_spread_rows: ") + hipGetErrorString(le);
                rc = ZKHIP_ERR_HIP;
                break;
            }
        }
        rc = zk_bases_precompute(ctx, b);
        if (rc) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    } while (0);
    if (rc) {
        (void)hipFree(b->d);
        delete b;
        return rc;
    }
    *out = b;
    return ZKHIP_OK;
}

int zkhip_bases_download(zkhip_ctx *ctx, const zkhip_bases *b, size_t offset, size_t n, uint64_t *affine_xy, uint8_t *is_infinity) {
    if (!ctx || !b || (n && (!affine_xy || !is_infinity))) return ZKHIP_ERR_INVALID;
    if (offset + n > b->n) return ZKHIP_ERR_RANGE;
    if (n == 0) return ZKHIP_OK;
    ZK_TRY(check_device(ctx));
    size_t pbytes = n * 2 * zk_coord_limbs64(b->curve, b->group) * 8;
    ZK_TRY(ctx->ws_reserve(zkhip_ctx::ws_round(pbytes) + zkhip_ctx::ws_round(n)));
    ctx->ws_reset();
    uint32_t *d_out = ctx->ws_take<uint32_t>(pbytes / 4);
    uint8_t *d_inf = ctx->ws_take<uint8_t>(n);
    ZK_TRY(zk_bases_from_mont(ctx, b, offset, n, d_out, d_inf));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(affine_xy, d_out, pbytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(is_infinity, d_inf, n, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

size_t zkhip_bases_size(const zkhip_bases *b) { return b ? b->n : 0; }

void zkhip_bases_free(zkhip_ctx *ctx, zkhip_bases *b) {
    if (!b) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(b->d);
    delete b;
}

// ---- MSM ----------------------------------------------------------------------------------------------
int zkhip_msm_dev(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const void *d_scalars, void *d_out_jacobian) {
    if (!ctx || !bases || !d_out_jacobian || (n && !d_scalars)) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    return zk_msm_run(ctx, bases, offset, n, (const uint32_t *)d_scalars, (uint32_t *)d_out_jacobian);
}

int zkhip_msm_batch_dev(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                        const void *const *d_scalars, void *const *d_out_jacobian) {
    if (!ctx || (count && (!bases || !offsets || !ns || !d_scalars || !d_out_jacobian))) return ZKHIP_ERR_INVALID;
    for (size_t i = 0; i < count; ++i)
        if (!bases[i] || !d_out_jacobian[i] || (ns[i] && !d_scalars[i])) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    return zk_msm_run_batch(ctx, count, bases, offsets, ns, (const uint32_t *const *)d_scalars, (uint32_t *const *)d_out_jacobian);
}

}  // extern "C"

// The device buffer zkhip_msm / zkhip_group_msm stage host scalars in: result in the first 512 bytes, scalars behind it.  The context keeps it
// (grow-only, freed with the context): no hipMalloc / hipFree per call (the free is a device synchronisation), and the SAME device addresses
// call after call, so the launch sequence replays as a HIP graph like the resident path's.
int zk_msm_host_reserve(zkhip_ctx *ctx, size_t n) {
    const size_t want = std::max<size_t>(1, n) * 32 + 512;
    if (ctx->msm_host_cap >= want) return 0;
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->msm_host_buf) (void)hipFree(ctx->msm_host_buf);
    ctx->msm_host_buf = nullptr;
    ctx->msm_host_cap = 0;
    zk_graphs_clear(ctx);  // graphs over the old buffer
    if (hipMalloc((void **)&ctx->msm_host_buf, want) != hipSuccess) {
        (void)hipGetLastError();
        return ZKHIP_ERR_OOM;
    }
    ctx->msm_host_cap = want;
    return 0;
}

extern "C" {

int zkhip_msm(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint64_t *scalars, uint64_t *out_jacobian) {
    if (!ctx || !bases || !out_jacobian || (n && !scalars)) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    const size_t obytes = 3 * zk_coord_limbs64(bases->curve, bases->group) * 8;
    ZK_TRY(zk_msm_host_reserve(ctx, n));
    uint32_t *d_o = ctx->msm_host_buf, *d_s = ctx->msm_host_buf + 128;
    if (n) ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_s, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_TRY(zk_msm_run(ctx, bases, offset, n, d_s, d_o));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(out_jacobian, d_o, obytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

int zkhip_jacobian_sum_dev(zkhip_ctx *ctx, int curve, int group, const void *d_points, size_t count, void *d_out_jacobian) {
    if (!ctx || !d_points || !d_out_jacobian) return ZKHIP_ERR_INVALID;
    if ((curve != CURVE_BLS12_381 && curve != CURVE_BN254) || (group != GROUP_G1 && group != GROUP_G2)) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    return zk_jac_sum(ctx, curve, group, (const uint32_t *)d_points, count, (uint32_t *)d_out_jacobian);
}

int zkhip_jacobian_to_affine(zkhip_ctx *ctx, int curve, int group, const uint64_t *jacobian, uint64_t *affine_xy, uint8_t *is_infinity) {
    if (!ctx || !jacobian || !affine_xy || !is_infinity) return ZKHIP_ERR_INVALID;
    if ((curve != CURVE_BLS12_381 && curve != CURVE_BN254) || (group != GROUP_G1 && group != GROUP_G2)) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    size_t cl = zk_coord_limbs64(curve, group) * 8;
    ZK_TRY(ctx->ws_reserve(4096));
    ctx->ws_reset();
    uint32_t *d_j = ctx->ws_take<uint32_t>(3 * cl / 4);
    uint32_t *d_a = ctx->ws_take<uint32_t>(2 * cl / 4);
    uint8_t *d_i = ctx->ws_take<uint8_t>(16);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_j, jacobian, 3 * cl, hipMemcpyHostToDevice, ctx->stream));
    ZK_TRY(zk_jac_to_affine(ctx, curve, group, d_j, d_a, d_i));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(affine_xy, d_a, 2 * cl, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(is_infinity, d_i, 1, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

// ---- NTT ----------------------------------------------------------------------------------------------
int zkhip_ntt_dev(zkhip_ctx *ctx, int curve, void *d_data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
                  const uint64_t *coset_gen) {
    if (!ctx || !omega || (batch && !d_data)) return ZKHIP_ERR_INVALID;
    ZK_TRY(check_device(ctx));
    return zk_ntt_run(ctx, curve, (uint32_t *)d_data, log_m, batch, omega, inverse ? 1 : 0, coset_gen);
}

int zkhip_ntt(zkhip_ctx *ctx, int curve, uint64_t *data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
              const uint64_t *coset_gen) {
    if (!ctx || !omega || (batch && !data)) return ZKHIP_ERR_INVALID;
    if (log_m > 32) return ZKHIP_ERR_RANGE;
    if (batch == 0) return ZKHIP_OK;
    ZK_TRY(check_device(ctx));
    size_t bytes = (batch << log_m) * 32;
    uint32_t *d = nullptr;
    ZK_HIP_CHECK(ctx, hipMalloc((void **)&d, bytes));
    int rc = 0;
    do {
        if (hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = ZKHIP_ERR_HIP;
            break;
        }
        rc = zk_ntt_run(ctx, curve, d, log_m, batch, omega, inverse ? 1 : 0, coset_gen);
        if (rc) break;
        if (hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            ctx->last_error = hipGetErrorString(hipGetLastError());
            rc = ZKHIP_ERR_HIP;
        }
    } while (0);
    (void)hipFree(d);
    return rc;
}

// ---- profiling ----------------------------------------------------------------------------------------
int zkhip_profile_enable(zkhip_ctx *ctx, int on) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ctx->prof_collect();
    ctx->prof.on = on != 0;
    return ZKHIP_OK;
}
int zkhip_profile_filter(zkhip_ctx *ctx, const char *prefix) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ctx->prof_collect();
    ctx->prof.filter = prefix ? prefix : "";
    return ZKHIP_OK;
}
int zkhip_profile_reset(zkhip_ctx *ctx) {
    if (!ctx) return ZKHIP_ERR_INVALID;
    ctx->prof_collect();
    ctx->prof.acc.clear();
    return ZKHIP_OK;
}
int zkhip_profile_get(zkhip_ctx *ctx, const char *prefix, double *total_ms, uint64_t *launches) {
    if (!ctx || !prefix) return ZKHIP_ERR_INVALID;
    ctx->prof_collect();
    double ms = 0;
    uint64_t cnt = 0;
    size_t pl = strlen(prefix);
    for (auto &kv : ctx->prof.acc) {
        if (kv.first.compare(0, pl, prefix) == 0) {
            ms += kv.second.first;
            cnt += kv.second.second;
        }
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    return ZKHIP_OK;
}
size_t zkhip_profile_dump(zkhip_ctx *ctx, char *buf, size_t cap) {
    if (!ctx) return 0;
    ctx->prof_collect();
    std::string s;
    char line[256];
    for (auto &kv : ctx->prof.acc) {
        snprintf(line, sizeof(line), "%s %.6f %llu\n", kv.first.c_str(), kv.second.first, (unsigned long long)kv.second.second);
        s += line;
    }
    if (buf && cap) {
        size_t k = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return s.size() + 1;
}

}  // extern "C"
