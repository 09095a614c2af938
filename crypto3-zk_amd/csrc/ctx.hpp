// Host-side context shared by the kernel families: stream, workspace arena, per-kernel HIP-event profiler.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/zkhip.h"

#define ZK_HIP_CHECK(ctx, expr)                                                                  \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e__);              \
            return e__ == hipErrorOutOfMemory ? ZKHIP_ERR_OOM : ZKHIP_ERR_HIP;                   \
        }                                                                                        \
    } while (0)

#define ZK_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != 0) return rc__; \
    } while (0)

static constexpr int ZK_MSM_MAX_C = 21;  // largest Pippenger window (bits); 2^(c-1) buckets per set (the sort handles <= 2^20 x sets / 1024 super-buckets)

// bits of the sticky device status word (kernels atomicOr them in; zkhip_device_status reports and clears)
enum : uint32_t { ZK_STATUS_GATHER_RANGE = 1u, ZK_STATUS_MSM_PLAN_OVERFLOW = 2u, ZK_STATUS_LOOKUP_NOT_IN_TABLE = 4u, ZK_STATUS_LOOKUP_SORT_OVERFLOW = 8u };

struct zkhip_bases {
    int curve, group;
    size_t n;
    size_t stride_u32;  // u32 words per affine point (2 * coordinate limbs)
    uint32_t *d;        // nslots tables of n Montgomery-form affine points each, (0,0) = infinity;
                        // the table of window w holds 2^off(w) P_i ("window tables": no Horner pass over the windows)
    int c_tab = 0, ntab = 1;  // window size the tables were built for (0: no tables) and the number of windows W
    // Window partition over GPUs: this object holds the tables of windows {w : w mod win_world == win_rank}
    // only, for ALL n points; an MSM over it yields the partial sum of those windows.  Slot 0 always holds the points
    // themselves (window 0, which belongs to rank 0; other ranks keep it as the source of their doubling chains).
    int win_rank = 0, win_world = 1;
    int nslots = 1;
    bool tables() const { return c_tab > 0; }
    int local_windows() const { return win_rank < ntab ? (ntab - win_rank + win_world - 1) / win_world : 0; }
    int slot_of_local(int lw) const { return lw + (win_rank != 0 ? 1 : 0); }  // local window lw = window win_rank + lw * win_world
};

struct ZkEventPair {
    hipEvent_t a, b;
};

struct ZkProfile {
    bool on = false;
    bool last_recorded = false;  // prof_begin recorded an event pair for the launch in flight
    std::string filter;          // non-empty: only kernels whose name starts with it are timed (fewer events in a timed region)
    std::vector<std::pair<std::string, ZkEventPair>> pending;
    std::vector<ZkEventPair> pool;
    std::map<std::string, std::pair<double, uint64_t>> acc;
};

struct NttTables;  // ntt.hip
struct DomTables;  // domain.hip

// a captured launch sequence (HIP graph) of one MSM / MSM batch, replayed when the same call comes again
struct ZkGraph {
    hipGraphExec_t exec = nullptr;
    uint64_t ws_epoch = 0;   // the workspace allocation the graph's addresses refer to
    void *d_ptrs = nullptr;  // batch: device array of the output pointers (kept with the graph)
};

struct zkhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t order_event = nullptr;  // zkhip_stream_wait: marks this context's stream for another context to wait on
    std::string last_error;
    // zkhip_malloc / zkhip_free keep freed blocks for reuse (sizes rounded to 64 KiB; a request takes a cached block up to 1/8 larger):
    // the polynomial-layer chains allocate the same GB-class temporaries again and again, and a raw hipMalloc / hipFree pair costs
    // milliseconds -- at times hundreds (measured: a 20-ms quotient chain with 170-230 ms outliers every third run).  "alloc_cache_mb"
    // caps what is kept (default 16 GiB of the 288; 0 turns the cache off); everything goes back at zkhip_destroy or when hipMalloc fails.
    std::multimap<size_t, void *> alloc_free;
    std::unordered_map<void *, size_t> alloc_live;
    size_t alloc_cached_bytes = 0, opt_alloc_cache_bytes = (size_t)16 << 30;
    std::mutex alloc_mutex;
    std::vector<uint32_t> lagrange_stage;  // host constants of zkhip_domain_lagrange_dev, alive until its copies ran
    uint32_t *d_status = nullptr;  // sticky device-side error flags (ZK_STATUS_*), read and cleared by zkhip_device_status
    // bump-allocated workspace, grown on demand, reused across calls
    char *ws = nullptr;
    size_t ws_cap = 0, ws_off = 0, ws_floor = 0;  // ws_floor: start of the per-call region (a batch parks data below it)
    std::unordered_set<const void *> lds_configured;  // kernels whose dynamic-LDS limit was raised on this context's device
    std::vector<uint64_t> lincomb_stage, lincomb_coeffs;  // host staging of zkhip_poly_lincomb_dev's tables
    std::vector<uint32_t> gate_stage;                     // host image of zkhip_gate_eval_dev's program
    std::vector<uint32_t *> batch_ptrs;            // host copy of a batch's output pointers (alive until the copy ran)
    // HIP graphs of repeated MSM calls (msm.hip: zk_graph_run)
    std::unordered_map<std::string, ZkGraph> graphs;
    std::unordered_map<std::string, int> graph_seen;
    uint64_t ws_epoch = 0;
    bool capturing = false;
    void *batch_dptrs_override = nullptr;  // during a batch capture: the graph-owned output-pointer array
    int opt_msm_graphs = 0;  // off: replaying the captured launch sequence measured no faster than issuing it
    uint32_t *msm_host_buf = nullptr;  // zkhip_msm (scalars in host memory): result + scalars on the device, kept across calls
    size_t msm_host_cap = 0;
    // pinned staging for small results
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    // options
    int opt_msm_window_bits = 0;
    int opt_msm_sets = 0;          // bucket sets S with window tables (entry (i, w) -> set w mod S); 0: from the lane target
    int opt_msm_sort_tile_log = 14;  // 14: 2^14-entry sort tiles (the MSM owns the GPU); 12: 2^12 (kernels of another context run alongside)
    uint32_t opt_ec_ntt_table_lanes = 0;  // EC-NTT: lanes per multiplication launch = window tables held at once (0: as many as fit 1 GiB)
    int opt_msm_share_sort = 1;    // batches: consecutive members over the same scalars and table geometry share one sort (msm_same_entries)
    int opt_msm_tail_quads = 1;    // group law over lane quads in the tail of small bucket sets (fu_quad.hpp); 0: pairs everywhere
    int opt_msm_tail_fold = 16;    // two-level tail (msm_core.hpp: row / column sums of the bucket index, then the old tail over 2 sets of ~sqrt(B) buckets) for table-backed sets of >= 2^k buckets; 0: off
    int opt_msm_tail_fold_g2 = 1;  // the two-level tail for G2 sets too (same threshold): a lone G2 MSM measures the same either way, a proof whose G2 MSM runs under its G1 MSMs gains the issue slots the shorter tail frees (Groth16 +2.6 %)
    int opt_msm_fold_run = 0;      // two-level tail: buckets a lane sums before the workgroup's tree (a power of two; 0: auto)
    int opt_msm_segment_log = -1;  // tail segments of 2^k buckets per lane; < 0: chosen from the lane count
    int opt_ntt_radix_log = 8;
    int opt_ntt_tile_log = 3;
    int opt_poly_coset_extend = 1;  // zkhip_poly_resize_dev n -> K n (K <= 16): the K - 1 new cosets by n-point transforms, the n known values copied (0: inverse + K n-point transform)
    int opt_ntt_pair = 1;  // log2 of the polynomials of a batch one NTT workgroup carries (1: pairs share indices, twiddles, factor-table reads)
    int opt_msm_precompute = 1;       // build window tables at upload for bases of >= opt_msm_precompute_min points
    int opt_msm_shard_rank = 0, opt_msm_shard_world = 1;  // window partition applied to bases uploaded from now on
    int opt_stream_priority = 0;      // < 0: the own stream was recreated with the highest priority, > 0: the lowest
    int opt_msm_precompute_min = 32;  // without tables the windows are combined by a serial Horner pass (~255 doublings on one lane: 3.8 ms)
    ZkProfile prof;
    std::vector<NttTables *> ntt_tables;
    std::vector<DomTables *> dom_tables;  // step / extended radix-2 domains (domain.hip)
    char *dom_ws = nullptr;               // scratch of zkhip_domain_fft_dev (the NTTs inside it use `ws`)
    size_t dom_ws_cap = 0;

    int ws_reserve(size_t bytes) {
        if (bytes <= ws_cap) return 0;
        if (capturing) {  // growing means a synchronisation and new addresses: not inside a stream capture
            last_error = "workspace growth during graph capture";
            return ZKHIP_ERR_HIP;
        }
        ++ws_epoch;
        if (ws) {
            hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                last_error = hipGetErrorString(e);
                return ZKHIP_ERR_HIP;
            }
            (void)hipFree(ws);
            ws = nullptr;
            ws_cap = 0;
        }
        size_t cap = bytes + (bytes >> 3) + (1 << 20);
        hipError_t e = hipMalloc((void **)&ws, cap);
        if (e != hipSuccess) {
            last_error = std::string("hipMalloc(workspace): ") + hipGetErrorString(e);
            return ZKHIP_ERR_OOM;
        }
        ws_cap = cap;
        return 0;
    }
    void ws_reset() { ws_off = ws_floor; }
    template <class T>
    T *ws_take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(ws + ws_off);
        ws_off += bytes;
        return p;
    }
    static size_t ws_round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

    // ---- profiler
    void prof_begin(const char *name) {
        prof.last_recorded = false;
        if (!prof.on) return;
        if (!prof.filter.empty() && strncmp(name, prof.filter.c_str(), prof.filter.size()) != 0) return;
        prof.last_recorded = true;
        ZkEventPair ev;
        if (!prof.pool.empty()) {
            ev = prof.pool.back();
            prof.pool.pop_back();
        } else {
            (void)hipEventCreate(&ev.a);
            (void)hipEventCreate(&ev.b);
        }
        (void)hipEventRecord(ev.a, stream);
        prof.pending.emplace_back(name, ev);
    }
    void prof_end() {
        if (!prof.on || !prof.last_recorded) return;
        (void)hipEventRecord(prof.pending.back().second.b, stream);
    }
    void prof_collect() {
        if (prof.pending.empty()) return;
        (void)hipStreamSynchronize(stream);
        for (auto &p : prof.pending) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, p.second.a, p.second.b);
            auto &a = prof.acc[p.first];
            a.first += ms;
            a.second += 1;
            prof.pool.push_back(p.second);
        }
        prof.pending.clear();
    }
};

// raise a kernel's dynamic-LDS limit once per context (the attribute is per device: a process may hold contexts on
// several GPUs, so a process-wide flag would skip it on the second device)
#define ZK_MAX_LDS(ctx, kernel, bytes)                                                                                              \
    do {                                                                                                                            \
        const void *fn__ = reinterpret_cast<const void *>(&kernel);                                                                 \
        if ((ctx)->lds_configured.insert(fn__).second)                                                                              \
            ZK_HIP_CHECK(ctx, hipFuncSetAttribute(fn__, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));                 \
    } while (0)

// launch + profile wrapper: ZK_LAUNCH(ctx, "name", kernel, grid, block, lds, args...)
#define ZK_LAUNCH(ctx, name, kernel, grid, block, lds, ...)                        \
    do {                                                                           \
        (ctx)->prof_begin(name);                                                   \
        hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);  \
        (ctx)->prof_end();                                                         \
        ZK_HIP_CHECK(ctx, hipGetLastError());                                      \
    } while (0)

// implemented in msm.hip / ntt.hip
int zk_msm_run(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint32_t *d_scalars, uint32_t *d_out_jac);
int zk_msm_run_batch(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                     const uint32_t *const *d_scalars, uint32_t *const *d_outs);
size_t zk_coord_limbs64(int curve, int group);  // u64 limbs per coordinate (Fq: 6/4, Fq2: 12/8)
int zk_bases_to_mont(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_canonical, const uint8_t *d_inf);
size_t zk_point_words(int curve, int group);
int zk_msm_auto_window(size_t n);
int zk_scalar_bits(int curve);  // bit length of the scalar-field modulus
int zk_bases_decompress(zkhip_ctx *ctx, zkhip_bases *b, const uint8_t *d_octets, uint32_t *d_err);  // wire.hip
void zk_graphs_clear(zkhip_ctx *ctx);  // destroy the cached MSM graphs
int zk_bases_precompute(zkhip_ctx *ctx, zkhip_bases *b);  // u32 words per affine point in device buffers
int zk_bases_from_mont(zkhip_ctx *ctx, const zkhip_bases *b, size_t offset, size_t n, uint32_t *d_out, uint8_t *d_inf);
int zk_bases_mul(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_base_canonical /* nullable: generator */, const uint32_t *d_scalars);
int zk_jac_sum(zkhip_ctx *ctx, int curve, int group, const uint32_t *d_pts, size_t count, uint32_t *d_out);
int zk_jac_to_affine(zkhip_ctx *ctx, int curve, int group, const uint32_t *d_jac, uint32_t *d_aff, uint8_t *d_inf);
int zk_ntt_run(zkhip_ctx *ctx, int curve, uint32_t *d_data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
               const uint64_t *coset);
int zk_ntt_extend(zkhip_ctx *ctx, int curve, uint32_t *d_coeffs, size_t log_m, size_t batch, const uint64_t *omega, uint32_t *d_out, size_t log_k,
                  const uint64_t *omega_big);
int zk_msm_host_reserve(zkhip_ctx *ctx, size_t n);  // zkhip.hip: ctx->msm_host_buf holds 512 B of result + n scalars
void zk_ntt_free_tables(zkhip_ctx *ctx);
void zk_dom_free_tables(zkhip_ctx *ctx);
