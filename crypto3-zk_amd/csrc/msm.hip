// Curve-independent host side of the MSM family: window selection, dispatch to the per-(curve, group) translation
// units (msm_bls_g1.hip, msm_bls_g2.hip, msm_bn_g1.hip, msm_bn_g2.hip; kernels and per-call logic in msm_core.hpp),
// batching, and the optional HIP-graph replay of repeated calls.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "msm_ops.hpp"
#include "zk_defs.hpp"

using namespace zkhip;

const MsmOps *zk_msm_ops(int curve, int group) {
    if (curve == CURVE_BLS12_381 && group == GROUP_G1) return zk_msm_ops_bls_g1();
    if (curve == CURVE_BLS12_381 && group == GROUP_G2) return zk_msm_ops_bls_g2();
    if (curve == CURVE_BN254 && group == GROUP_G1) return zk_msm_ops_bn_g1();
    if (curve == CURVE_BN254 && group == GROUP_G2) return zk_msm_ops_bn_g2();
    return nullptr;
}

#define ZK_OPS(curve, group)                         \
    const MsmOps *ops = zk_msm_ops(curve, group);    \
    if (!ops) return ZKHIP_ERR_INVALID

size_t zk_coord_limbs64(int curve, int group) {
    size_t fq = curve == CURVE_BLS12_381 ? 6 : 4;
    return fq * (group == GROUP_G2 ? 2 : 1);
}

int zk_bases_to_mont(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_canonical, const uint8_t *d_inf) {
    if (b->n == 0) return 0;
    ZK_OPS(b->curve, b->group);
    return ops->to_mont(ctx, b, d_canonical, d_inf);
}

size_t zk_point_words(int curve, int group) {
    const MsmOps *ops = zk_msm_ops(curve, group);
    return ops ? ops->point_words : 0;
}

int zk_bases_from_mont(zkhip_ctx *ctx, const zkhip_bases *b, size_t offset, size_t n, uint32_t *d_out, uint8_t *d_inf) {
    if (n == 0) return 0;
    ZK_OPS(b->curve, b->group);
    return ops->from_mont(ctx, b, offset, n, d_out, d_inf);
}

int zk_bases_mul(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_base_canonical, const uint32_t *d_scalars) {
    if (b->n == 0) return 0;
    ZK_OPS(b->curve, b->group);
    return ops->mul(ctx, b, d_base_canonical, d_scalars);
}

int zk_jac_to_affine(zkhip_ctx *ctx, int curve, int group, const uint32_t *d_jac, uint32_t *d_aff, uint8_t *d_inf) {
    ZK_OPS(curve, group);
    return ops->jac_to_affine(ctx, d_jac, d_aff, d_inf);
}

int zk_jac_sum(zkhip_ctx *ctx, int curve, int group, const uint32_t *d_pts, size_t count, uint32_t *d_out) {
    ZK_OPS(curve, group);
    return ops->jac_sum(ctx, d_pts, count, d_out);
}

int zk_bases_precompute(zkhip_ctx *ctx, zkhip_bases *b) {
    if (!b->tables() || b->n == 0) return 0;
    ZK_OPS(b->curve, b->group);
    return ops->precompute(ctx, b);
}

static int ilog2(size_t v) {
    int l = 0;
    while (((size_t)2 << l) <= v) ++l;
    return l;
}

int zk_scalar_bits(int curve) { return curve == CURVE_BLS12_381 ? 255 : 254; }  // bit length of r

// Window size from the number of points (tools/msm_window_sweep.py, profiles/r02_msm_window_sweep.json).  With window
// tables all W = ceil(bitlen(r) / c) windows of a point feed the same bucket set(s) (msm_core.hpp): the work is n W mixed
// additions + ~3 full additions per bucket of reduction, and the accumulation kernel wants ~2^19 lanes (buckets) of a few
// entries each.  From 2^19 points on one set of 2^19 buckets does it (c = 20: 13 windows instead of the 16 of c = 16;
// c = 21 has the same 13 windows and twice the buckets to reduce, c = 22 would need 2^21); below, c follows the sweep and c stops at 17 and the set is replicated (zk_msm_target_lanes) -- at 2^10 .. 2^16 points
// that is one set per window again, which the sweep shows to be as good as anything there.
// ZKHIP_MSM_WINDOW_BITS in the environment overrides the automatic choice (experiments).
int zk_msm_auto_window(size_t n) {
    static const int env_c = []() {
        const char *e = getenv("ZKHIP_MSM_WINDOW_BITS");
        return e ? atoi(e) : 0;
    }();
    if (env_c > 0) return std::max(2, std::min(ZK_MSM_MAX_C, env_c));
    int l = ilog2(n);
    if ((double)n >= 1.41421356237 * (double)((size_t)1 << l)) ++l;
    // profiles/r02_msm_window_sweep_g1.json: 2^10: c11 / S24, 2^12: c13 / S20, 2^14: c15 / S8, 2^16: c15 / S17, 2^18: c16 / S16 ~ c17 / S8,
    // 2^20, 2^21: c20 / S1
    const int c = l >= 19 ? 20 : (l == 18 ? 17 : (l == 17 ? 16 : (l >= 15 ? 15 : l + 1)));
    return std::max(2, std::min(ZK_MSM_MAX_C, c));
}

// Lanes the accumulation kernel should have at least (one per bucket): below that the bucket set is replicated into S
// sets (entry (i, w) -> set w mod S) that are folded bucket-wise afterwards.
size_t zk_msm_target_lanes() { return (size_t)1 << 19; }

static int zk_msm_run_direct(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint32_t *d_scalars, uint32_t *d_out_jac);

// ---- HIP graphs of repeated calls ------------------------------------------------------------------------------
// An MSM is ~30 dependent launches (a batch of four: ~100); a prover repeats the same call -- same bases, same
// resident buffers -- for every proof.  The third identical call is captured into a HIP graph (the first two ran
// directly, so every lazy allocation and attribute is settled) and later ones replay it: one graph launch instead
// of the launch sequence.  A graph bakes the workspace addresses, so it is dropped when the workspace is
// reallocated; profiling (per-kernel events) bypasses it.  MEASURED: no gain on ROCm 7.2 / MI355X -- 0.663 ms per
// 2^10-point MSM replayed against 0.660 ms issued directly, the same at every size up to 2^18 (the gaps between
// dependent kernels are on the device side, not in the host's launch path) -- so the option "msm_graphs" is OFF by
// default; the path stays for runtimes where graph launches are cheaper, and is covered by the GPU tests.
void zk_graphs_clear(zkhip_ctx *ctx) {
    for (auto &g : ctx->graphs) {
        if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
        if (g.second.d_ptrs) (void)hipFree(g.second.d_ptrs);
    }
    ctx->graphs.clear();
    ctx->graph_seen.clear();
}

template <class T>
static void key_add(std::string &k, const T &v) {
    k.append(reinterpret_cast<const char *>(&v), sizeof(T));
}
static void key_add_bases(std::string &k, const zkhip_bases *b) {
    key_add(k, b);
    key_add(k, b->d);
    key_add(k, b->n);
    key_add(k, b->c_tab);
    key_add(k, b->ntab);
    key_add(k, b->win_rank);
    key_add(k, b->win_world);
    key_add(k, b->curve);
    key_add(k, b->group);
}
static std::string key_begin(zkhip_ctx *ctx, char kind) {
    std::string k(1, kind);
    key_add(k, ctx->stream);
    key_add(k, ctx->opt_msm_window_bits);
    key_add(k, ctx->opt_msm_segment_log);
    key_add(k, ctx->opt_msm_sets);
    key_add(k, ctx->opt_msm_tail_quads);
    key_add(k, ctx->opt_msm_tail_fold);
    key_add(k, ctx->opt_msm_fold_run);
    key_add(k, ctx->opt_msm_tail_fold_g2);
    key_add(k, ctx->opt_msm_share_sort);
    key_add(k, ctx->opt_msm_sort_tile_log);  // the captured launch sequence depends on the sort's tile shape
    return k;
}

template <class Enqueue>
static int zk_graph_run(zkhip_ctx *ctx, const std::string &key, Enqueue &&enqueue, size_t nptrs = 0, uint32_t *const *ptrs = nullptr) {
    if (!ctx->opt_msm_graphs || ctx->prof.on || ctx->capturing) return enqueue();
    auto it = ctx->graphs.find(key);
    if (it != ctx->graphs.end()) {
        if (it->second.ws_epoch == ctx->ws_epoch) {
            ZK_HIP_CHECK(ctx, hipGraphLaunch(it->second.exec, ctx->stream));
            return 0;
        }
        (void)hipGraphExecDestroy(it->second.exec);  // the workspace moved: recapture below
        if (it->second.d_ptrs) (void)hipFree(it->second.d_ptrs);
        ctx->graphs.erase(it);
    }
    int &seen = ctx->graph_seen[key];
    if (seen < 0 || ++seen < 3) return enqueue();
    if (ctx->graphs.size() >= 64) zk_graphs_clear(ctx);
    ZkGraph g;
    if (nptrs) {
        if (hipMalloc(&g.d_ptrs, nptrs * sizeof(void *)) != hipSuccess) return enqueue();
        if (hipMemcpy(g.d_ptrs, ptrs, nptrs * sizeof(void *), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(g.d_ptrs);
            return enqueue();
        }
    }
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        if (g.d_ptrs) (void)hipFree(g.d_ptrs);
        ctx->graph_seen[key] = -1;
        return enqueue();
    }
    ctx->capturing = true;
    ctx->batch_dptrs_override = g.d_ptrs;
    const int rc = enqueue();
    ctx->batch_dptrs_override = nullptr;
    ctx->capturing = false;
    hipGraph_t graph = nullptr;
    const hipError_t e_end = hipStreamEndCapture(ctx->stream, &graph);
    bool ok = rc == 0 && e_end == hipSuccess && graph != nullptr;
    if (ok) ok = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {  // nothing ran during the capture: do it directly, and stop trying for this call shape
        (void)hipGetLastError();
        if (g.d_ptrs) (void)hipFree(g.d_ptrs);
        ctx->graph_seen[key] = -1;
        return enqueue();
    }
    g.ws_epoch = ctx->ws_epoch;
    ctx->graphs[key] = g;
    ZK_HIP_CHECK(ctx, hipGraphLaunch(g.exec, ctx->stream));
    return 0;
}

static int zk_msm_run_batch_direct(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                                   const uint32_t *const *d_scalars, uint32_t *const *d_outs);

int zk_msm_run_batch(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                     const uint32_t *const *d_scalars, uint32_t *const *d_outs) {
    if (count == 0) return 0;
    for (size_t i = 0; i < count; ++i)
        if (offsets[i] + ns[i] > bases[i]->n || ns[i] >= (1ull << 31)) return ZKHIP_ERR_RANGE;
    std::string key = key_begin(ctx, 'B');
    bool uniform = true;  // one tail group: the graph-owned pointer array then serves the single shared reduction
    for (size_t i = 0; i < count; ++i) {
        key_add_bases(key, bases[i]);
        key_add(key, offsets[i]);
        key_add(key, ns[i]);
        key_add(key, d_scalars[i]);
        key_add(key, d_outs[i]);
        uniform = uniform && bases[i]->tables() && bases[i]->curve == bases[0]->curve && bases[i]->group == bases[0]->group &&
                  bases[i]->c_tab == bases[0]->c_tab;
    }
    auto direct = [&]() { return zk_msm_run_batch_direct(ctx, count, bases, offsets, ns, d_scalars, d_outs); };
    if (!uniform || count < 2) return direct();
    return zk_graph_run(ctx, key, direct, count, d_outs);
}

static int zk_msm_run_batch_direct(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                                   const uint32_t *const *d_scalars, uint32_t *const *d_outs) {
    if (count == 0) return 0;
    for (size_t i = 0; i < count; ++i)
        if (offsets[i] + ns[i] > bases[i]->n || ns[i] >= (1ull << 31)) return ZKHIP_ERR_RANGE;
    // members that share (curve, group, window size) and have window tables form a batch; the rest run alone
    std::vector<char> done(count, 0);
    for (size_t i = 0; i < count; ++i) {
        if (done[i]) continue;
        std::vector<const zkhip_bases *> gb;
        std::vector<size_t> go, gn;
        std::vector<const uint32_t *> gs;
        std::vector<uint32_t *> gd;
        for (size_t j = i; j < count; ++j) {
            if (done[j] || !bases[j]->tables() || !bases[i]->tables() || bases[j]->curve != bases[i]->curve || bases[j]->group != bases[i]->group ||
                bases[j]->c_tab != bases[i]->c_tab)
                continue;
            done[j] = 1;
            gb.push_back(bases[j]);
            go.push_back(offsets[j]);
            gn.push_back(ns[j]);
            gs.push_back(d_scalars[j]);
            gd.push_back(d_outs[j]);
        }
        if (gb.size() >= 2) {
            ZK_OPS(gb[0]->curve, gb[0]->group);
            ZK_TRY(ops->batch(ctx, gb.size(), gb.data(), go.data(), gn.data(), gs.data(), gd.data()));
        } else {
            done[i] = 1;
            ZK_TRY(zk_msm_run_direct(ctx, bases[i], offsets[i], ns[i], d_scalars[i], d_outs[i]));
        }
    }
    return 0;
}

int zk_msm_run(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint32_t *d_scalars, uint32_t *d_out_jac) {
    if (offset + n > bases->n) return ZKHIP_ERR_RANGE;
    if (n == 0 || n >= (1ull << 31) || bases->win_world > 1) return zk_msm_run_direct(ctx, bases, offset, n, d_scalars, d_out_jac);
    std::string key = key_begin(ctx, 'S');
    key_add_bases(key, bases);
    key_add(key, offset);
    key_add(key, n);
    key_add(key, d_scalars);
    key_add(key, d_out_jac);
    return zk_graph_run(ctx, key, [&]() { return zk_msm_run_direct(ctx, bases, offset, n, d_scalars, d_out_jac); });
}

// window partition: a rank with no window of its own (more ranks than windows, or too few points for tables: then rank 0
// runs the whole MSM) contributes the point at infinity
static bool msm_rank_idle(const zkhip_bases *b) {
    if (b->win_world <= 1) return false;
    return b->tables() ? b->local_windows() == 0 : b->win_rank != 0;
}

static int zk_msm_run_direct(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint32_t *d_scalars, uint32_t *d_out_jac) {
    if (offset + n > bases->n) return ZKHIP_ERR_RANGE;
    if (n >= (1ull << 31)) return ZKHIP_ERR_RANGE;
    ZK_OPS(bases->curve, bases->group);
    if (n == 0 || msm_rank_idle(bases)) return ops->write_infinity(ctx, d_out_jac);
    return ops->run(ctx, bases, offset, n, d_scalars, d_out_jac);
}
