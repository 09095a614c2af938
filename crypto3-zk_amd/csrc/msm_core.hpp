// Pippenger multi-scalar multiplication for gfx950 (MI355X): kernels and per-(curve, group) host logic.
// Included by one translation unit per (curve, group) -- msm_bls_g1.hip, msm_bls_g2.hip, msm_bn_g1.hip, msm_bn_g2.hip --
// so that the four instantiations compile in parallel; msm.hip holds the curve-independent host side.
//
// Replaces algebra::multiexp<multiexp_method_BDLO12> / multiexp_with_mixed_addition as called at
//   zk/snark/systems/ppzksnark/r1cs_gg_ppzksnark/prover.hpp:108-139   (A, B, H, L queries)
//   zk/commitments/polynomial/kzg.hpp:143-148, 409-435                  (KZG commit)
//   zk/commitments/polynomial/knowledge_commitment_multiexp.hpp:107     (sparse (G2,G1) query)
//
// Algorithm.  Scalars are folded to |s| <= (r - 1) / 2 and cut into W signed digits of ~c bits (msm_recode.hpp).  With
// WINDOW TABLES (table w holds 2^off(w) P_i for every point, built once at upload) every (point, window) pair is an
// addition of equal weight: all W n pairs are sorted by |digit| into ONE set of B = 2^(c-1) buckets -- or, when that
// leaves too few buckets to fill the chip, into S sets (pair (i, w) goes to set w mod S) that are folded bucket-wise
// afterwards.  One lane per bucket accumulates its entries with XYZZ mixed additions; sum_b (b + 1) bucket[b] follows.
// Merging the windows lets c grow to ~log2(n) + 0..1 (c = 20 at 2^20 points: 13 windows instead of the 16 of c = 16,
// 19 % fewer additions) with the same number of lanes (2^19 buckets of ~26 entries against 16 x 2^15 of ~32).
// Without tables (fewer than 32 points, or tables that do not fit) the sets are the windows themselves and the
// final kernel combines them by a Horner pass.
//
// Data layout in HBM
//   bases    : nslots tables x n x {x, y}  lazy Montgomery limbs (fu.hpp), 16 words per coordinate, AoS, (0,0) = infinity
//   scalars  : n x 8 u32             canonical little-endian
//   dig      : W x n u32             signed digit of scalar i in (local) window w: (|d|-1) | sign<<31, NONE if d = 0
//   offs     : nb + 1 u32            exclusive prefix of the bucket sizes, nb = S * B
//   idx      : (#non-zero digits) u32  entry = (table slot * bases_n + point index) | sign<<31, grouped by (set, bucket)
//   order    : nb u32                bucket ids by descending size
//   buckets  : nb XYZZ               bucket sums
//   segsum / winsum                  per-workgroup weighted sums of the bucket reduction / per-set sums
//
// Kernels (all integer VALU; no MFMA -- this is modular arithmetic, not a dense contraction):
//   msm_digits_only   scalar -> signed digits                           (streams 32 B/scalar, coalesced)
//   msm_sort_*        two-level counting sort of the entries by (set, bucket), counters in LDS only, tiles staged in LDS
//   msm_scan_*        exclusive prefix (local scan, top scan, add-back), shared by both sorts
//   msm_size_*        order of the buckets by descending size
//   msm_bucket_acc_lds  one lane per bucket (G1) or one even / odd lane pair per bucket (G2, fu2_pair.hpp): gather affine
//                     points, XYZZ mixed additions, accumulator coordinates in LDS                          <- dominant
//   msm_plan_large / msm_bucket_large / msm_large_combine   buckets far above the mean, one workgroup per 4096-entry task
//   msm_bucket_merge  S > 1: fold the S equal-weight sets bucket by bucket (radix-4 tree)
//   msm_fold          (round 5, table-backed sets of >= 2^16 buckets) row and column sums of the bucket index b = h C + l: the reduction
//                     sum_b (b + 1) bucket[b] becomes sum_l (l + 1) COL[l] + C sum_h h ROW[h], i.e. the three kernels below over 2 sets of
//                     C ~ sqrt(B) buckets instead of one of B
//   msm_bucket_red    running sums over segments of L buckets + (seg L) * segment sum per lane, LDS tree per workgroup
//   msm_window_sum    LDS tree over the per-workgroup partials of a set
//   msm_final         (Horner over the windows when there are no tables,) XYZZ -> Jacobian, Montgomery -> canonical
//   msm_final_fold    the same after msm_fold: COL sum + 2^log2(C) ROW sum
// Point order inside a bucket depends on LDS-atomic arrival order; the group law is exact, so the sum
// (compared in affine) does not.
#pragma once
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#ifndef ZK_NOINLINE_MUL2
#define ZK_NOINLINE_MUL2 1  // G2 (Fq2) products stay out of line; G1 products are inlined (out of line measured 27 % slower)
#endif
#include "ctx.hpp"
#include "curve.hpp"
#include "msm_ops.hpp"
#include "msm_recode.hpp"
#include "msm_bucket_acc.hpp"
#include "fu_pair.hpp"
#include "fu_quad.hpp"

namespace {

using namespace zkhip;

constexpr uint32_t MSM_LARGE_BUCKET = 128;  // buckets above this many entries (and 4 x the mean) are split across workgroups
constexpr uint32_t MSM_LARGE_CHUNK = 4096;  // entries per task of a split bucket
#ifndef MSM_G1_THREADS
#define MSM_G1_THREADS 64  // lanes per bucket-accumulation workgroup: 64 measured 5 % faster than 128 / 256 (finer refill)
#endif
#ifndef MSM_G2_THREADS
#define MSM_G2_THREADS 256
#endif
#ifndef MSM_G2_WAVES
#define MSM_G2_WAVES 2
#endif

// exclusive scan of `count` u32 counters in three launches: per-block (1024 counters) local scan + block
// totals, scan of the totals by one workgroup, add-back.  offs[count] = total; cursor = copy of offs.
__global__ __launch_bounds__(256) void msm_scan_local(const uint32_t *__restrict__ hist, uint32_t count, uint32_t *__restrict__ offs,
                                                      uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x, base = blockIdx.x * 1024 + t * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = base + k < count ? hist[base + k] : 0;
        s += v[k];
    }
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t x = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < count) offs[base + k] = run;
        run += v[k];
    }
    if (t == 255) block_sums[blockIdx.x] = part[255];
}

__global__ __launch_bounds__(1024) void msm_scan_top(uint32_t *__restrict__ block_sums, uint32_t nblocks, uint32_t *__restrict__ total) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t lo = min(nblocks, t * per), hi = min(nblocks, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += block_sums[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t x = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t x = block_sums[i];
        block_sums[i] = run;
        run += x;
    }
    if (t == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void msm_scan_add(uint32_t *__restrict__ offs, uint32_t count, const uint32_t *__restrict__ block_sums,
                                                    uint32_t *__restrict__ cursor) {
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    const uint32_t add = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < count) {
            uint32_t x = offs[base + k] + add;
            offs[base + k] = x;
            cursor[base + k] = x;
        }
    }
}

// the same three steps in ONE launch for short arrays (small MSMs are bound by launch gaps, not by work): one workgroup walks the
// array in 1024-counter strips, carrying the running total
__global__ __launch_bounds__(1024) void msm_scan_single(const uint32_t *__restrict__ hist, uint32_t count, uint32_t *__restrict__ offs,
                                                        uint32_t *__restrict__ cursor) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < count; base += 1024) {
        const uint32_t v = base + t < count ? hist[base + t] : 0;
        part[t] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            uint32_t x = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += x;
            __syncthreads();
        }
        const uint32_t excl = carry + part[t] - v;
        if (base + t < count) {
            offs[base + t] = excl;
            if (cursor != offs) cursor[base + t] = excl;
        }
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) offs[count] = carry;
}
constexpr uint32_t MSM_SCAN_SINGLE_MAX = 1u << 16;

// exclusive scan of hist[0 .. count) into offs[0 .. count], offs[count] = total (bsums: scratch for the three-launch form)
inline int msm_scan(zkhip_ctx *ctx, const char *name, const uint32_t *hist, uint32_t count, uint32_t *offs, uint32_t *bsums) {
    if (count <= MSM_SCAN_SINGLE_MAX) {
        ZK_LAUNCH(ctx, name, msm_scan_single, dim3(1), dim3(1024), 0, hist, count, offs, offs);
        return 0;
    }
    const uint32_t nblk = (count + 1023) / 1024;
    ZK_LAUNCH(ctx, name, msm_scan_local, dim3(nblk), dim3(256), 0, hist, count, offs, bsums);
    ZK_LAUNCH(ctx, name, msm_scan_top, dim3(1), dim3(1024), 0, bsums, nblk, offs + count);
    ZK_LAUNCH(ctx, name, msm_scan_add, dim3(nblk), dim3(256), 0, offs, count, bsums, offs);
    return 0;
}

// Window partition (wrank of wworld): the carry chain runs over all windows, only the digits of windows
// w = wrank + k wworld are kept, as local window k.
template <class FR>
__global__ __launch_bounds__(256) void msm_digits_only(const uint32_t *__restrict__ scalars, uint32_t n, MsmWindows win, uint32_t wrank,
                                                       uint32_t wworld, uint32_t *__restrict__ dig) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    const uint32_t flip = msm_fold_scalar<FR>(scalars + (size_t)i * 8, s) ? 0x80000000u : 0u;  // |s| <= (r - 1) / 2
    uint32_t carry = 0, next = wrank, lw = 0;
    for (int w = 0; w < win.W; ++w) {
        uint32_t d = msm_recode(s, win.off(w), win.width(w), carry);
        if ((uint32_t)w != next) continue;
        dig[(size_t)lw * n + i] = d == DIG_NONE ? d : d ^ flip;
        next += wworld;
        ++lw;
    }
}

// ---- bucket sort without global atomics ------------------------------------------------------------------
// Random global atomics top out near 11 G/s on this chip (1.5 ms for the 16.8 M digits of a 2^20 MSM), so the
// (set, bucket) counting sort is done as a two-level partition whose counters all live in LDS:
//   pass 1  split the key range into "super-buckets" by the high bits of key = set * B + bucket: per-tile LDS
//           histogram -> global exclusive scan over (super-bucket, window, tile) -> per-tile LDS ranks;
//   pass 2  one workgroup per super-bucket: LDS counting sort over the low bits (<= 10), which also yields the final
//           bucket offsets.
// Both passes stage their tile / chunk in LDS first and write it out in runs per destination (one 4-byte store per
// entry straight to its slot cost 7x write amplification: PMC 619 MB written for 84 MB of output): with 2^14-entry
// tiles and <= 2^10 destinations per pass a run is >= 16 entries.
// A window-w tile covers points [tile * SORT_TILE, ...) of that window; the entry it emits for point i is
// (slot(w) * bases_n + base_off + i) | sign, the index of the table row the accumulation kernel gathers.
struct SortGeom {
    uint32_t n;           // points
    uint32_t W;           // (local) windows
    uint32_t B;           // buckets per set
    uint32_t S;           // sets: set(w) = w % S
    uint32_t lowb;        // low key bits (pass 2)
    uint32_t nsuper;      // ceil(S * B / 2^lowb)
    uint32_t tile;        // entries per tile
    uint32_t ntile;       // tiles per window
    uint32_t slot0;       // table slot of local window 0 (slot(lw) = slot0 + lw); all windows share slot 0 without tables
    uint32_t tables;      // 1: window lw reads table slot0 + lw; 0: every window reads the points themselves
    uint32_t bases_n;     // points per table
    uint32_t base_off;    // first point of this MSM inside the table
};
// Two tile shapes: 2^14 entries staged by 1024 lanes (~140 KiB of LDS: the workgroup owns its CU) when the MSM has the GPU
// to itself, and 2^12 entries by 256 lanes (~40 KiB) when another context's kernels share the chip (option
// "msm_sort_tile_log" = 12: the Groth16 shim's G2 multiexp on a second stream keeps ~98 KiB of every CU's LDS busy, a
// 140-KiB workgroup would wait for a CU to drain completely -- measured 7.8 ms for a 0.09-ms kernel).
struct SortBig {
    static constexpr uint32_t TILE = 16384, THREADS = 1024;
};
struct SortSmall {
    static constexpr uint32_t TILE = 4096, THREADS = 256;
};
constexpr uint32_t SORT_MAX_LOW = 10;   // <= 1024 destinations in pass 2
constexpr uint32_t SORT_MAX_SUPER = 1024;  // LDS: 3 counters per super-bucket next to the 128 KiB staged tile

ZK_D uint32_t sort_key(const SortGeom &g, uint32_t w, uint32_t d) { return (w % g.S) * g.B + (d & 0x7FFFFFFFu); }
ZK_D uint32_t sort_entry(const SortGeom &g, uint32_t w, uint32_t i, uint32_t d) {
    return ((g.tables ? (g.slot0 + w) * g.bases_n : 0u) + g.base_off + i) | (d & 0x80000000u);
}

// bh[sb * (W * ntile) + w * ntile + tile] = number of entries of window w, tile `tile`, in super-bucket sb
template <class SZ>
__global__ __launch_bounds__(SZ::THREADS) void msm_sort_hist(const uint32_t *__restrict__ dig, SortGeom g, uint32_t *__restrict__ bh) {
    constexpr uint32_t SORT_TILE = SZ::TILE, SORT_THREADS = SZ::THREADS;
    extern __shared__ uint32_t lh[];  // nsuper
    const uint32_t tile = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    for (uint32_t k = t; k < g.nsuper; k += SORT_THREADS) lh[k] = 0;
    __syncthreads();
    const uint32_t lo = tile * SORT_TILE, hi = min(g.n, lo + SORT_TILE);
    for (uint32_t i = lo + t; i < hi; i += SORT_THREADS) {
        uint32_t d = dig[(size_t)w * g.n + i];
        if (d != DIG_NONE) atomicAdd(&lh[sort_key(g, w, d) >> g.lowb], 1u);
    }
    __syncthreads();
    const size_t col = (size_t)w * g.ntile + tile, stride = (size_t)g.W * g.ntile;
    for (uint32_t k = t; k < g.nsuper; k += SORT_THREADS) bh[(size_t)k * stride + col] = lh[k];
}

// block-wide exclusive scan of `count` (<= SORT_MAX_SUPER) LDS counters in place; returns the total
template <uint32_t SORT_THREADS>
ZK_D uint32_t block_excl_scan(uint32_t *v, uint32_t count, uint32_t *scratch /* SORT_THREADS */) {
    const uint32_t t = threadIdx.x, per = (count + SORT_THREADS - 1) / SORT_THREADS;
    const uint32_t lo = min(count, t * per), hi = min(count, lo + per);
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; ++k) s += v[k];
    scratch[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
        uint32_t x = t >= d ? scratch[t - d] : 0;
        __syncthreads();
        scratch[t] += x;
        __syncthreads();
    }
    const uint32_t total = scratch[SORT_THREADS - 1];
    uint32_t run = scratch[t] - s;
    for (uint32_t k = lo; k < hi; ++k) {
        uint32_t x = v[k];
        v[k] = run;
        run += x;
    }
    __syncthreads();
    return total;
}

// tmp_idx / tmp_key: entries grouped by super-bucket; order inside a group is arbitrary.
template <class SZ>
__global__ __launch_bounds__(SZ::THREADS) void msm_sort_split(const uint32_t *__restrict__ dig, SortGeom g, const uint32_t *__restrict__ bo,
                                                              uint32_t *__restrict__ tmp_idx, uint16_t *__restrict__ tmp_key) {
    constexpr uint32_t SORT_TILE = SZ::TILE, SORT_THREADS = SZ::THREADS;
    extern __shared__ uint32_t sm[];
    uint32_t *loff = sm;                      // nsuper + 1: local exclusive offsets
    uint32_t *cur = loff + g.nsuper + 1;      // nsuper
    uint32_t *gbase = cur + g.nsuper;         // nsuper
    uint32_t *scratch = gbase + g.nsuper;     // SORT_THREADS
    uint32_t *sidx = scratch + SORT_THREADS;  // SORT_TILE
    uint32_t *skey = sidx + SORT_TILE;        // SORT_TILE full keys
    const uint32_t tile = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    const size_t col = (size_t)w * g.ntile + tile, stride = (size_t)g.W * g.ntile;
    for (uint32_t k = t; k < g.nsuper; k += SORT_THREADS) {
        loff[k] = 0;
        gbase[k] = bo[(size_t)k * stride + col];
    }
    __syncthreads();
    const uint32_t lo = tile * SORT_TILE, hi = min(g.n, lo + SORT_TILE);
    for (uint32_t i = lo + t; i < hi; i += SORT_THREADS) {  // this tile's histogram again (cheaper than reloading it)
        uint32_t d = dig[(size_t)w * g.n + i];
        if (d != DIG_NONE) atomicAdd(&loff[sort_key(g, w, d) >> g.lowb], 1u);
    }
    __syncthreads();
    const uint32_t total = block_excl_scan<SORT_THREADS>(loff, g.nsuper, scratch);
    if (t == 0) loff[g.nsuper] = total;
    for (uint32_t k = t; k < g.nsuper; k += SORT_THREADS) cur[k] = loff[k];
    __syncthreads();
    for (uint32_t i = lo + t; i < hi; i += SORT_THREADS) {
        uint32_t d = dig[(size_t)w * g.n + i];
        if (d == DIG_NONE) continue;
        const uint32_t key = sort_key(g, w, d);
        const uint32_t r = atomicAdd(&cur[key >> g.lowb], 1u);
        sidx[r] = sort_entry(g, w, i, d);
        skey[r] = key;
    }
    __syncthreads();
    const uint32_t lmask = (1u << g.lowb) - 1;
    for (uint32_t s = t; s < total; s += SORT_THREADS) {
        const uint32_t key = skey[s], bin = key >> g.lowb, pos = gbase[bin] + (s - loff[bin]);
        tmp_idx[pos] = sidx[s];
        tmp_key[pos] = (uint16_t)(key & lmask);
    }
}

// one workgroup per super-bucket: final order + bucket offsets offs[sb * 2^lowb + low].
// The group is sorted chunk by chunk inside LDS and written out in runs per key (same reason as above).
// NLOW = 2^lowb <= 1024 counters, CPT = NLOW / THREADS of them per lane.
constexpr uint32_t SORT_NLOW_MAX = 1u << SORT_MAX_LOW;

template <uint32_t THREADS>
ZK_D void block_scan_counts(const uint32_t *cnt, uint32_t *excl, uint32_t *scratch) {  // excl[k] = sum_{j < k} cnt[j], k < SORT_NLOW_MAX
    constexpr uint32_t CPT = SORT_NLOW_MAX / THREADS;
    const uint32_t t = threadIdx.x;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < CPT; ++k) s += cnt[t * CPT + k];
    scratch[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < THREADS; d <<= 1) {
        uint32_t x = t >= d ? scratch[t - d] : 0;
        __syncthreads();
        scratch[t] += x;
        __syncthreads();
    }
    uint32_t run = scratch[t] - s;
#pragma unroll
    for (uint32_t k = 0; k < CPT; ++k) {
        excl[t * CPT + k] = run;
        run += cnt[t * CPT + k];
    }
    __syncthreads();
}

template <class SZ>
__global__ __launch_bounds__(SZ::THREADS) void msm_sort_final(const uint32_t *__restrict__ tmp_idx, const uint16_t *__restrict__ tmp_key, SortGeom g,
                                                              uint32_t nb, const uint32_t *__restrict__ bo, uint32_t *__restrict__ offs,
                                                              uint32_t *__restrict__ idx) {
    constexpr uint32_t SORT_TILE = SZ::TILE, SORT_THREADS = SZ::THREADS;
    __shared__ uint32_t cnt[SORT_NLOW_MAX], cursor[SORT_NLOW_MAX], loff[SORT_NLOW_MAX], cur[SORT_NLOW_MAX], scratch[SORT_THREADS];
    extern __shared__ uint32_t sm[];
    uint32_t *sidx = sm;                                            // SORT_TILE
    uint16_t *skey = reinterpret_cast<uint16_t *>(sidx + SORT_TILE);  // SORT_TILE
    const uint32_t grp = blockIdx.x, t = threadIdx.x;
    const size_t stride = (size_t)g.W * g.ntile;
    const uint32_t start = bo[(size_t)grp * stride];
    const uint32_t end = bo[(size_t)(grp + 1) * stride];  // bo has one trailing entry = total (grp + 1 == nsuper)
    const uint32_t nlow = 1u << g.lowb;                   // <= SORT_NLOW_MAX
    for (uint32_t k = t; k < SORT_NLOW_MAX; k += SORT_THREADS) cnt[k] = 0;
    __syncthreads();
    for (uint32_t k = start + t; k < end; k += SORT_THREADS) atomicAdd(&cnt[tmp_key[k]], 1u);
    __syncthreads();
    block_scan_counts<SORT_THREADS>(cnt, cursor, scratch);  // cursor[key] = offset of key inside the group
    for (uint32_t k = t; k < nlow; k += SORT_THREADS) {
        const uint32_t bucket = grp * nlow + k;
        cursor[k] += start;  // next free slot of key k
        if (bucket < nb) offs[bucket] = cursor[k];
    }
    if (grp + 1 == g.nsuper && t == 0) offs[nb] = end;
    __syncthreads();
    for (uint32_t c0 = start; c0 < end; c0 += SORT_TILE) {
        const uint32_t c1 = min(end, c0 + SORT_TILE);
        for (uint32_t k = t; k < SORT_NLOW_MAX; k += SORT_THREADS) cnt[k] = 0;
        __syncthreads();
        for (uint32_t k = c0 + t; k < c1; k += SORT_THREADS) atomicAdd(&cnt[tmp_key[k]], 1u);
        __syncthreads();
        block_scan_counts<SORT_THREADS>(cnt, loff, scratch);
        for (uint32_t k = t; k < SORT_NLOW_MAX; k += SORT_THREADS) cur[k] = loff[k];
        __syncthreads();
        for (uint32_t k = c0 + t; k < c1; k += SORT_THREADS) {
            const uint32_t key = tmp_key[k];
            const uint32_t r = atomicAdd(&cur[key], 1u);
            sidx[r] = tmp_idx[k];
            skey[r] = (uint16_t)key;
        }
        __syncthreads();
        for (uint32_t s = t; s < c1 - c0; s += SORT_THREADS) {
            const uint32_t key = skey[s];
            idx[cursor[key] + (s - loff[key])] = sidx[s];
        }
        __syncthreads();
        for (uint32_t k = t; k < SORT_NLOW_MAX; k += SORT_THREADS) cursor[k] += cnt[k];
        __syncthreads();
    }
}

// ---- order of buckets by descending size (counting sort, counters in LDS) --------------------------------------
// Lanes of a wave run in lockstep (a wave costs the largest bucket among its 64) and a workgroup holds its
// registers until its last wave retires, so buckets are handed out in globally sorted order: every wave and every
// workgroup sees near-equal trip counts, long buckets start first, empty and large (split elsewhere) buckets
// collect at the end.  `large` is the split threshold of this MSM (max(MSM_LARGE_BUCKET, 4 x mean bucket size)).
// bin 0: MSM_LARGE_BUCKET < size <= large; bin 1 + MSM_LARGE_BUCKET - size for 1 <= size <= MSM_LARGE_BUCKET; last bin:
// empty or split.
constexpr uint32_t SIZE_BINS = MSM_LARGE_BUCKET + 2;

ZK_D uint32_t size_bin(uint32_t size, uint32_t large) {
    if (size == 0 || size > large) return SIZE_BINS - 1;
    return size > MSM_LARGE_BUCKET ? 0 : 1 + MSM_LARGE_BUCKET - size;
}

__global__ __launch_bounds__(256) void msm_size_hist(const uint32_t *__restrict__ offs, uint32_t nbuckets, uint32_t nblocks,
                                                     uint32_t large, uint32_t *__restrict__ bh) {
    __shared__ uint32_t lh[SIZE_BINS];
    const uint32_t t = threadIdx.x;
    if (t < SIZE_BINS) lh[t] = 0;
    __syncthreads();
    for (uint32_t g = blockIdx.x * 1024 + t; g < min(nbuckets, (blockIdx.x + 1) * 1024); g += 256)
        atomicAdd(&lh[size_bin(offs[g + 1] - offs[g], large)], 1u);
    __syncthreads();
    if (t < SIZE_BINS) bh[(size_t)t * nblocks + blockIdx.x] = lh[t];
}

__global__ __launch_bounds__(256) void msm_size_scatter(const uint32_t *__restrict__ offs, uint32_t nbuckets, uint32_t nblocks,
                                                        uint32_t large, const uint32_t *__restrict__ bo, uint32_t *__restrict__ order) {
    __shared__ uint32_t cur[SIZE_BINS];
    const uint32_t t = threadIdx.x;
    if (t < SIZE_BINS) cur[t] = bo[(size_t)t * nblocks + blockIdx.x];
    __syncthreads();
    for (uint32_t g = blockIdx.x * 1024 + t; g < min(nbuckets, (blockIdx.x + 1) * 1024); g += 256)
        order[atomicAdd(&cur[size_bin(offs[g + 1] - offs[g], large)], 1u)] = g;
}

// ---- large buckets ---------------------------------------------------------------------------------------
// Skewed scalars (Groth16 witnesses full of 0/1 values, a top window with only a few significant bits)
// put thousands of points into single buckets; one lane per bucket would serialise them.  Buckets above
// the threshold are cut into tasks of MSM_LARGE_CHUNK entries, one workgroup per task
// (strided mixed additions + LDS tree), and the per-task partial sums of a bucket are folded afterwards.
// plan[0] = number of tasks, plan[1] = number of large buckets; tasks[t] = {bucket, lo, hi};
// large[j] = {bucket, first task, task count}.
__global__ __launch_bounds__(256) void msm_plan_large(const uint32_t *__restrict__ offs, uint32_t nbuckets, uint32_t *__restrict__ plan,
                                                      uint32_t *__restrict__ tasks, uint32_t *__restrict__ large, uint32_t task_cap,
                                                      uint32_t large_cap, uint32_t thresh, uint32_t *__restrict__ status) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nbuckets) return;
    uint32_t lo = offs[g], hi = offs[g + 1], size = hi - lo;
    if (size <= thresh) return;
    uint32_t nt = (size + MSM_LARGE_CHUNK - 1) / MSM_LARGE_CHUNK;
    uint32_t first = atomicAdd(&plan[0], nt);
    uint32_t slot = atomicAdd(&plan[1], 1u);
    if (first + nt > task_cap || slot >= large_cap) {  // capacities are worst-case (see host): never expected, but never silent
        atomicOr(status, ZK_STATUS_MSM_PLAN_OVERFLOW);
        return;
    }
    for (uint32_t t = 0; t < nt; ++t) {
        tasks[3 * (first + t)] = g;
        tasks[3 * (first + t) + 1] = lo + t * MSM_LARGE_CHUNK;
        tasks[3 * (first + t) + 2] = min(hi, lo + (t + 1) * MSM_LARGE_CHUNK);
    }
    large[3 * slot] = g;
    large[3 * slot + 1] = first;
    large[3 * slot + 2] = nt;
}

// (all kernels below: F is the type a lane holds, LPB lanes share one point -- fu2_pair.hpp; `t` is the point slot)
template <class F, int LPB>
__global__ __launch_bounds__(128) void msm_bucket_large(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ idx,
                                                        const uint32_t *__restrict__ plan, const uint32_t *__restrict__ tasks,
                                                        uint32_t *__restrict__ partials) {
    constexpr int NL = FieldOps<F>::WORDS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t ntasks = plan[0];
    for (uint32_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const uint32_t lo = tasks[3 * task + 1], hi = tasks[3 * task + 2], t = threadIdx.x / LPB;
        XYZZ<F> acc = XYZZ<F>::infinity();
        for (uint32_t k = lo + t; k < hi; k += blockDim.x / LPB) {
            uint32_t e = idx[k];
            Affine<F> p = affine_load<F>(bases + (size_t)(e & 0x7FFFFFFFu) * (2 * NL));
            acc = xyzz_madd(acc, p, (e >> 31) != 0);
        }
        xyzz_store<F>(lds + (size_t)t * (4 * NL), acc);
        __syncthreads();
        for (uint32_t d = blockDim.x / LPB / 2; d >= 1; d >>= 1) {
            if (t < d) {
                acc = xyzz_add(xyzz_load<F>(lds + (size_t)t * (4 * NL)), xyzz_load<F>(lds + (size_t)(t + d) * (4 * NL)));
                xyzz_store<F>(lds + (size_t)t * (4 * NL), acc);
            }
            __syncthreads();
        }
        if (t == 0) xyzz_store<F>(partials + (size_t)task * (4 * NL), acc);
        __syncthreads();
    }
}

// one wave per large bucket: lanes fold the bucket's task partials in strides, then an LDS tree
template <class F, int LPB>
__global__ __launch_bounds__(64) void msm_large_combine(const uint32_t *__restrict__ plan, const uint32_t *__restrict__ large,
                                                        const uint32_t *__restrict__ partials, uint32_t *__restrict__ buckets) {
    constexpr int NL = FieldOps<F>::WORDS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t nlarge = plan[1], t = threadIdx.x / LPB;
    for (uint32_t j = blockIdx.x; j < nlarge; j += gridDim.x) {
        const uint32_t g = large[3 * j], first = large[3 * j + 1], nt = large[3 * j + 2];
        XYZZ<F> acc = XYZZ<F>::infinity();
        for (uint32_t k = t; k < nt; k += 64 / LPB) acc = xyzz_add(acc, xyzz_load<F>(partials + (size_t)(first + k) * (4 * NL)));
        if (nt > 1) {  // uniform over the wave
            xyzz_store<F>(lds + (size_t)t * (4 * NL), acc);
            __syncthreads();
            for (uint32_t d = 32 / LPB; d >= 1; d >>= 1) {
                if (t < d && t + d < nt) {
                    acc = xyzz_add(xyzz_load<F>(lds + (size_t)t * (4 * NL)), xyzz_load<F>(lds + (size_t)(t + d) * (4 * NL)));
                    xyzz_store<F>(lds + (size_t)t * (4 * NL), acc);
                }
                __syncthreads();
            }
        }
        if (t == 0) xyzz_store<F>(buckets + (size_t)g * (4 * NL), acc);
        __syncthreads();
    }
}

// buckets[s][b] += buckets[s + q][b] + buckets[s + 2q][b] + buckets[s + 3q][b] for s < q (sets >= cur do not exist):
// one radix-4 level of the tree that folds the equal-weight sets
template <class F, int LPB>
__global__ __launch_bounds__(256) void msm_bucket_merge(uint32_t *__restrict__ buckets, uint32_t B, uint32_t q, uint32_t cur) {
    constexpr int NL = FieldOps<F>::WORDS;
    uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / LPB;  // = s * B + b
    if (g >= q * B) return;
    const uint32_t s = g / B;
    uint32_t *dst = buckets + (size_t)g * (4 * NL);
    XYZZ<F> acc = xyzz_load<F>(dst);
    for (uint32_t k = 1; k < 4; ++k)
        if (s + k * q < cur) acc = xyzz_add(acc, xyzz_load<F>(buckets + ((size_t)g + (size_t)k * q * B) * (4 * NL)));
    xyzz_store<F>(dst, acc);
}

// ---- tail: sum_b (b + 1) * bucket[b] per set -----------------------------------------------------------
// Every operation here is a full (projective) addition or doubling, a few thousand dependent instructions that a
// lone wave issues at one per ~4 cycles: the tail is bound by the LENGTH of its dependency chain as long as its lanes
// fit the chip.  Stage 1 gives every lane one segment of L buckets and folds the lanes of a workgroup in an LDS tree;
// stage 2 folds the per-workgroup partials of a set (one more tree).
// segment `seg` covers buckets [seg*L, seg*L + L) (bucket b holds digit value b + 1):
//   sum_b (b + 1) * bucket[b]  restricted to the segment
//       = sum_b (b - seg*L + 1) * bucket[b]  +  (seg*L) * sum_b bucket[b]
constexpr int MSM_TAIL_THREADS = 256;

template <class F>
ZK_D XYZZ<F> block_tree_sum(uint32_t *lds, XYZZ<F> acc, uint32_t t, uint32_t nthreads) {
    constexpr int NL = FieldOps<F>::WORDS;
    xyzz_store<F>(lds + (size_t)t * (4 * NL), acc);
    __syncthreads();
    for (uint32_t d = nthreads / 2; d >= 1; d >>= 1) {
        if (t < d) {
            acc = xyzz_add(xyzz_load<F>(lds + (size_t)t * (4 * NL)), xyzz_load<F>(lds + (size_t)(t + d) * (4 * NL)));
            xyzz_store<F>(lds + (size_t)t * (4 * NL), acc);
        }
        __syncthreads();
    }
    return acc;  // lane 0: the sum
}

// grid = sets x nblk workgroups; partial[s * nblk + j] = weighted sum of segments [SLOTS j, SLOTS j + SLOTS) of set s.
// Two waves per SIMD: the G1 kernels then spill (pairs 154 registers, one lane 765); bound 1 has no spills (330 / 446 VGPRs) and
// measures the same for a single MSM (0.699 against 0.716 ms at 2^19 buckets) but worse for batches, whose many waves want
// the second slot (50 x 2^19 buckets: 17.4 against 13.7 ms).
template <class F, int LPB>
#ifndef ZK_TAIL_WAVES
#define ZK_TAIL_WAVES 2  // workgroups of the reduction per CU the register allocation leaves room for (1: no spills; measured)
#endif
__global__ __launch_bounds__(MSM_TAIL_THREADS, ZK_TAIL_WAVES) void msm_bucket_red(const uint32_t *__restrict__ buckets, uint32_t B, uint32_t L, uint32_t nseg,
                                                                   uint32_t nblk, uint32_t *__restrict__ partial) {
    constexpr int NL = FieldOps<F>::WORDS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t SLOTS = MSM_TAIL_THREADS / LPB;
    const uint32_t t = threadIdx.x / LPB, w = blockIdx.x / nblk, seg = (blockIdx.x % nblk) * SLOTS + t;
    XYZZ<F> sum = XYZZ<F>::infinity();
    if (seg < nseg) {
        const uint32_t *base = buckets + ((size_t)w * B + (size_t)seg * L) * (4 * NL);
        if (L == 1) {
            sum = xyzz_mul_small(xyzz_load<F>(base), seg + 1);
        } else {
            XYZZ<F> run = XYZZ<F>::infinity();
            for (int b = (int)L - 1; b >= 0; --b) {
                run = xyzz_add(run, xyzz_load<F>(base + (size_t)b * (4 * NL)));
                sum = xyzz_add(sum, run);
            }
            if (seg != 0) sum = xyzz_add(sum, xyzz_mul_small(run, seg * L));
        }
    }
    sum = block_tree_sum<F>(lds, sum, t, SLOTS);
    if (t == 0) xyzz_store<F>(partial + (size_t)blockIdx.x * (4 * NL), sum);
}

// one workgroup (64 or 256 lanes: msm_window_threads) per set: winsum[s] = sum_j partial[s][j]
template <class F, int LPB>
__global__ __launch_bounds__(256) void msm_window_sum(const uint32_t *__restrict__ partial, uint32_t nblk, uint32_t *__restrict__ winsum) {
    constexpr int NL = FieldOps<F>::WORDS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t w = blockIdx.x, t = threadIdx.x / LPB, slots = blockDim.x / LPB;
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (uint32_t j = t; j < nblk; j += slots) acc = xyzz_add(acc, xyzz_load<F>(partial + ((size_t)w * nblk + j) * (4 * NL)));
    if (nblk > 1) acc = block_tree_sum<F>(lds, acc, t, slots);
    if (t == 0) xyzz_store<F>(winsum + (size_t)w * (4 * NL), acc);
}
// every operation is ~10 us of one wave: a tree level more (256 lanes) is cheaper than three more serial additions per lane
inline uint32_t msm_window_threads(uint32_t nblk, int lanes_per_point) { return (size_t)nblk * lanes_per_point > 128 ? 256u : 64u; }

// ---- two-level tail (round 5) -----------------------------------------------------------------------------------------------
// The running sums above are a DEPENDENT chain of 2 L additions per lane plus the (seg L) multiple: at 2^19 buckets ~76 operations of
// ~9 us each, 0.8 ms of a 2.85-ms MSM with one wave per SIMD.  Split the bucket index instead: bucket b = h C + l (l < C, h < R,
// B = R C) holds digit value b + 1 = h C + (l + 1), so
//     sum_b (b + 1) bucket[b]  =  sum_l (l + 1) COL[l]  +  C sum_h h ROW[h],     COL[l] = sum_h bucket[h C + l],   ROW[h] = sum_l bucket[h C + l]
// -- 2 B additions like the running sums, but every one of them sits in an independent SUM, and what is left for the dependent chain is the old
// tail over 2 sets of C = 2^ceil(log2(B) / 2) buckets (COL as it is, ROW[h] in bucket h - 1, the rest empty) and log2(C) doublings of the second
// set's sum.  (The "split the windows" step of bucket-method MSMs on GPUs, applied to the ONE merged bucket set the window tables leave.)
struct MsmFold {
    uint32_t B, C, R, log_c;
    uint32_t run;           // buckets a lane sums before the workgroup's tree takes over
    uint32_t m_row, m_col;  // runs (= LDS partial sums) per row: C / run, per column: R / run
};
inline MsmFold msm_fold_geom(const zkhip_ctx *ctx, uint32_t B) {
    MsmFold g;
    int lb = 0;
    while ((1u << lb) < B) ++lb;
    g.B = B;
    g.log_c = (uint32_t)((lb + 1) / 2);
    g.C = 1u << g.log_c;
    g.R = B >> g.log_c;
    // measured (tools/msm_profile.py, G1): 2^19 buckets: 8 and 16 the same (0.31 ms), 32 slower (0.52: half the SIMDs idle); 2^16 buckets: 8: 0.150,
    // 4: 0.108, 2: 0.094 ms
    g.run = ctx->opt_msm_fold_run > 0 ? (uint32_t)ctx->opt_msm_fold_run : (B >= (1u << 18) ? 8u : 4u);
    g.run = std::max(1u, std::min(g.run, g.R));
    g.m_row = g.C / g.run;
    g.m_col = g.R / g.run;
    return g;
}

// grid = sets x B / (RUN SA) workgroups, SA = threads / LA points: the first half of a set's workgroups sum ROWS (SA / m_row whole rows each), the second
// half COLUMNS (SA / m_col whole columns each).
//   phase 1, bound by WORK: every point slot sums RUN buckets of its row / column with the lane shape FA (one lane per point for G1: an addition
//            in the fewest issue slots), interleaved so that adjacent lanes read adjacent buckets, and leaves the sum in LDS;
//   phase 2, bound by LATENCY: binary trees over the m partial sums of each row / column in LDS with the lane shape FT (G1: lane quads);
//   level2[2 set][l] = COL[l], level2[2 set + 1][h - 1] = ROW[h] (row 0 has weight 0: dropped; entries R - 1 ... C - 1 of the second set stay
//   empty = zeroed by the caller).
template <class FA, int LA, class FT, int LT>
__global__ __launch_bounds__(MSM_TAIL_THREADS, ZK_TAIL_WAVES) void msm_fold(const uint32_t *__restrict__ buckets, MsmFold g, uint32_t *__restrict__ level2) {
    constexpr int NL = FieldOps<FA>::WORDS;
    constexpr uint32_t SA = MSM_TAIL_THREADS / LA, ST = MSM_TAIL_THREADS / LT, PW = 4 * NL;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t half_wgs = g.B / (SA * g.run);
    const uint32_t w = blockIdx.x / (2 * half_wgs), k = blockIdx.x % (2 * half_wgs);
    const bool col = k >= half_wgs;
    const uint32_t wg = col ? k - half_wgs : k, s = threadIdx.x / LA;
    const uint32_t m = col ? g.m_col : g.m_row, G = SA / m;  // partial sums per output, outputs of this workgroup
    uint32_t first, stride, slot;
    if (!col) {  // slot (r, j): buckets h C + j + i m_row of row h = wg G + r
        first = (wg * G + s / m) * g.C + s % m;
        stride = m;
        slot = s;
    } else {  // slot (j, ll), ll fastest: buckets (j + i m_col) C + l of column l = wg G + ll; its sum goes to LDS slot ll m_col + j
        const uint32_t j = s / G, ll = s % G;
        first = j * g.C + wg * G + ll;
        stride = m * g.C;
        slot = ll * m + j;
    }
    {
        const uint32_t *src = buckets + ((size_t)w * g.B + first) * PW;
        XYZZ<FA> acc = xyzz_load<FA>(src);
        for (uint32_t i = 1; i < g.run; ++i) acc = xyzz_add(acc, xyzz_load<FA>(src + (size_t)i * stride * PW));
        xyzz_store<FA>(lds + (size_t)slot * PW, acc);
    }
    __syncthreads();
    const uint32_t t = threadIdx.x / LT;
    for (uint32_t half = m / 2; half >= 1; half >>= 1) {
        const uint32_t adds = G * half;
        for (uint32_t idx = t; idx < adds; idx += ST) {
            uint32_t *p = lds + (size_t)((idx / half) * m + idx % half) * PW;
            xyzz_store<FT>(p, xyzz_add(xyzz_load<FT>(p), xyzz_load<FT>(p + (size_t)half * PW)));
        }
        __syncthreads();
    }
    for (uint32_t i = t; i < G; i += ST) {
        const uint32_t o = wg * G + i;  // row h / column l
        if (col || o != 0)
            xyzz_store<FT>(level2 + (((size_t)2 * w + (col ? 0 : 1)) * g.C + (col ? o : o - 1)) * PW, xyzz_load<FT>(lds + (size_t)i * m * PW));
    }
}

// out[i] = winsum[2 i] + 2^shift winsum[2 i + 1] as canonical Jacobian; outs == nullptr: the single output `out`
template <class F, int LPB>
__global__ __launch_bounds__(64) void msm_final_fold(const uint32_t *__restrict__ winsum, uint32_t count, uint32_t shift, uint32_t *const *__restrict__ outs,
                                                     uint32_t *__restrict__ out) {
    constexpr int NL = FieldOps<F>::WORDS;
    if (blockIdx.x >= count || threadIdx.x >= LPB) return;
    XYZZ<F> acc = xyzz_load<F>(winsum + ((size_t)2 * blockIdx.x + 1) * (4 * NL));
    if (!acc.is_inf())
        for (uint32_t i = 0; i < shift; ++i) acc = xyzz_dbl(acc);
    acc = xyzz_add(acc, xyzz_load<F>(winsum + (size_t)2 * blockIdx.x * (4 * NL)));
    Jacobian<F> j = xyzz_to_jacobian(acc);
    constexpr int CW = FieldOps<F>::CANON_WORDS;
    uint32_t *dst = outs ? outs[blockIdx.x] : out;
    FieldOps<F>::to_canonical(dst, j.X);
    FieldOps<F>::to_canonical(dst + CW, j.Y);
    FieldOps<F>::to_canonical(dst + 2 * CW, j.Z);
}

// result = sum_w 2^off(w) winsum[w]  (Horner from the top window; a single set with tables), emitted as canonical Jacobian
template <class F, int LPB>
__global__ __launch_bounds__(64) void msm_final(const uint32_t *__restrict__ winsum, int W, MsmWindows win, uint32_t *__restrict__ out_jac) {
    constexpr int NL = FieldOps<F>::WORDS;
    if (blockIdx.x != 0 || threadIdx.x >= LPB) return;
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (int w = W - 1; w >= 0; --w) {
        if (!acc.is_inf())
            for (int i = 0; i < win.width(w); ++i) acc = xyzz_dbl(acc);
        acc = xyzz_add(acc, xyzz_load<F>(winsum + (size_t)w * (4 * NL)));
    }
    Jacobian<F> j = xyzz_to_jacobian(acc);
    constexpr int CW = FieldOps<F>::CANON_WORDS;
    FieldOps<F>::to_canonical(out_jac, j.X);
    FieldOps<F>::to_canonical(out_jac + CW, j.Y);
    FieldOps<F>::to_canonical(out_jac + 2 * CW, j.Z);
}

// ---- bases maintenance ----------------------------------------------------------------------------
// canonical affine (x | y, CANON_WORDS each) -> device form (Montgomery, WORDS each); flagged points -> (0, 0)
template <class F>
__global__ __launch_bounds__(256) void bases_to_mont(const uint32_t *__restrict__ canon, const uint8_t *__restrict__ inf, uint32_t n,
                                                     uint32_t *__restrict__ pts) {
    typedef FieldOps<F> O;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *c = canon + (size_t)i * (2 * O::CANON_WORDS);
    Affine<F> a;
    if (inf != nullptr && inf[i]) a = Affine<F>::infinity();
    else a = {O::from_canonical(c), O::from_canonical(c + O::CANON_WORDS)};
    affine_store<F>(pts + (size_t)i * (2 * O::WORDS), a);
}

template <class F>
__global__ __launch_bounds__(256) void bases_from_mont(const uint32_t *__restrict__ pts, uint32_t n, uint32_t *__restrict__ out,
                                                       uint8_t *__restrict__ inf) {
    typedef FieldOps<F> O;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> a = affine_load<F>(pts + (size_t)i * (2 * O::WORDS));
    inf[i] = a.is_inf() ? 1 : 0;
    uint32_t *o = out + (size_t)i * (2 * O::CANON_WORDS);
    O::to_canonical(o, a.x);
    O::to_canonical(o + O::CANON_WORDS, a.y);
}

// pts[i] = scalars[i] * base, double-and-add from the top bit, then one inversion per point
template <class F>
__global__ __launch_bounds__(64) void bases_mul(uint32_t *__restrict__ pts, const uint32_t *__restrict__ base_canonical,
                                                const uint32_t *__restrict__ scalars, uint32_t n) {
    constexpr int NL = FieldOps<F>::WORDS;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> g = {FieldOps<F>::from_canonical(base_canonical), FieldOps<F>::from_canonical(base_canonical + FieldOps<F>::CANON_WORDS)};
    const uint32_t *s = scalars + (size_t)i * 8;
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (int b = 255; b >= 0; --b) {
        acc = xyzz_dbl(acc);
        if ((s[b >> 5] >> (b & 31)) & 1) acc = xyzz_madd(acc, g);
    }
    affine_store<F>(pts + (size_t)i * (2 * NL), xyzz_to_affine(acc));
}

// ---- fixed-base batch exponentiation (batch_exp of generator.hpp:187-214: every point is a multiple of ONE base) --------------------
// tab[(w << 8) + d] = d 2^(8 w) base, affine, d in [1, 256), w < 32.  One lane per window: 8 w doublings, 254 mixed additions in
// XYZZ parked in `tmp` (5 field elements per entry: X, Y, ZZ, ZZZ, prefix product), ONE inversion (Montgomery's trick), back down.
constexpr int FIXED_WBITS = 8, FIXED_NW = 32, FIXED_ROW = 1 << FIXED_WBITS;
template <class F>
__global__ __launch_bounds__(64) void bases_fixed_table(const uint32_t *__restrict__ base_canonical, uint32_t *__restrict__ tab, uint32_t *__restrict__ tmp) {
    typedef FieldOps<F> O;
    constexpr int NL = O::WORDS;
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= FIXED_NW) return;
    Affine<F> g = {O::from_canonical(base_canonical), O::from_canonical(base_canonical + O::CANON_WORDS)};
    XYZZ<F> P = XYZZ<F>::from_affine(g);
    for (uint32_t k = 0; k < w * FIXED_WBITS; ++k) P = xyzz_dbl(P);
    uint32_t *mine = tmp + (size_t)w * FIXED_ROW * (5 * NL);
    XYZZ<F> acc = P;
    F pre = F::one();
    for (uint32_t d = 1; d < FIXED_ROW; ++d) {  // entry d: d P (P is not affine: full additions; the table is built once per call)
        if (d > 1) acc = xyzz_add(acc, P);
        uint32_t *slot = mine + (size_t)d * (5 * NL);
        xyzz_store<F>(slot, acc);
        pre = O::mul(pre, O::mul(acc.ZZ, acc.ZZZ));
        O::store(slot + 4 * NL, pre);
    }
    F inv = O::inv(pre);
    for (uint32_t d = FIXED_ROW; d-- > 1;) {
        const uint32_t *slot = mine + (size_t)d * (5 * NL);
        XYZZ<F> q = xyzz_load<F>(slot);
        F before = d > 1 ? O::load(mine + (size_t)(d - 1) * (5 * NL) + 4 * NL) : F::one();
        F dinv = O::mul(inv, before);
        inv = O::mul(inv, O::mul(q.ZZ, q.ZZZ));
        Affine<F> a = {O::mul(q.X, O::mul(dinv, q.ZZZ)), O::mul(q.Y, O::mul(dinv, q.ZZ))};
        affine_store<F>(tab + ((size_t)w * FIXED_ROW + d) * (2 * NL), a);
    }
}
// pts[i] = scalars[i] * base = sum_w tab[w][byte w of the scalar]: at most 32 mixed additions per point instead of 256 doublings + ~128
// additions.  A lane takes FIXED_CHUNK consecutive points, parks their XYZZ sums in `tmp` and shares ONE inversion among them.
constexpr uint32_t FIXED_CHUNK = 8;
template <class F>
__global__ __launch_bounds__(64) void bases_mul_fixed(uint32_t *__restrict__ pts, const uint32_t *__restrict__ tab, const uint32_t *__restrict__ scalars,
                                                      uint32_t n, uint32_t *__restrict__ tmp) {
    typedef FieldOps<F> O;
    constexpr int NL = O::WORDS;
    const uint32_t lo = (blockIdx.x * blockDim.x + threadIdx.x) * FIXED_CHUNK;
    if (lo >= n) return;
    const uint32_t cnt = n - lo < FIXED_CHUNK ? n - lo : FIXED_CHUNK;
    F pre = F::one();
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t *s = scalars + (size_t)(lo + k) * 8;
        XYZZ<F> acc = XYZZ<F>::infinity();
        for (uint32_t w = 0; w < FIXED_NW; ++w) {
            const uint32_t d = (s[w >> 2] >> ((w & 3) * 8)) & 0xFFu;
            if (d) acc = xyzz_madd(acc, affine_load<F>(tab + ((size_t)w * FIXED_ROW + d) * (2 * NL)));
        }
        uint32_t *slot = tmp + (size_t)(lo + k) * (5 * NL);
        xyzz_store<F>(slot, acc);
        if (!acc.is_inf()) pre = O::mul(pre, O::mul(acc.ZZ, acc.ZZZ));
        O::store(slot + 4 * NL, pre);
    }
    F inv = O::inv(pre);
    for (uint32_t k = cnt; k-- > 0;) {
        const uint32_t *slot = tmp + (size_t)(lo + k) * (5 * NL);
        XYZZ<F> q = xyzz_load<F>(slot);
        if (q.is_inf()) {
            affine_store<F>(pts + (size_t)(lo + k) * (2 * NL), Affine<F>::infinity());
            continue;
        }
        F before = k > 0 ? O::load(tmp + (size_t)(lo + k - 1) * (5 * NL) + 4 * NL) : F::one();
        F dinv = O::mul(inv, before);
        inv = O::mul(inv, O::mul(q.ZZ, q.ZZZ));
        Affine<F> a = {O::mul(q.X, O::mul(dinv, q.ZZZ)), O::mul(q.Y, O::mul(dinv, q.ZZ))};
        affine_store<F>(pts + (size_t)(lo + k) * (2 * NL), a);
    }
}

// Window tables: the table of window w holds 2^off(w) P_i in affine form.  One lane per point: width(w - 1) doublings
// per window in XYZZ, the intermediate points of the windows THIS object keeps parked in `tmp`, one shared inversion
// (Montgomery's trick over the lane's own denominators ZZ*ZZZ), then the affine results are written to their slots.
// Window partition (wrank of wworld): only windows w = wrank + k wworld are kept, in slot k + (wrank != 0); slot 0 holds
// the points themselves (= window 0, rank 0's).  tmp: kept x cnt entries of 5 field elements (X, Y, ZZ, ZZZ, prefix product).
template <class F>
__global__ __launch_bounds__(64) void bases_precompute_range(uint32_t *__restrict__ pts, uint32_t n, uint32_t lo, uint32_t cnt, MsmWindows win,
                                                             uint32_t wrank, uint32_t wworld, uint32_t *__restrict__ tmp) {
    const int W = win.W;
    typedef FieldOps<F> O;
    constexpr int NL = O::WORDS;
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;  // index inside the chunk
    if (j >= cnt) return;
    const uint32_t i = lo + j;
    const uint32_t extra = wrank != 0 ? 1u : 0u;
    auto slot_ptr = [&](uint32_t w) { return pts + ((size_t)(w / wworld + extra) * n + i) * (2 * NL); };  // w mod wworld == wrank
    Affine<F> p = affine_load<F>(pts + (size_t)i * (2 * NL));
    if (p.is_inf()) {
        for (uint32_t w = (wrank != 0 ? wrank : wworld); w < (uint32_t)W; w += wworld) affine_store<F>(slot_ptr(w), Affine<F>::infinity());
        return;
    }
    XYZZ<F> acc = XYZZ<F>::from_affine(p);
    F pre = F::one();
    uint32_t kept = 0;
    for (int w = 1; w < W; ++w) {
        for (int k = 0; k < win.width(w - 1); ++k) acc = xyzz_dbl(acc);  // now 2^off(w) P
        if ((uint32_t)w % wworld != wrank) continue;
        uint32_t *slot = tmp + ((size_t)kept * cnt + j) * (5 * NL);
        xyzz_store<F>(slot, acc);
        pre = O::mul(pre, O::mul(acc.ZZ, acc.ZZZ));
        O::store(slot + 4 * NL, pre);
        ++kept;
    }
    if (kept == 0) return;
    F inv = O::inv(pre);  // 1 / prod (ZZ_w ZZZ_w) over the kept windows
    uint32_t w = ((uint32_t)(W - 1) - wrank) / wworld * wworld + wrank;  // the last kept window (kept > 0: it is >= 1)
    for (uint32_t k = kept; k-- > 0; w -= wworld) {
        const uint32_t *slot = tmp + ((size_t)k * cnt + j) * (5 * NL);
        XYZZ<F> q = xyzz_load<F>(slot);
        F before = k > 0 ? O::load(tmp + ((size_t)(k - 1) * cnt + j) * (5 * NL) + 4 * NL) : F::one();
        F dinv = O::mul(inv, before);            // 1 / (ZZ_w ZZZ_w)
        inv = O::mul(inv, O::mul(q.ZZ, q.ZZZ));  // drop this factor
        Affine<F> a = {O::mul(q.X, O::mul(dinv, q.ZZZ)), O::mul(q.Y, O::mul(dinv, q.ZZ))};
        affine_store<F>(slot_ptr(w), a);
    }
}

template <class F>
__global__ void jac_to_affine_k(const uint32_t *__restrict__ jac, uint32_t *__restrict__ aff, uint8_t *__restrict__ inf) {
    typedef FieldOps<F> O;
    constexpr int CW = O::CANON_WORDS;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Jacobian<F> j = {O::from_canonical(jac), O::from_canonical(jac + CW), O::from_canonical(jac + 2 * CW)};
    XYZZ<F> q = xyzz_from_jacobian(j);
    inf[0] = q.is_inf() ? 1 : 0;
    Affine<F> a = xyzz_to_affine(q);
    O::to_canonical(aff, a.x);
    O::to_canonical(aff + CW, a.y);
}

// out = sum_i jac[i] (canonical Jacobian in and out); one lane, `count` is the number of GPUs
template <class F>
__global__ void jac_sum_k(const uint32_t *__restrict__ jac, uint32_t count, uint32_t *__restrict__ out) {
    typedef FieldOps<F> O;
    constexpr int CW = O::CANON_WORDS;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t *p = jac + (size_t)i * 3 * CW;
        Jacobian<F> j = {O::from_canonical(p), O::from_canonical(p + CW), O::from_canonical(p + 2 * CW)};
        acc = xyzz_add(acc, xyzz_from_jacobian(j));
    }
    Jacobian<F> r = xyzz_to_jacobian(acc);
    O::to_canonical(out, r.X);
    O::to_canonical(out + CW, r.Y);
    O::to_canonical(out + 2 * CW, r.Z);
}

template <class F>
__global__ void msm_write_infinity(uint32_t *out_jac) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    constexpr int CW = FieldOps<F>::CANON_WORDS;
    FieldOps<F>::to_canonical(out_jac, F::one());
    FieldOps<F>::to_canonical(out_jac + CW, F::one());
    FieldOps<F>::to_canonical(out_jac + 2 * CW, F::zero());
}

// one workgroup per MSM of a batch: set sum -> canonical Jacobian at that MSM's output pointer
template <class F, int LPB>
__global__ __launch_bounds__(64) void msm_final_batch(const uint32_t *__restrict__ winsum, uint32_t count, uint32_t *const *__restrict__ outs) {
    constexpr int NL = FieldOps<F>::WORDS;
    constexpr int CW = FieldOps<F>::CANON_WORDS;
    if (blockIdx.x >= count || threadIdx.x >= LPB) return;
    Jacobian<F> j = xyzz_to_jacobian(xyzz_load<F>(winsum + (size_t)blockIdx.x * (4 * NL)));
    uint32_t *out = outs[blockIdx.x];
    FieldOps<F>::to_canonical(out, j.X);
    FieldOps<F>::to_canonical(out + CW, j.Y);
    FieldOps<F>::to_canonical(out + 2 * CW, j.Z);
}

// ---- host side ------------------------------------------------------------------------------------

// Buckets per tail lane (segment length L).  Every operation of the tail is a full addition or doubling on a chain: a
// segment costs 2 L (running sums) + ~38 (the multiple seg L, a 20-bit double-and-add whose additions run on every step of
// a diverged wave) + 8 (LDS tree) dependent operations, each ~13.6 us on one lane, 9.2 us on a lane pair (fu_pair.hpp).
// Measured (tools/msm_profile.py, 2^19 buckets, G1): one lane per bucket L = 8: 0.84 ms, 16: 1.06, 4: 1.38; lane pairs
// L = 16: 0.72, 8: 0.92 (two waves per SIMD: the operations slow down more than the chain shortens).
inline uint32_t msm_tail_segment(const zkhip_ctx *ctx, uint32_t B, size_t sets, int lanes_per_point, bool g2) {
    uint32_t L = 1;
    if (ctx->opt_msm_segment_log >= 0) L = 1u << std::min(ctx->opt_msm_segment_log, 8);
    else {
        const size_t lanes = sets * B * (size_t)lanes_per_point;
        // one wave per SIMD (2^16 lanes) is where an operation is fastest -- the lane-pair group law (fu_pair.hpp: 9.2 us per
        // operation alone on a SIMD, 14.8 us when two waves share it), G2's pair-split Fq2 likewise: L doubles up to 16 to get
        // there, and beyond only when the lanes would otherwise exceed ~4 waves per SIMD (batches of many MSMs)
        (void)g2;
        while (L < 16 && lanes / L > 65536) L <<= 1;
        while (L < 64 && lanes / L > 262144) L <<= 1;
    }
    return std::min(B, L);
}

// ---- the two-level tail on the host: buffers and launches --------------------------------------------------------------------
// Applies to merged bucket sets with window tables (ONE set per MSM) of at least 2^opt_msm_tail_fold buckets whose rows and columns fit a workgroup.
// G2 (lane pairs throughout): a lone 2^20-point MSM measures the same either way (tail 1.94 against 1.97 ms), but the two levels are HALF the
// additions, and a Groth16 proof runs its G2 MSM under the G1 MSMs, which take the issue slots (50.7-51.0 -> 51.9-52.4 M constraints/s, A/B on
// one box, three times interleaved): on by default, with its own switch msm_tail_fold_g2.
template <class F>
bool msm_fold_applies(const zkhip_ctx *ctx, uint32_t B, bool tables) {
    if (FieldOps<F>::WORDS > 16 && !ctx->opt_msm_tail_fold_g2) return false;
    if (!tables || ctx->opt_msm_tail_fold <= 0 || B < (1u << std::min(ctx->opt_msm_tail_fold, 30))) return false;
    const MsmFold g = msm_fold_geom(ctx, B);
    const uint32_t sa = MSM_TAIL_THREADS / BucketLane<F>::LANES;
    return g.m_col >= 1 && g.m_row <= sa && g.m_col <= sa && B >= sa * g.run && g.m_row * g.run == g.C && g.m_col * g.run == g.R;
}
struct MsmFoldBuffers {
    MsmFold g;
    size_t level2_words, segsum_words, winsum_words;
    uint32_t *level2 = nullptr, *segsum = nullptr, *winsum = nullptr;
    size_t need() const { return zkhip_ctx::ws_round(level2_words * 4) + zkhip_ctx::ws_round(segsum_words * 4) + zkhip_ctx::ws_round(winsum_words * 4); }
    void take(zkhip_ctx *ctx) {
        level2 = ctx->ws_take<uint32_t>(level2_words);
        segsum = ctx->ws_take<uint32_t>(segsum_words);
        winsum = ctx->ws_take<uint32_t>(winsum_words);
    }
};
template <class F>
MsmFoldBuffers msm_fold_buffers(const zkhip_ctx *ctx, uint32_t B, size_t nsets) {
    constexpr size_t PW = (size_t)4 * FieldOps<F>::WORDS;  // words per XYZZ point
    MsmFoldBuffers fb;
    fb.g = msm_fold_geom(ctx, B);
    fb.level2_words = nsets * 2 * fb.g.C * PW;
    fb.segsum_words = nsets * 2 * ((fb.g.C + 63) / 64) * PW;  // segments of >= 1 bucket, >= 64 of them per workgroup
    fb.winsum_words = nsets * 2 * PW;
    return fb;
}

// the old tail over the 2 nsets level-2 sets with lane type TX, then the outputs
template <class TX, int XL, bool G2>
int msm_fold_level2(zkhip_ctx *ctx, const MsmFoldBuffers &fb, size_t nsets, uint32_t *const *d_outs, uint32_t *d_out) {
    constexpr int NL = FieldOps<TX>::WORDS;
    const size_t sets2 = 2 * nsets;
    const uint32_t C = fb.g.C, L2 = msm_tail_segment(ctx, C, sets2, XL, G2), slots = MSM_TAIL_THREADS / XL;
    const uint32_t nseg = C / L2, nblk = (nseg + slots - 1) / slots;
    // (a variant of msm_bucket_red compiled for one-bucket segments only spills 37 instead of 173 registers and measures the same: 0.156 ms)
    ZK_MAX_LDS(ctx, (msm_bucket_red<TX, XL>), (size_t)slots * 4 * NL * 4);
    ZK_LAUNCH(ctx, "msm_bucket_red", (msm_bucket_red<TX, XL>), dim3((unsigned)(sets2 * nblk)), dim3(MSM_TAIL_THREADS), (size_t)slots * 4 * NL * 4, fb.level2, C, L2,
              nseg, nblk, fb.segsum);
    const uint32_t wt = msm_window_threads(nblk, XL);
    ZK_LAUNCH(ctx, "msm_window_sum", (msm_window_sum<TX, XL>), dim3((unsigned)sets2), dim3(wt), (size_t)wt / XL * 4 * NL * 4, fb.segsum, nblk, fb.winsum);
    ZK_LAUNCH(ctx, "msm_final", (msm_final_fold<TX, XL>), dim3((unsigned)nsets), dim3(64), 0, fb.winsum, (uint32_t)nsets, fb.g.log_c, d_outs, d_out);
    return 0;
}

// sets[nsets][B] merged buckets -> nsets canonical Jacobian results (d_outs: device array of output pointers, or the one output d_out).
// TL / TLPB: the lane shape of the level-2 tail when the quads do not apply.
template <class F, class TL, int TLPB>
int msm_fold_tail(zkhip_ctx *ctx, const uint32_t *sets, size_t nsets, const MsmFoldBuffers &fb, uint32_t *const *d_outs, uint32_t *d_out) {
    constexpr int NL = FieldOps<F>::WORDS;
    constexpr bool G2 = FieldOps<F>::WORDS > 16;
    typedef typename BucketLane<F>::type FL;
    constexpr int LPB = BucketLane<F>::LANES;
    typedef typename QuadLane<F>::type TQ;  // the trees are latency: quads where the field has them (else this IS the tail lane)
    constexpr int QL = QuadLane<F>::LANES;
    const MsmFold &g = fb.g;
    ZK_HIP_CHECK(ctx, hipMemsetAsync(fb.level2, 0, fb.level2_words * 4, ctx->stream));  // empty buckets (the second set's upper part stays so)
    const size_t lds = (size_t)MSM_TAIL_THREADS / LPB * 4 * NL * 4;
    ZK_MAX_LDS(ctx, (msm_fold<FL, LPB, TQ, QL>), lds);
    ZK_LAUNCH(ctx, "msm_fold", (msm_fold<FL, LPB, TQ, QL>), dim3((unsigned)(nsets * (g.B / (MSM_TAIL_THREADS / LPB * g.run)) * 2)), dim3(MSM_TAIL_THREADS), lds,
              sets, g, fb.level2);
    if (QuadLane<F>::AVAILABLE && ctx->opt_msm_tail_quads && 2 * nsets * g.C <= ((size_t)1 << 18))
        return msm_fold_level2<TQ, QL, G2>(ctx, fb, nsets, d_outs, d_out);
    return msm_fold_level2<TL, TLPB, G2>(ctx, fb, nsets, d_outs, d_out);
}

// the tail workgroups keep 256 XYZZ points in LDS (56 KiB for G1, 128 KiB for BLS12-381 G2)
template <class F>
int msm_tail_attr(zkhip_ctx *ctx) {
    typedef typename TailLane<F>::type TL;
    constexpr int TLPB = TailLane<F>::LANES;
    ZK_MAX_LDS(ctx, (msm_bucket_red<TL, TLPB>), MSM_TAIL_THREADS / TLPB * 4 * FieldOps<F>::WORDS * 4);
    return 0;
}

// The geometry of one MSM call: window size, windows, sets, buckets.
struct MsmPlan {
    int c, W_all, W;      // window bits; windows of the scalar; windows this call handles (window partition: a subset)
    uint32_t B, S, nb;    // buckets per set, sets, S * B
    MsmWindows win;
    uint32_t wrank, wworld;
    bool tables;
};

inline MsmPlan msm_plan(const zkhip_ctx *ctx, const zkhip_bases *bases, size_t n) {
    MsmPlan p;
    p.tables = bases->tables();
    int c = p.tables ? bases->c_tab : ctx->opt_msm_window_bits;
    if (c <= 0) c = zk_msm_auto_window(n);
    p.c = std::max(2, std::min(ZK_MSM_MAX_C, c));
    if (!p.tables) p.c = std::min(p.c, 16);  // one bucket set PER WINDOW there (W 2^(c-1) buckets): stay within the sort's key range
    p.W_all = msm_windows(zk_scalar_bits(bases->curve), p.c);  // scalars are folded to |s| <= (r - 1) / 2: no carry out of the top window
    p.win = msm_make_windows(zk_scalar_bits(bases->curve), p.W_all);
    // window partition over GPUs: this call handles the windows {w : w mod win_world == win_rank} only (all equal-weight
    // thanks to the tables)
    p.wrank = p.tables ? (uint32_t)bases->win_rank : 0u;
    p.wworld = p.tables ? (uint32_t)bases->win_world : 1u;
    p.W = p.tables ? bases->local_windows() : p.W_all;
    p.B = 1u << (p.c - 1);
    // sets: without tables the windows keep their own buckets (different weights: Horner in msm_final); with tables all
    // windows share ONE set unless that leaves too few buckets (= lanes) to fill the chip
    if (!p.tables) p.S = (uint32_t)p.W;
    else {
        int s = ctx->opt_msm_sets;
        if (s <= 0) s = (int)((zk_msm_target_lanes() + p.B - 1) / p.B);
        p.S = (uint32_t)std::max(1, std::min(p.W, s));
        while (p.S > 1 && (uint64_t)p.S * p.B > ((uint64_t)SORT_MAX_SUPER << SORT_MAX_LOW)) --p.S;
    }
    p.nb = p.S * p.B;
    return p;
}

// the sort's launch sequence for one tile shape
template <class SZ>
int msm_sort_run(zkhip_ctx *ctx, const SortGeom &g, int W, uint32_t nb, uint32_t nbh, uint32_t nblk, const uint32_t *dig, uint32_t *bh, uint32_t *bo,
                 uint32_t *bsums, uint32_t *tmp_idx, uint16_t *tmp_key, uint32_t *offs, uint32_t *idx) {
    const size_t lds_hist = (size_t)g.nsuper * 4;
    const size_t lds_split = ((size_t)3 * g.nsuper + 1 + SZ::THREADS + 2 * SZ::TILE) * 4;
    const size_t lds_final = (size_t)SZ::TILE * 6;
    if (lds_split > 48 * 1024) ZK_MAX_LDS(ctx, msm_sort_split<SZ>, 160 * 1024 - 256);
    if (lds_final > 40 * 1024) ZK_MAX_LDS(ctx, msm_sort_final<SZ>, 160 * 1024 - (4 * SORT_NLOW_MAX + SZ::THREADS) * 4 - 256);
    ZK_LAUNCH(ctx, "msm_sort_hist", msm_sort_hist<SZ>, dim3(g.ntile, W), dim3(SZ::THREADS), lds_hist, dig, g, bh);
    (void)nblk;
    ZK_TRY(msm_scan(ctx, "msm_scan", bh, nbh, bo, bsums));
    ZK_LAUNCH(ctx, "msm_sort_split", msm_sort_split<SZ>, dim3(g.ntile, W), dim3(SZ::THREADS), lds_split, dig, g, bo, tmp_idx, tmp_key);
    ZK_LAUNCH(ctx, "msm_sort_final", msm_sort_final<SZ>, dim3(g.nsuper), dim3(SZ::THREADS), lds_final, tmp_idx, tmp_key, g, nb, bo, offs, idx);
    return 0;
}

template <class F>
int msm_run_t(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint32_t *d_scalars, uint32_t *d_out_jac,
              uint32_t *batch_slot = nullptr, size_t *need_out = nullptr, bool reuse_sort = false) {
    constexpr int NL = FieldOps<F>::WORDS;
    typedef typename BucketLane<F>::type FL;           // what a lane holds in the bucket kernels
    constexpr int LPB = BucketLane<F>::LANES;          // lanes per point (2 for G2: fu2_pair.hpp)
    typedef typename TailLane<F>::type TL;             // ... and in the latency-bound tail (G1: group law over a lane pair, fu_pair.hpp)
    constexpr int TLPB = TailLane<F>::LANES;
    constexpr bool G2 = FieldOps<F>::WORDS > 16;
    const MsmPlan P = msm_plan(ctx, bases, n);
    const int W = P.W;
    const uint32_t B = P.B, S = P.S, nb = P.nb;
    // every sort offset / prefix sum / idx position is a u32 over the W * n entries, and an entry addresses a table row
    if ((uint64_t)W * n >= (1ull << 32) || (uint64_t)bases->nslots * bases->n >= (1ull << 31)) {
        ctx->last_error = "MSM of " + std::to_string(n) + " points x " + std::to_string(W) + " windows exceeds the 32-bit entry index (split the range)";
        return ZKHIP_ERR_RANGE;
    }
    const int Sr = P.tables ? 1 : W;  // sets left after the equal-weight merge
    const uint32_t L = msm_tail_segment(ctx, B, Sr, TLPB, G2);  // buckets per tail segment
    const uint32_t tail_slots = MSM_TAIL_THREADS / TLPB;  // points per tail workgroup
    const uint32_t nseg = B / L, nblk_tail = (nseg + tail_slots - 1) / tail_slots;
    // the quad variant of the tail (below) may cut the set into more workgroups: the partial-sum buffer holds either
    const uint32_t Lq_cap = msm_tail_segment(ctx, B, Sr, QuadLane<F>::LANES, G2);
    const uint32_t nblk_cap = std::max(nblk_tail, (B / Lq_cap + MSM_TAIL_THREADS / QuadLane<F>::LANES - 1) / (MSM_TAIL_THREADS / QuadLane<F>::LANES));
    // two-level LDS counting sort (see msm_sort_*): the low bits inside a super-bucket, the rest across super-buckets
    SortGeom g;
    g.n = (uint32_t)n;
    g.W = (uint32_t)W;
    g.B = B;
    g.S = S;
    g.lowb = (uint32_t)std::min<int>(SORT_MAX_LOW, P.c - 1);
    g.nsuper = (nb + (1u << g.lowb) - 1) >> g.lowb;
    const bool big_tiles = ctx->opt_msm_sort_tile_log >= 14;
    g.tile = big_tiles ? SortBig::TILE : SortSmall::TILE;
    g.ntile = (uint32_t)((n + g.tile - 1) / g.tile);
    g.tables = P.tables ? 1u : 0u;
    g.slot0 = P.tables ? (uint32_t)bases->slot_of_local(0) : 0u;
    g.bases_n = (uint32_t)bases->n;
    g.base_off = (uint32_t)offset;
    if (g.nsuper > SORT_MAX_SUPER) return ZKHIP_ERR_RANGE;
    const size_t ncol = (size_t)W * g.ntile;
    const size_t nbh64 = (size_t)g.nsuper * ncol;
    if (nbh64 >= (1ull << 31)) return ZKHIP_ERR_RANGE;
    const uint32_t nbh = (uint32_t)nbh64, nblk = (nbh + 1023) / 1024;

    size_t need = 0;
    need += zkhip_ctx::ws_round((size_t)W * n * 4);        // dig
    need += zkhip_ctx::ws_round((size_t)nbh * 4);          // per-tile super-bucket histogram
    need += zkhip_ctx::ws_round(((size_t)nbh + 1) * 4);    // its exclusive scan
    need += zkhip_ctx::ws_round((size_t)nblk * 4);         // block sums
    need += zkhip_ctx::ws_round((size_t)W * n * 4);        // tmp_idx
    need += zkhip_ctx::ws_round((size_t)W * n * 2);        // tmp_key
    need += zkhip_ctx::ws_round(((size_t)nb + 1) * 4);     // offs
    need += zkhip_ctx::ws_round((size_t)W * n * 4);        // idx
    need += zkhip_ctx::ws_round((size_t)nb * 4 * NL * 4);  // buckets
    const uint32_t sblk = (nb + 1023) / 1024, nsh = SIZE_BINS * sblk, sblk2 = (nsh + 1023) / 1024;
    need += 2 * zkhip_ctx::ws_round(((size_t)nsh + 1) * 4) + zkhip_ctx::ws_round((size_t)sblk2 * 4) + zkhip_ctx::ws_round((size_t)nb * 4);  // size sort
    need += zkhip_ctx::ws_round((size_t)Sr * nblk_cap * 4 * NL * 4);
    need += zkhip_ctx::ws_round((size_t)std::max(1, Sr) * 4 * NL * 4);
    const bool fold = !batch_slot && msm_fold_applies<F>(ctx, B, P.tables);  // batches fold once, for all members (msm_batch_tail)
    MsmFoldBuffers fb;
    if (fold) {
        fb = msm_fold_buffers<F>(ctx, B, 1);
        need += fb.need();
    }
    // worst-case plan of the large-bucket path: every entry in a large bucket
    const size_t entries = (size_t)W * n;
    const uint32_t large_thresh = (uint32_t)std::max<size_t>(MSM_LARGE_BUCKET, 4 * ((entries + nb - 1) / nb));
    const uint32_t large_cap = (uint32_t)(entries / large_thresh + 1);
    const uint32_t task_cap = (uint32_t)(entries / MSM_LARGE_CHUNK + large_cap + 1);
    need += zkhip_ctx::ws_round(16) + zkhip_ctx::ws_round((size_t)task_cap * 12) + zkhip_ctx::ws_round((size_t)large_cap * 12);
    need += zkhip_ctx::ws_round((size_t)task_cap * 4 * NL * 4);
    if (need_out) {  // dry run: workspace size only
        *need_out = need;
        return 0;
    }
    ZK_TRY(ctx->ws_reserve(ctx->ws_floor + need));
    ctx->ws_reset();
    uint32_t *dig = ctx->ws_take<uint32_t>((size_t)W * n);
    uint32_t *bh = ctx->ws_take<uint32_t>(nbh);
    uint32_t *bo = ctx->ws_take<uint32_t>((size_t)nbh + 1);
    uint32_t *bsums = ctx->ws_take<uint32_t>(nblk);
    uint32_t *tmp_idx = ctx->ws_take<uint32_t>((size_t)W * n);
    uint16_t *tmp_key = ctx->ws_take<uint16_t>((size_t)W * n);
    uint32_t *offs = ctx->ws_take<uint32_t>((size_t)nb + 1);
    uint32_t *idx = ctx->ws_take<uint32_t>((size_t)W * n);
    uint32_t *buckets = ctx->ws_take<uint32_t>((size_t)nb * 4 * NL);
    if (batch_slot && S == 1) buckets = batch_slot;  // a batch member with ONE set accumulates straight into its slot of the batch (no copy of 2^19 buckets = 117 MB)
    uint32_t *sh = ctx->ws_take<uint32_t>((size_t)nsh + 1);
    uint32_t *so = ctx->ws_take<uint32_t>((size_t)nsh + 1);
    uint32_t *ssums = ctx->ws_take<uint32_t>(sblk2);
    uint32_t *order = ctx->ws_take<uint32_t>(nb);
    uint32_t *segsum = ctx->ws_take<uint32_t>((size_t)Sr * nblk_cap * 4 * NL);
    uint32_t *winsum = ctx->ws_take<uint32_t>((size_t)std::max(1, Sr) * 4 * NL);
    if (fold) fb.take(ctx);
    uint32_t *plan = ctx->ws_take<uint32_t>(4);
    uint32_t *tasks = ctx->ws_take<uint32_t>((size_t)task_cap * 3);
    uint32_t *large = ctx->ws_take<uint32_t>((size_t)large_cap * 3);
    uint32_t *partials = ctx->ws_take<uint32_t>((size_t)task_cap * 4 * NL);

    const uint32_t *d_b = bases->d;  // entries address table rows from the start of the bases object

    // reuse_sort (batches: the previous member had the SAME scalars over an entry-compatible bases object -- msm_same_entries): the
    // sorted entries, bucket offsets, size order and large-bucket plan in the workspace are this member's too (same sizes, so the same
    // addresses); only the gathers read another table
    if (!reuse_sort) {
        unsigned gn = (unsigned)((n + 255) / 256);
        if (bases->curve == CURVE_BLS12_381)
            ZK_LAUNCH(ctx, "msm_digits", msm_digits_only<BlsFr>, dim3(gn), dim3(256), 0, d_scalars, (uint32_t)n, P.win, P.wrank, P.wworld, dig);
        else ZK_LAUNCH(ctx, "msm_digits", msm_digits_only<BnFr>, dim3(gn), dim3(256), 0, d_scalars, (uint32_t)n, P.win, P.wrank, P.wworld, dig);
        ZK_TRY((big_tiles ? msm_sort_run<SortBig> : msm_sort_run<SortSmall>)(ctx, g, W, nb, nbh, nblk, dig, bh, bo, bsums, tmp_idx, tmp_key, offs, idx));
        // buckets by descending size
        ZK_LAUNCH(ctx, "msm_size_sort", msm_size_hist, dim3(sblk), dim3(256), 0, offs, nb, sblk, large_thresh, sh);
        ZK_TRY(msm_scan(ctx, "msm_size_sort", sh, nsh, so, ssums));
        ZK_LAUNCH(ctx, "msm_size_sort", msm_size_scatter, dim3(sblk), dim3(256), 0, offs, nb, sblk, large_thresh, so, order);
    }
    if constexpr (FieldOps<F>::WORDS <= 16) {
        // G1: accumulator coordinates in LDS, three waves per SIMD
        constexpr int NT = MSM_G1_THREADS;
        size_t lds_acc = LdsAcc<F, NT>::BYTES;
        ZK_MAX_LDS(ctx, (msm_bucket_acc_lds<F, NT, 3>), lds_acc);
        ZK_LAUNCH(ctx, "msm_bucket_acc", (msm_bucket_acc_lds<F, NT, 3>), dim3((nb + NT - 1) / NT), dim3(NT), lds_acc, d_b, offs, idx, nb, large_thresh,
                  order, buckets);
    } else {
        // G2: every bucket is an even / odd lane pair, each lane holding one component of the Fq2 coordinates
        // (fu2_pair.hpp): a lane then carries what a G1 lane carries -- two waves per SIMD instead of one.
        constexpr int NT = MSM_G2_THREADS, WAVES = MSM_G2_WAVES;
        size_t lds_acc = LdsAcc<FL, NT>::BYTES;
        ZK_MAX_LDS(ctx, (msm_bucket_acc_lds<FL, NT, WAVES, LPB>), lds_acc);
        ZK_LAUNCH(ctx, "msm_bucket_acc", (msm_bucket_acc_lds<FL, NT, WAVES, LPB>), dim3((unsigned)(((size_t)nb * LPB + NT - 1) / NT)), dim3(NT), lds_acc, d_b,
                  offs, idx, nb, large_thresh, order, buckets);
    }
    // large buckets: plan on the device (no host round trip), then fixed-size grids that read the plan
    if (!reuse_sort) {
        ZK_HIP_CHECK(ctx, hipMemsetAsync(plan, 0, 16, ctx->stream));
        ZK_LAUNCH(ctx, "msm_plan_large", msm_plan_large, dim3((nb + 255) / 256), dim3(256), 0, offs, nb, plan, tasks, large, task_cap, large_cap, large_thresh,
                  ctx->d_status);
    }
    {
        size_t lds_large = (size_t)128 / LPB * 4 * NL * 4;
        if (lds_large > 48 * 1024) ZK_MAX_LDS(ctx, (msm_bucket_large<FL, LPB>), lds_large);
        unsigned grid_large = (unsigned)std::min<size_t>(task_cap, 512);  // persistent: workgroups loop over the task list
        ZK_LAUNCH(ctx, "msm_bucket_large", (msm_bucket_large<FL, LPB>), dim3(grid_large), dim3(128), lds_large, d_b, idx, plan, tasks, partials);
        ZK_LAUNCH(ctx, "msm_bucket_large", (msm_large_combine<FL, LPB>), dim3((unsigned)std::min<uint32_t>(large_cap, 256)), dim3(64),
                  (size_t)64 / LPB * 4 * NL * 4, plan, large, partials, buckets);
    }
    if (P.tables) {
        for (uint32_t cur = S; cur > 1;) {
            const uint32_t q = (cur + 3) / 4;
            ZK_LAUNCH(ctx, "msm_bucket_merge", (msm_bucket_merge<FL, LPB>), dim3((unsigned)(((size_t)q * B * LPB + 255) / 256)), dim3(256), 0, buckets, B, q,
                      cur);
            cur = q;
        }
    }
    if (batch_slot) {  // batched call: hand the merged buckets over, the reduction runs once for the whole batch
        if (buckets != batch_slot) ZK_HIP_CHECK(ctx, hipMemcpyAsync(batch_slot, buckets, (size_t)B * 4 * NL * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    if (fold) return msm_fold_tail<F, TL, TLPB>(ctx, buckets, 1, fb, nullptr, d_out_jac);
    // Small bucket sets leave lanes to spare even as pairs: the group law then runs over lane QUADS (fu_quad.hpp: 4 product steps per
    // addition instead of 7) with segments twice as long -- from 2^18 buckets down; at 2^19 the longer segments eat the gain.
    if (QuadLane<F>::AVAILABLE && ctx->opt_msm_tail_quads && (size_t)Sr * B <= ((size_t)1 << 18)) {
        typedef typename QuadLane<F>::type TQ;
        constexpr int QL = QuadLane<F>::LANES;
        const uint32_t Lq = msm_tail_segment(ctx, B, Sr, QL, G2), slots_q = MSM_TAIL_THREADS / QL;
        const uint32_t nseg_q = B / Lq, nblk_q = (nseg_q + slots_q - 1) / slots_q;  // <= nblk_cap: the partial-sum buffer fits
        ZK_MAX_LDS(ctx, (msm_bucket_red<TQ, QL>), MSM_TAIL_THREADS / QL * 4 * NL * 4);
        ZK_LAUNCH(ctx, "msm_bucket_red", (msm_bucket_red<TQ, QL>), dim3((unsigned)Sr * nblk_q), dim3(MSM_TAIL_THREADS), (size_t)slots_q * 4 * NL * 4, buckets, B,
                  Lq, nseg_q, nblk_q, segsum);
        const uint32_t wt = msm_window_threads(nblk_q, QL);
        ZK_LAUNCH(ctx, "msm_window_sum", (msm_window_sum<TQ, QL>), dim3(Sr), dim3(wt), (size_t)wt / QL * 4 * NL * 4, segsum, nblk_q, winsum);
        ZK_LAUNCH(ctx, "msm_final", (msm_final<TQ, QL>), dim3(1), dim3(64), 0, winsum, Sr, P.win, d_out_jac);
        return 0;
    }
    ZK_TRY(msm_tail_attr<F>(ctx));
    ZK_LAUNCH(ctx, "msm_bucket_red", (msm_bucket_red<TL, TLPB>), dim3((unsigned)Sr * nblk_tail), dim3(MSM_TAIL_THREADS), (size_t)tail_slots * 4 * NL * 4, buckets, B,
              L, nseg, nblk_tail, segsum);
    const uint32_t wthreads = msm_window_threads(nblk_tail, TLPB);
    ZK_LAUNCH(ctx, "msm_window_sum", (msm_window_sum<TL, TLPB>), dim3(Sr), dim3(wthreads), (size_t)wthreads / TLPB * 4 * NL * 4, segsum, nblk_tail, winsum);
    ZK_LAUNCH(ctx, "msm_final", (msm_final<TL, TLPB>), dim3(1), dim3(64), 0, winsum, Sr, P.win, d_out_jac);
    return 0;
}

// Build the window tables of a bases object whose slot 0 (the points) is filled (called once at upload).
template <class F>
int bases_precompute_t(zkhip_ctx *ctx, zkhip_bases *b) {
    constexpr int NL = FieldOps<F>::WORDS;
    if (!b->tables() || b->n == 0) return 0;
    const size_t chunk = 1u << 18;  // bounds the temporary to kept * 2^18 * 5 field elements
    const int kept = b->local_windows() - (b->win_rank == 0 ? 1 : 0);  // window 0 is the points themselves
    if (kept <= 0) return 0;
    size_t per = (size_t)kept * 5 * NL * 4;
    ZK_TRY(ctx->ws_reserve(per * std::min(chunk, b->n) + 4096));
    for (size_t lo = 0; lo < b->n; lo += chunk) {
        size_t cnt = std::min(chunk, b->n - lo);
        ctx->ws_reset();
        uint32_t *tmp = ctx->ws_take<uint32_t>(per / 4 * cnt);
        ZK_LAUNCH(ctx, "bases_precompute", bases_precompute_range<F>, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, b->d, (uint32_t)b->n,
                  (uint32_t)lo, (uint32_t)cnt, msm_make_windows(zk_scalar_bits(b->curve), b->ntab), (uint32_t)b->win_rank, (uint32_t)b->win_world, tmp);
    }
    return 0;
}

// Do two members of a batch produce the same sorted entries?  An entry is (bucket of the digit, table row of the point): the same
// scalars, range and table geometry give the same list whatever the points are -- the queries of one Groth16 proof that run over the
// same assignment vector (A, the dense B.h, the padded L) then share one digit extraction, sort, size order and large-bucket plan.
inline bool msm_same_entries(const zkhip_bases *a, size_t off_a, size_t n_a, const uint32_t *s_a, const zkhip_bases *b, size_t off_b, size_t n_b,
                             const uint32_t *s_b) {
    return s_a == s_b && n_a == n_b && off_a == off_b && a->curve == b->curve && a->n == b->n && a->c_tab == b->c_tab && a->ntab == b->ntab &&
           a->win_rank == b->win_rank && a->win_world == b->win_world && a->nslots == b->nslots && a->tables() && b->tables();
}

// Several MSMs over table-backed bases of one group with one window size: per MSM digits / sort / accumulate /
// merge as usual, then ONE bucket reduction, set sum and output conversion for the whole batch.
// TL / TLPB: what a lane of the shared tail holds.  A few members: the latency-bound shape of a single MSM (G1: the group law over
// lane pairs).  Many members (50 KZG columns x 2^19 buckets) have lanes to spare and are bound by WORK: one lane per point then
// does the same additions without the pair's selects and DPP swaps (measured: msm_bucket_red of a 50-column 2^20-row commit
// 16.2 -> 13.7 ms).
template <class F, class TL, int TLPB>
int msm_batch_tail(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                   const uint32_t *const *d_scalars, uint32_t *const *d_outs) {
    constexpr int NL = FieldOps<F>::WORDS;
    const int c = bases[0]->c_tab;
    const uint32_t B = 1u << (c - 1);
    const uint32_t L = msm_tail_segment(ctx, B, count, TLPB, FieldOps<F>::WORDS > 16);
    const uint32_t tail_slots = MSM_TAIL_THREADS / TLPB;  // points per tail workgroup
    const uint32_t nseg = B / L, nblk_tail = (nseg + tail_slots - 1) / tail_slots;
    size_t max_need = 0;
    for (size_t i = 0; i < count; ++i) {
        size_t need = 0;
        if (ns[i] == 0 || bases[i]->local_windows() == 0) continue;
        ZK_TRY(msm_run_t<F>(ctx, bases[i], offsets[i], ns[i], nullptr, nullptr, nullptr, &need));
        max_need = std::max(max_need, need);
    }
    const size_t slot_words = (size_t)B * 4 * NL;
    size_t fixed = zkhip_ctx::ws_round(count * slot_words * 4) + zkhip_ctx::ws_round(count * nblk_tail * 4 * NL * 4) +
                   zkhip_ctx::ws_round(count * 4 * NL * 4) + zkhip_ctx::ws_round(count * sizeof(void *));
    const bool fold = msm_fold_applies<F>(ctx, B, true);
    MsmFoldBuffers fb;
    if (fold) {
        fb = msm_fold_buffers<F>(ctx, B, count);
        fixed += fb.need();
    }
    ctx->ws_floor = 0;
    ZK_TRY(ctx->ws_reserve(fixed + max_need));
    ctx->ws_reset();
    uint32_t *slots = ctx->ws_take<uint32_t>(count * slot_words);
    uint32_t *segsum = ctx->ws_take<uint32_t>(count * nblk_tail * 4 * NL);
    uint32_t *winsum = ctx->ws_take<uint32_t>(count * 4 * NL);
    uint32_t **d_ptrs = ctx->ws_take<uint32_t *>(count);
    if (fold) fb.take(ctx);
    ctx->ws_floor = ctx->ws_off;  // the per-MSM stages bump-allocate above the batch area
    int rc = 0;
    size_t prev = (size_t)-1;  // the member whose sort the workspace holds
    for (size_t i = 0; i < count && rc == 0; ++i) {
        if (ns[i] == 0 || bases[i]->local_windows() == 0) {
            hipError_t e = hipMemsetAsync(slots + i * slot_words, 0, slot_words * 4, ctx->stream);  // all buckets at infinity
            if (e != hipSuccess) rc = ZKHIP_ERR_HIP;
        } else {
            const bool reuse = ctx->opt_msm_share_sort && prev != (size_t)-1 &&
                               msm_same_entries(bases[prev], offsets[prev], ns[prev], d_scalars[prev], bases[i], offsets[i], ns[i], d_scalars[i]);
            rc = msm_run_t<F>(ctx, bases[i], offsets[i], ns[i], d_scalars[i], nullptr, slots + i * slot_words, nullptr, reuse);
            prev = i;
        }
    }
    ctx->ws_floor = 0;
    if (rc) return rc;
    if (ctx->batch_dptrs_override) {  // graph capture: the output pointers already sit in a device array owned by the graph
        d_ptrs = static_cast<uint32_t **>(ctx->batch_dptrs_override);
    } else {
        ctx->batch_ptrs.assign(d_outs, d_outs + count);
        ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_ptrs, ctx->batch_ptrs.data(), count * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    }
    if (fold) return msm_fold_tail<F, TL, TLPB>(ctx, slots, count, fb, d_ptrs, nullptr);
    ZK_MAX_LDS(ctx, (msm_bucket_red<TL, TLPB>), MSM_TAIL_THREADS / TLPB * 4 * NL * 4);
    ZK_LAUNCH(ctx, "msm_bucket_red", (msm_bucket_red<TL, TLPB>), dim3((unsigned)count * nblk_tail), dim3(MSM_TAIL_THREADS), (size_t)tail_slots * 4 * NL * 4, slots,
              B, L, nseg, nblk_tail, segsum);
    const uint32_t wthreads = msm_window_threads(nblk_tail, TLPB);
    ZK_LAUNCH(ctx, "msm_window_sum", (msm_window_sum<TL, TLPB>), dim3((unsigned)count), dim3(wthreads), (size_t)wthreads / TLPB * 4 * NL * 4, segsum, nblk_tail, winsum);
    ZK_LAUNCH(ctx, "msm_final", (msm_final_batch<TL, TLPB>), dim3((unsigned)count), dim3(64), 0, winsum, (uint32_t)count, d_ptrs);
    return 0;
}

template <class F>
int msm_batch_t(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                const uint32_t *const *d_scalars, uint32_t *const *d_outs) {
    typedef typename BucketLane<F>::type FL;
    typedef typename TailLane<F>::type TL;
    const size_t buckets = count << (bases[0]->c_tab - 1);
    // >= 2^21 buckets in all (the four G1 multiexps of a proof: msm_bucket_red 2.16 -> 1.87 ms; 50 KZG columns: 16.2 -> 13.7):
    // the one-lane shape keeps every SIMD busy by itself -- work-bound
    if (!std::is_same<FL, TL>::value && buckets >= ((size_t)1 << 21))
        return msm_batch_tail<F, FL, BucketLane<F>::LANES>(ctx, count, bases, offsets, ns, d_scalars, d_outs);
    return msm_batch_tail<F, TL, TailLane<F>::LANES>(ctx, count, bases, offsets, ns, d_scalars, d_outs);
}

// ---- the per-(curve, group) operation table ----------------------------------------------------------------------
template <class F>
int op_run(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint32_t *d_scalars, uint32_t *d_out_jac) {
    return msm_run_t<F>(ctx, bases, offset, n, d_scalars, d_out_jac);
}
template <class F>
int op_to_mont(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_canonical, const uint8_t *d_inf) {
    ZK_LAUNCH(ctx, "bases_to_mont", bases_to_mont<F>, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, d_canonical, d_inf, (uint32_t)b->n, b->d);
    return 0;
}
template <class F>
int op_from_mont(zkhip_ctx *ctx, const zkhip_bases *b, size_t offset, size_t n, uint32_t *d_out, uint8_t *d_inf) {
    const uint32_t *src = b->d + offset * b->stride_u32;
    ZK_LAUNCH(ctx, "bases_from_mont", bases_from_mont<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, src, (uint32_t)n, d_out, d_inf);
    return 0;
}
template <class F>
int op_mul(zkhip_ctx *ctx, zkhip_bases *b, const uint32_t *d_base_canonical, const uint32_t *d_scalars) {
    constexpr int NL = FieldOps<F>::WORDS;
    if (b->n < 4096) {  // a handful of points: the table would cost more than it saves
        ZK_LAUNCH(ctx, "bases_mul", bases_mul<F>, dim3((unsigned)((b->n + 63) / 64)), dim3(64), 0, b->d, d_base_canonical, d_scalars, (uint32_t)b->n);
        return 0;
    }
    // fixed-base windows: a 32 x 256-entry affine table of the base, then <= 32 mixed additions per point (round 4: a 2^20-constraint
    // key's batch exponentiations were 256 doublings + ~128 additions + one Fermat inversion PER POINT)
    const size_t tab_words = (size_t)FIXED_NW * FIXED_ROW * 2 * NL, tmp_words = std::max<size_t>((size_t)FIXED_NW * FIXED_ROW, b->n) * 5 * NL;
    ZK_TRY(ctx->ws_reserve(zkhip_ctx::ws_round(tab_words * 4) + zkhip_ctx::ws_round(tmp_words * 4)));
    ctx->ws_reset();
    uint32_t *tab = ctx->ws_take<uint32_t>(tab_words), *tmp = ctx->ws_take<uint32_t>(tmp_words);
    ZK_LAUNCH(ctx, "bases_mul", bases_fixed_table<F>, dim3(1), dim3(64), 0, d_base_canonical, tab, tmp);
    const size_t lanes = (b->n + FIXED_CHUNK - 1) / FIXED_CHUNK;
    ZK_LAUNCH(ctx, "bases_mul", bases_mul_fixed<F>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, b->d, tab, d_scalars, (uint32_t)b->n, tmp);
    return 0;
}
template <class F>
int op_jac_to_affine(zkhip_ctx *ctx, const uint32_t *d_jac, uint32_t *d_aff, uint8_t *d_inf) {
    ZK_LAUNCH(ctx, "jac_to_affine", jac_to_affine_k<F>, dim3(1), dim3(64), 0, d_jac, d_aff, d_inf);
    return 0;
}
template <class F>
int op_jac_sum(zkhip_ctx *ctx, const uint32_t *d_pts, size_t count, uint32_t *d_out) {
    ZK_LAUNCH(ctx, "jac_sum", jac_sum_k<F>, dim3(1), dim3(64), 0, d_pts, (uint32_t)count, d_out);
    return 0;
}
template <class F>
int op_write_infinity(zkhip_ctx *ctx, uint32_t *d_out_jac) {
    ZK_LAUNCH(ctx, "msm_write_infinity", msm_write_infinity<F>, dim3(1), dim3(64), 0, d_out_jac);
    return 0;
}

template <class F>
const MsmOps *msm_make_ops() {
    static const MsmOps ops = {op_run<F>,        msm_batch_t<F>,      bases_precompute_t<F>, op_to_mont<F>,        op_from_mont<F>, op_mul<F>,
                               op_jac_to_affine<F>, op_jac_sum<F>, op_write_infinity<F>, (size_t)2 * FieldOps<F>::WORDS};
    return &ops;
}

}  // namespace
