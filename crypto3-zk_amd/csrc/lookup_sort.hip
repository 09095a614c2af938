// placeholder's sort_polynomials (zk/snark/systems/plonk/placeholder/lookup_argument.hpp:565-638) on the device.
//
// The reference: an unordered_map counts every value of the reduced table columns (usable rows, columns one after the other) and every
// looked-up input; then ONE serial walk over the table values emits, whenever the value changes, the value that just ended -- a single
// zero if it was zero (the walk starts from a virtual zero), `count` copies otherwise -- and after the walk the last value's copies
// (unless it is zero); the emitted sequence is dealt over |input| + |value| vectors of usable_rows entries (zero behind), and entry
// usable_rows of every vector but the last repeats the head of the next.
//
// Here, over v = the concatenated table values (Lv = k_val * usable_rows of them, canonical Fr in HBM; equality of limbs is equality
// in the field):
//   1. run starts   start(t) = t == 0 || v[t] != v[t - 1]; an exclusive scan numbers the runs; st[r] = first position of run r
//   2. hash insert  every run head goes into an open-addressing table of u32 slots (slot = run index, keys compared through v[st[slot]]):
//                   the first run of a value is its canonical run c(r); cnt[c(r)] += length(r)           (atomicCAS / atomicAdd)
//   3. inputs       every input element finds its value's canonical run and adds 1 to its count; a value that is in no table raises
//                   ZK_STATUS_LOOKUP_NOT_IN_TABLE (the reference asserts; without assertions it counts a key it never emits -- same output)
//   4. sizes        size(r) = value(r) == 0 ? (r is the last run ? 0 : 1) : cnt[c(r)]   (a value whose runs are NOT adjacent is emitted
//                   cnt times per run, as the reference's walk does); + 1 leading zero when the first value is non-zero; exclusive scan
//   5. emit         one lane per output entry: binary search of its emission index in the run offsets -> the run's value; the stitch
//                   and the zero tails in the same pass.  More emitted entries than the vectors hold (the reference writes past its
//                   vectors there) raises ZK_STATUS_LOOKUP_SORT_OVERFLOW and the excess is dropped.
// No arithmetic: 32-byte compares, u32 scans, atomics.  Everything on the context's stream; no host round trip but the final
// synchronisation that keeps the staged pointer table alive.
#include <algorithm>

#include "ctx.hpp"

namespace {

constexpr uint32_t LS_THREADS = 256, LS_PER = 4, LS_TILE = LS_THREADS * LS_PER;  // scan: 1024 entries per workgroup
constexpr uint32_t LS_EMPTY = 0xffffffffu;

struct Key {
    uint4 a, b;
};
__device__ __forceinline__ bool key_eq(const Key &x, const Key &y) {
    return x.a.x == y.a.x && x.a.y == y.a.y && x.a.z == y.a.z && x.a.w == y.a.w && x.b.x == y.b.x && x.b.y == y.b.y && x.b.z == y.b.z && x.b.w == y.b.w;
}
__device__ __forceinline__ bool key_zero(const Key &x) { return (x.a.x | x.a.y | x.a.z | x.a.w | x.b.x | x.b.y | x.b.z | x.b.w) == 0; }
__device__ __forceinline__ Key key_load(const uint32_t *col, size_t row) {
    const uint4 *q = reinterpret_cast<const uint4 *>(col) + 2 * row;
    return Key{q[0], q[1]};
}
// element t of the concatenation of `cols` (u rows each)
__device__ __forceinline__ Key key_at(const uint32_t *const *cols, uint32_t u, uint32_t t) { return key_load(cols[t / u], t % u); }
__device__ __forceinline__ uint32_t key_hash(const Key &k) {
    // the table values are either small integers or theta-compressed (uniform) field elements: fold all eight words, then a 64-bit finaliser
    uint64_t h = ((uint64_t)k.a.y << 32 | k.a.x) ^ (((uint64_t)k.a.w << 32 | k.a.z) * 0x9e3779b97f4a7c15ull);
    h ^= (((uint64_t)k.b.y << 32 | k.b.x) * 0xc2b2ae3d27d4eb4full) ^ (((uint64_t)k.b.w << 32 | k.b.z) * 0x165667b19e3779f9ull);
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return (uint32_t)h;
}

// ---- exclusive u32 scan, three kernels: tile-local scan + tile totals, scan of the totals by one workgroup, add-back
// Sums SATURATE at `cap` (a table whose equal values are not adjacent emits count copies per run -- 2^39 entries for 2^20 rows
// alternating two values -- and a wrapped u32 total could pass for a small one: the overflow flag would stay down and the offsets would
// stop being monotone).  Saturating addition is associative and monotone, so the scans stay scans; every prefix below `cap` is exact.
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b, uint32_t cap) {
    const uint64_t s = (uint64_t)a + b;
    return s > cap ? cap : (uint32_t)s;
}
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t *lds, uint32_t &total, uint32_t cap = 0xffffffffu) {  // x: this lane's value; returns its exclusive prefix
    const uint32_t t = threadIdx.x;
    lds[t] = x > cap ? cap : x;
    __syncthreads();
    for (uint32_t d = 1; d < LS_THREADS; d <<= 1) {
        const uint32_t o = t >= d ? lds[t - d] : 0;
        __syncthreads();
        lds[t] = sat_add(lds[t], o, cap);
        __syncthreads();
    }
    total = lds[LS_THREADS - 1];
    const uint32_t excl = t ? lds[t - 1] : 0;
    __syncthreads();
    return excl;
}

// start flags of the table values, scanned per tile: run_idx[t] = runs started before t inside the tile (fixed up by ls_scan_add)
__global__ __launch_bounds__(LS_THREADS) void ls_starts(const uint32_t *const *__restrict__ vals, uint32_t u, uint32_t lv, uint32_t *__restrict__ run_idx,
                                                        uint8_t *__restrict__ start, uint32_t *__restrict__ tile_tot) {
    __shared__ uint32_t lds[LS_THREADS];
    const uint32_t lo = blockIdx.x * LS_TILE + threadIdx.x * LS_PER;
    uint32_t f[LS_PER], sum = 0;
    Key prev;
    if (lo > 0 && lo < lv) prev = key_at(vals, u, lo - 1);
#pragma unroll
    for (uint32_t i = 0; i < LS_PER; ++i) {
        const uint32_t t = lo + i;
        f[i] = 0;
        if (t < lv) {
            const Key k = key_at(vals, u, t);
            f[i] = t == 0 || !key_eq(k, prev);
            prev = k;
        }
        sum += f[i];
    }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, lds, total);
#pragma unroll
    for (uint32_t i = 0; i < LS_PER; ++i) {
        const uint32_t t = lo + i;
        if (t < lv) {
            run_idx[t] = run;
            start[t] = (uint8_t)f[i];
            run += f[i];
        }
    }
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = total;
}

// exclusive scan of `count` tile totals in place by ONE workgroup (a serial loop over tiles of LS_TILE); the grand total goes to *total
__global__ __launch_bounds__(LS_THREADS) void ls_scan_top(uint32_t *__restrict__ v, uint32_t count, uint32_t *__restrict__ total, uint32_t cap) {
    __shared__ uint32_t lds[LS_THREADS];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < count; base += LS_TILE) {
        const uint32_t lo = base + threadIdx.x * LS_PER;
        uint32_t x[LS_PER], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < LS_PER; ++i) {
            x[i] = lo + i < count ? v[lo + i] : 0;
            sum = sat_add(sum, x[i], cap);
        }
        uint32_t tot;
        uint32_t run = sat_add(carry, block_excl_scan(sum, lds, tot, cap), cap);
#pragma unroll
        for (uint32_t i = 0; i < LS_PER; ++i) {
            if (lo + i < count) v[lo + i] = run;
            run = sat_add(run, x[i], cap);
        }
        carry = sat_add(carry, tot, cap);
    }
    if (threadIdx.x == 0) *total = carry;
}

// run_idx += its tile's prefix; the run heads scatter their position: st[run] = t
__global__ __launch_bounds__(LS_THREADS) void ls_heads(uint32_t *__restrict__ run_idx, const uint8_t *__restrict__ start, const uint32_t *__restrict__ tile_pre,
                                                       uint32_t lv, uint32_t *__restrict__ st) {
    const uint32_t t = blockIdx.x * LS_THREADS + threadIdx.x;
    if (t >= lv) return;
    const uint32_t r = run_idx[t] + tile_pre[t / LS_TILE];
    run_idx[t] = r;
    if (start[t]) st[r] = t;
}

// every run head into the hash table; canon[r] = the run that owns the value's slot; cnt[canon[r]] += length(r)
__global__ __launch_bounds__(LS_THREADS) void ls_insert(const uint32_t *const *__restrict__ vals, uint32_t u, uint32_t lv, const uint32_t *__restrict__ st,
                                                        const uint32_t *__restrict__ runs_p, uint32_t *__restrict__ slots, uint32_t mask,
                                                        uint32_t *__restrict__ canon, uint32_t *__restrict__ cnt) {
    const uint32_t r = blockIdx.x * LS_THREADS + threadIdx.x, runs = *runs_p;
    if (r >= runs) return;
    const uint32_t head = st[r], len = (r + 1 < runs ? st[r + 1] : lv) - head;
    const Key k = key_at(vals, u, head);
    uint32_t h = key_hash(k) & mask, owner;
    for (;;) {
        const uint32_t old = atomicCAS(&slots[h], LS_EMPTY, r);
        if (old == LS_EMPTY) {
            owner = r;
            break;
        }
        if (key_eq(key_at(vals, u, st[old]), k)) {
            owner = old;
            break;
        }
        h = (h + 1) & mask;
    }
    canon[r] = owner;
    atomicAdd(&cnt[owner], len);
}

// every input element adds one to its value's count
__global__ __launch_bounds__(LS_THREADS) void ls_count_inputs(const uint32_t *const *__restrict__ ins, const uint32_t *const *__restrict__ vals, uint32_t u, uint32_t li,
                                                              const uint32_t *__restrict__ st, const uint32_t *__restrict__ slots, uint32_t mask,
                                                              uint32_t *__restrict__ cnt, uint32_t *__restrict__ status) {
    const uint32_t t = blockIdx.x * LS_THREADS + threadIdx.x;
    if (t >= li) return;
    const Key k = key_at(ins, u, t);
    uint32_t h = key_hash(k) & mask;
    for (;;) {
        const uint32_t s = slots[h];
        if (s == LS_EMPTY) {
            atomicOr(status, ZK_STATUS_LOOKUP_NOT_IN_TABLE);
            return;
        }
        if (key_eq(key_at(vals, u, st[s]), k)) {
            if (!key_zero(k)) atomicAdd(&cnt[s], 1u);  // the count of zero is never emitted (one zero per zero run): no traffic for the unselected rows
            return;
        }
        h = (h + 1) & mask;
    }
}

// size of run r's emission, scanned per tile (slot 0 of the sequence is the leading zero of a walk that starts on a non-zero value)
__global__ __launch_bounds__(LS_THREADS) void ls_sizes(const uint32_t *const *__restrict__ vals, uint32_t u, const uint32_t *__restrict__ st,
                                                       const uint32_t *__restrict__ runs_p, const uint32_t *__restrict__ canon, const uint32_t *__restrict__ cnt,
                                                       uint32_t *__restrict__ off, uint32_t *__restrict__ tile_tot, uint32_t cap) {
    __shared__ uint32_t lds[LS_THREADS];
    const uint32_t runs = *runs_p, lo = blockIdx.x * LS_TILE + threadIdx.x * LS_PER;
    uint32_t s[LS_PER], sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < LS_PER; ++i) {
        const uint32_t r = lo + i;
        s[i] = 0;
        if (r < runs) {
            const bool zero = key_zero(key_at(vals, u, st[r]));
            s[i] = zero ? (r + 1 < runs ? 1u : 0u) : cnt[canon[r]];
            if (r == 0 && !zero) s[i] = sat_add(s[i], 1, cap);  // the leading zero rides on run 0 (ls_emit tells them apart)
        }
        sum = sat_add(sum, s[i], cap);
    }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, lds, total, cap);
#pragma unroll
    for (uint32_t i = 0; i < LS_PER; ++i) {
        if (lo + i < runs) off[lo + i] = run;
        run = sat_add(run, s[i], cap);
    }
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(LS_THREADS) void ls_offsets(uint32_t *__restrict__ off, const uint32_t *__restrict__ tile_pre, const uint32_t *__restrict__ runs_p,
                                                         const uint32_t *__restrict__ total_p, uint32_t capacity, uint32_t *__restrict__ status) {
    const uint32_t r = blockIdx.x * LS_THREADS + threadIdx.x, runs = *runs_p;
    if (r < runs) off[r] = sat_add(off[r], tile_pre[r / LS_TILE], capacity + 1);
    if (r == 0) {
        off[runs] = *total_p;  // sentinel
        if (*total_p > capacity) atomicOr(status, ZK_STATUS_LOOKUP_SORT_OVERFLOW);  // the sums saturate at capacity + 1: no wrapped total slips under
    }
}

// one lane per entry of the K output vectors (n entries each): rows < u take the emitted sequence, row u of every vector but the last the
// head of the next, everything else is zero
__global__ __launch_bounds__(LS_THREADS) void ls_emit(const uint32_t *const *__restrict__ vals, uint32_t u, const uint32_t *__restrict__ st,
                                                      const uint32_t *__restrict__ off, const uint32_t *__restrict__ runs_p, uint32_t *const *__restrict__ outs,
                                                      uint32_t n, uint32_t kk) {
    const size_t g = (size_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (g >= (size_t)kk * n) return;
    const uint32_t vec = (uint32_t)(g / n), row = (uint32_t)(g % n), runs = *runs_p;
    Key out{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    uint64_t e = ~0ull;
    if (row < u)
        e = (uint64_t)vec * u + row;
    else if (row == u && vec + 1 < kk)
        e = (uint64_t)(vec + 1) * u;
    if (runs && e < off[runs]) {
        // the run whose emission holds entry e: the last r with off[r] <= e
        uint32_t lo = 0, hi = runs;  // invariant: off[lo] <= e < off[hi]
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (off[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        const Key k = key_at(vals, u, st[lo]);
        // run 0 of a walk that starts on a non-zero value carries the leading zero in front of its copies
        if (!(lo == 0 && e == 0 && !key_zero(k))) out = k;
    }
    uint4 *q = reinterpret_cast<uint4 *>(outs[vec]) + 2 * (size_t)row;
    q[0] = out.a;
    q[1] = out.b;
}

}  // namespace

extern "C" int zkhip_lookup_sort_dev(zkhip_ctx *ctx, size_t k_in, const void *const *d_input, size_t k_val, const void *const *d_value, size_t n, size_t usable_rows,
                                     void *const *d_sorted) {
    if (!ctx || (k_in && !d_input) || (k_val && !d_value) || !d_sorted) return ZKHIP_ERR_INVALID;
    const size_t kk = k_in + k_val;
    if (kk == 0) return ZKHIP_OK;
    if (k_in >= 4096 || k_val >= 4096 || n >= ((size_t)1 << 31)) return ZKHIP_ERR_RANGE;
    if (n && usable_rows >= n) return ZKHIP_ERR_RANGE;
    if (kk * usable_rows >= ((size_t)1 << 31) || kk * n >= ((size_t)1 << 38)) return ZKHIP_ERR_RANGE;  // positions and emission indices are u32
    if (k_val * usable_rows >= ((size_t)1 << 30)) return ZKHIP_ERR_RANGE;                                // the hash table (2 slots per table entry, a power of two) is indexed by u32
    for (size_t i = 0; n && i < k_in; ++i)
        if (!d_input[i]) return ZKHIP_ERR_INVALID;
    for (size_t i = 0; n && i < k_val; ++i)
        if (!d_value[i]) return ZKHIP_ERR_INVALID;
    for (size_t i = 0; n && i < kk; ++i)
        if (!d_sorted[i]) return ZKHIP_ERR_INVALID;
    if (n == 0) return ZKHIP_OK;
    ZK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t u = (uint32_t)usable_rows, lv = (uint32_t)(k_val * usable_rows), li = (uint32_t)(k_in * usable_rows);
    const uint32_t vt = (lv + LS_TILE - 1) / LS_TILE;  // tiles over the values = upper bound of the tiles over the runs
    uint32_t table = 1024;
    while (table < 2 * (size_t)lv) table <<= 1;  // load factor <= 1/2 whatever the number of runs
    size_t need = zkhip_ctx::ws_round(2 * kk * sizeof(void *)) + zkhip_ctx::ws_round((size_t)lv * 4) * 4 + zkhip_ctx::ws_round((size_t)lv + 4) +
                  zkhip_ctx::ws_round(((size_t)lv + 1) * 4) + zkhip_ctx::ws_round(((size_t)vt + 1) * 4) * 2 + zkhip_ctx::ws_round((size_t)table * 4) + zkhip_ctx::ws_round(64);
    ZK_TRY(ctx->ws_reserve(need));
    ctx->ws_reset();
    const uint32_t **d_ptrs = ctx->ws_take<const uint32_t *>(2 * kk);  // inputs | values | outputs
    uint32_t *d_run = ctx->ws_take<uint32_t>(lv), *d_st = ctx->ws_take<uint32_t>(lv), *d_canon = ctx->ws_take<uint32_t>(lv), *d_cnt = ctx->ws_take<uint32_t>(lv);
    uint8_t *d_start = ctx->ws_take<uint8_t>((size_t)lv + 4);
    uint32_t *d_off = ctx->ws_take<uint32_t>((size_t)lv + 1);
    uint32_t *d_tile_a = ctx->ws_take<uint32_t>((size_t)vt + 1), *d_tile_b = ctx->ws_take<uint32_t>((size_t)vt + 1);
    uint32_t *d_slots = ctx->ws_take<uint32_t>(table);
    uint32_t *d_scal = ctx->ws_take<uint32_t>(16);  // [0] runs, [1] emitted entries
    ctx->batch_ptrs.clear();
    for (size_t i = 0; i < k_in; ++i) ctx->batch_ptrs.push_back((uint32_t *)d_input[i]);
    for (size_t i = 0; i < k_val; ++i) ctx->batch_ptrs.push_back((uint32_t *)d_value[i]);
    for (size_t i = 0; i < kk; ++i) ctx->batch_ptrs.push_back((uint32_t *)d_sorted[i]);
    ZK_HIP_CHECK(ctx, hipMemcpyAsync(d_ptrs, ctx->batch_ptrs.data(), 2 * kk * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    const uint32_t *const *p_in = d_ptrs, *const *p_val = d_ptrs + k_in;
    uint32_t *const *p_out = (uint32_t *const *)(d_ptrs + kk);
    ZK_HIP_CHECK(ctx, hipMemsetAsync(d_scal, 0, 64, ctx->stream));
    if (lv) {
        ZK_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, (size_t)lv * 4, ctx->stream));
        ZK_HIP_CHECK(ctx, hipMemsetAsync(d_slots, 0xff, (size_t)table * 4, ctx->stream));
        ZK_LAUNCH(ctx, "lookup_sort", ls_starts, dim3(vt), dim3(LS_THREADS), 0, p_val, u, lv, d_run, d_start, d_tile_a);
        ZK_LAUNCH(ctx, "lookup_sort", ls_scan_top, dim3(1), dim3(LS_THREADS), 0, d_tile_a, vt, d_scal, 0xffffffffu);
        const dim3 per_value((lv + LS_THREADS - 1) / LS_THREADS);
        ZK_LAUNCH(ctx, "lookup_sort", ls_heads, per_value, dim3(LS_THREADS), 0, d_run, d_start, d_tile_a, lv, d_st);
        // the run count stays on the device: the per-run kernels are launched over its upper bound lv and read it
        ZK_LAUNCH(ctx, "lookup_sort", ls_insert, per_value, dim3(LS_THREADS), 0, p_val, u, lv, d_st, d_scal, d_slots, table - 1, d_canon, d_cnt);
        if (li)
            ZK_LAUNCH(ctx, "lookup_sort", ls_count_inputs, dim3((li + LS_THREADS - 1) / LS_THREADS), dim3(LS_THREADS), 0, p_in, p_val, u, li, d_st, d_slots, table - 1, d_cnt,
                      ctx->d_status);
        const uint32_t cap1 = (uint32_t)(kk * usable_rows) + 1;  // emission sizes and offsets saturate one above what the vectors hold
        ZK_LAUNCH(ctx, "lookup_sort", ls_sizes, dim3(vt), dim3(LS_THREADS), 0, p_val, u, d_st, d_scal, d_canon, d_cnt, d_off, d_tile_b, cap1);
        ZK_LAUNCH(ctx, "lookup_sort", ls_scan_top, dim3(1), dim3(LS_THREADS), 0, d_tile_b, vt, d_scal + 1, cap1);
        ZK_LAUNCH(ctx, "lookup_sort", ls_offsets, per_value, dim3(LS_THREADS), 0, d_off, d_tile_b, d_scal, d_scal + 1, (uint32_t)(kk * usable_rows), ctx->d_status);
    } else if (li) {
        // inputs without a table: nothing they could be found in
        ZK_HIP_CHECK(ctx, hipMemsetAsync(d_slots, 0xff, (size_t)table * 4, ctx->stream));
        ZK_LAUNCH(ctx, "lookup_sort", ls_count_inputs, dim3((li + LS_THREADS - 1) / LS_THREADS), dim3(LS_THREADS), 0, p_in, p_val, u, li, d_st, d_slots, table - 1, d_cnt,
                  ctx->d_status);
    }
    const size_t total = kk * n;
    ZK_LAUNCH(ctx, "lookup_sort", ls_emit, dim3((unsigned)((total + LS_THREADS - 1) / LS_THREADS)), dim3(LS_THREADS), 0, p_val, u, d_st, d_off, d_scal, p_out, (uint32_t)n,
              (uint32_t)kk);
    ZK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the staged pointer table may be reused after return
    return ZKHIP_OK;
}
