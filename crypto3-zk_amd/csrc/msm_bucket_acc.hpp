// Bucket accumulation with the running sum's coordinates in LDS (included by msm.hip).
#pragma once
#include "curve.hpp"
#include "fu2_pair.hpp"

namespace zkhip {

// ---- bucket accumulation with the accumulator in LDS ----------------------------------------------------------
// A wave issues one VALU instruction per ~4 cycles, so throughput scales with waves per SIMD (measured linear up to
// 4, tools/mulbench).  Holding X, Y, ZZ of the running sum in LDS (conflict-free [coord][quad][lane] uint4 planes)
// instead of VGPRs brings the G1 kernel under 168 registers (three waves per SIMD instead of two); the G2 kernel
// runs on lane pairs (fu2_pair.hpp) that each hold half of every Fq2 coordinate, at two waves per SIMD.  ZZZ stays
// in registers; the mixed addition reads the LDS coordinates where it uses them and writes the results back.
template <class F>
struct LimbView;  // the 29-bit limbs of a coordinate as one flat sequence
template <class U>
struct LimbView<Fu<U>> {
    static constexpr int N = U::L;
    ZK_D static uint32_t get(const Fu<U> &x, int i) { return x.v[i]; }
    ZK_D static void set(Fu<U> &x, int i, uint32_t v) { x.v[i] = v; }
};
template <class U>
struct LimbView<Fu2h<U>> {
    static constexpr int N = U::L;
    ZK_D static uint32_t get(const Fu2h<U> &x, int i) { return x.v.v[i]; }
    ZK_D static void set(Fu2h<U> &x, int i, uint32_t v) { x.v.v[i] = v; }
};

template <class F, int NT>
struct LdsAcc {
    typedef LimbView<F> V;
    static constexpr int Q = (V::N + 3) / 4;  // uint4 per coordinate
    static constexpr size_t BYTES = (size_t)3 * Q * NT * 16;
    uint4 *base;  // [3][Q][NT]
    uint32_t t;
    ZK_D F get(int coord) const {
        uint32_t w[4 * Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            uint4 v = base[(coord * Q + q) * NT + t];
            w[4 * q] = v.x, w[4 * q + 1] = v.y, w[4 * q + 2] = v.z, w[4 * q + 3] = v.w;
        }
        F r;
#pragma unroll
        for (int i = 0; i < V::N; ++i) V::set(r, i, w[i]);
        return r;
    }
    ZK_D void put(int coord, const F &x) const {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            uint4 v;
            v.x = 4 * q + 0 < V::N ? V::get(x, 4 * q + 0) : 0u;
            v.y = 4 * q + 1 < V::N ? V::get(x, 4 * q + 1) : 0u;
            v.z = 4 * q + 2 < V::N ? V::get(x, 4 * q + 2) : 0u;
            v.w = 4 * q + 3 < V::N ? V::get(x, 4 * q + 3) : 0u;
            base[(coord * Q + q) * NT + t] = v;
        }
    }
};

// F is the type a LANE holds: the coordinate field itself (LPB = 1 lane per bucket), or one half of an Fq2
// coordinate (Fu2h, LPB = 2: an even / odd lane pair per bucket, fu2_pair.hpp).
template <class F, int NT, int WAVES, int LPB = 1>
__global__ __launch_bounds__(NT, WAVES) void msm_bucket_acc_lds(const uint32_t *__restrict__ bases, const uint32_t *__restrict__ offs,
                                                           const uint32_t *__restrict__ idx, uint32_t nbuckets, uint32_t large,
                                                           const uint32_t *__restrict__ order, uint32_t *__restrict__ buckets) {
    typedef FieldOps<F> O;
    constexpr int NL = O::WORDS;
    extern __shared__ __attribute__((aligned(16))) uint4 acc_lds[];
    const uint32_t tid = threadIdx.x, slot = (blockIdx.x * NT + tid) / LPB;
    if (slot >= nbuckets) return;
    const uint32_t g = order[slot];  // buckets by descending size (msm_size_*)
    const uint32_t lo = offs[g], hi = offs[g + 1];
    if (hi - lo > large) return;
    LdsAcc<F, NT> A = {acc_lds, tid};
    enum { CX = 0, CY = 1, CZZ = 2 };
    F ZZZ = F::zero();
    bool inf = true;
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t e = idx[k];
        Affine<F> p = affine_load<F>(bases + (size_t)(e & 0x7FFFFFFFu) * (2 * NL));  // entry = table row (slot * n + point) | sign
        if (p.is_inf()) continue;
        if (e >> 31) p.y = O::template sub<O::K1>(F::zero(), p.y);
        if (inf) {
            A.put(CX, p.x);
            A.put(CY, p.y);
            A.put(CZZ, F::one());
            ZZZ = F::one();
            inf = false;
            continue;
        }
        // xyzz_madd (curve.hpp) with X, Y, ZZ fetched from LDS at their points of use
        F Pd = O::template sub<O::K2>(O::mul(p.x, A.get(CZZ)), A.get(CX));
        F R = O::template sub<O::K2>(O::mul(p.y, ZZZ), A.get(CY));
        F PP = O::sqr(Pd);
        if (O::is_zero_product(PP)) {  // same x: doubling or cancellation (rare)
            if (O::is_zero(R)) {
                XYZZ<F> d = xyzz_dbl_affine(p);
                A.put(CX, d.X);
                A.put(CY, d.Y);
                A.put(CZZ, d.ZZ);
                ZZZ = d.ZZZ;
            } else {
                inf = true;
            }
            continue;
        }
        F PPP = O::mul(Pd, PP);
        F Q = O::mul(A.get(CX), PP);
        F X3 = O::template sub<O::K1>(O::sqr(R), O::add(PPP, O::add(Q, Q)));
        F Y3 = O::template mul_sub<O::K2>(R, O::template sub<O::K2>(Q, X3), A.get(CY), PPP);
        A.put(CX, X3);
        A.put(CY, Y3);
        A.put(CZZ, O::mul(A.get(CZZ), PP));
        ZZZ = O::mul(ZZZ, PPP);
    }
    XYZZ<F> out = XYZZ<F>::infinity();
    if (!inf) out = {A.get(CX), A.get(CY), A.get(CZZ), ZZZ};
    xyzz_store<F>(buckets + (size_t)g * (4 * NL), out);
}

}  // namespace zkhip
