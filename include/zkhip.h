/* zkhip -- C ABI of the MI355X-native MSM / NTT proving backend for NilFoundation/crypto3-zk.
 *
 * The reference is a header-only C++ template library with no FFI; its hot path leaves the repository at
 * the calls listed below (all into crypto3-algebra / crypto3-math).  Every entry point here states the
 * reference interface it replaces; INTEGRATION.md shows the C++ shim a crypto3-zk maintainer adds on top.
 *
 * Conventions
 *   - Plain C, caller owns every host buffer, nothing throws; return 0 on success, negative zkhip_status
 *     otherwise (the reference asserts, compiled out in release: prover.hpp:77,88-89; kzg.hpp:145,413).
 *   - Field elements cross the boundary as CANONICAL (non-Montgomery) little-endian 64-bit limbs:
 *       Fr (both curves) 4 limbs; BLS12-381 Fq 6 limbs; BN254 Fq 4 limbs; Fq2 = c0 limbs then c1 limbs.
 *     G1 affine = x | y; G2 affine = x.c0 | x.c1 | y.c0 | y.c1.  Infinity travels as a separate flag.
 *     The reference's in-memory representation lives in crypto3-multiprecision (not observable from the
 *     zk tree), so the shim converts through canonical integers and never memcpy's reference objects.
 *   - Group results are returned as Jacobian (X, Y, Z) with x = X/Z^2, y = Y/Z^3 and Z = 0 for infinity,
 *     the 3-coordinate shape of the reference's G::value_type; compare in affine.
 *   - omega and the coset generator are ARGUMENTS (arithmetic_params<F> lives in crypto3-algebra).
 *   - One context per GPU per process (one process per GPU); a context is not thread-safe.  A caller that wants SEVERAL GPUs behind
 *     one call -- the shape of the reference, whose parallelism sits inside process() / commit() -- takes a device group
 *     (zkhip_group_init below): one context per GPU, one host thread, the exchange inside the library.
 *   - The library has no CPU fallback: without a usable HIP device every call fails with
 *     ZKHIP_ERR_NO_DEVICE.
 */
#ifndef ZKHIP_H
#define ZKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkhip_ctx zkhip_ctx;
typedef struct zkhip_bases zkhip_bases;

enum zkhip_curve { ZKHIP_BLS12_381 = 0, ZKHIP_BN254 = 1 };
enum zkhip_group { ZKHIP_G1 = 1, ZKHIP_G2 = 2 };
enum zkhip_status {
    ZKHIP_OK = 0,
    ZKHIP_ERR_NO_DEVICE = -1,
    ZKHIP_ERR_INVALID = -2,
    ZKHIP_ERR_HIP = -3,
    ZKHIP_ERR_OOM = -4,
    ZKHIP_ERR_RANGE = -5
};

/* ---- context -------------------------------------------------------------------------------- */
int zkhip_init(int device_id, zkhip_ctx **out);
void zkhip_destroy(zkhip_ctx *ctx);
const char *zkhip_strerror(int status);
const char *zkhip_last_error(const zkhip_ctx *ctx); /* HIP error text of the last ZKHIP_ERR_HIP */
/* All work is enqueued on this stream (default: a stream the context creates). */
int zkhip_set_stream(zkhip_ctx *ctx, void *hip_stream);
/* Two contexts on one GPU are two in-order streams (each with its own workspace).  zkhip_stream_wait(waiter, signal):
 * everything enqueued on `waiter` after this call runs after everything enqueued on `signal` before it (an event;
 * the host does not block).  The Groth16 shim runs the G2 multiexp of the B query on a second context this way, so
 * that its latency-bound reduction tail overlaps the G1 multiexps (prover.hpp:116-139 are independent). */
int zkhip_stream_wait(zkhip_ctx *waiter, zkhip_ctx *signal);
int zkhip_device(const zkhip_ctx *ctx);
int zkhip_sync(zkhip_ctx *ctx);
/* Sticky error flags raised by kernels since the last call (bit 0: a gather index out of range, bit 1: the MSM's
 * large-bucket plan overflowed its capacity, bit 2: zkhip_lookup_sort_dev met a looked-up value that is in no table, bit 3: its
 * emitted sequence does not fit the sorted vectors).  Synchronises the stream, clears the flags; returns ZKHIP_ERR_RANGE if
 * any was set (the results computed meanwhile are then not to be used).  The shims call it once per proof. */
int zkhip_device_status(zkhip_ctx *ctx, uint32_t *flags /* nullable */);
/* Tunables: "msm_window_bits" (0 = auto), "msm_sets" (bucket sets with window tables, 0 = auto), "msm_segment_log" (tail: 2^k buckets
 * per lane, < 0 = auto), "msm_tail_quads" (1: the tail of bucket sets up to 2^18 runs the group law over lane quads, 0: pairs everywhere), "msm_tail_fold" (k > 0: the bucket reduction of table-backed sets of >= 2^k buckets is two-level -- row and column sums of the bucket index, then the old tail over 2 sets of ~sqrt(B) buckets; 0: off; default 16), "msm_fold_run" (buckets a lane of that kernel sums before the workgroup trees, 0 = auto), "msm_tail_fold_g2" (1: G2 sets too), "msm_share_sort" (1: consecutive members of zkhip_msm_batch_dev with the same scalars pointer, range and
 * table geometry share one digit extraction / sort / large-bucket plan; 0: every member sorts), "msm_sort_tile_log" (14: the MSM has the GPU to itself; 12: kernels of another context run alongside and
 * the sort's LDS tiles must fit next to them), "ntt_radix_log", "ntt_tile_log", "poly_coset_extend" (1: zkhip_poly_resize_dev grows n -> K n, K <= 16, by n-point transforms over the K - 1 new cosets; 0: one K n-point transform), "ec_ntt_table_lanes" (zkhip_ec_ntt_dev: points multiplied per launch = per-lane window tables held at
 * once; 0 = as many as fit 1 GiB), "msm_precompute" / "msm_precompute_min" (window
 * tables at upload), "msm_graphs" (HIP-graph replay of repeated MSM calls; off), "msm_shard_rank" / "msm_shard_world" (window
 * partition over GPUs: bases uploaded from now on hold the tables of windows {w : w mod world == rank} only, and an MSM over them
 * yields the partial sum of those windows). */
int zkhip_set_option(zkhip_ctx *ctx, const char *name, int64_t value);
/* "stream_priority" (value < 0: highest, > 0: lowest, 0: default) recreates the context's own stream with that scheduling priority:
 * of two contexts on one GPU the higher one's workgroups are dispatched first (set it before any work is enqueued). */
int zkhip_get_option(const zkhip_ctx *ctx, const char *name, int64_t *value);

/* ---- device memory (plumbing for callers that keep vectors resident) --------------------------
 * zkhip_free synchronises the device the block lives on and keeps the block for the next zkhip_malloc of its size class (option
 * "alloc_cache_mb": megabytes kept per context, default 16384, 0 = hand every block back at once); the cache is emptied at
 * zkhip_destroy and when the driver runs out of memory.
 * OWNERSHIP: a block belongs to the context it was allocated through, whichever context it is freed through (the free settles it with
 * its owner).  zkhip_destroy returns the CACHED blocks to the driver and leaves the blocks the caller still holds alone: they lose
 * their owner, stay valid, and a later zkhip_free through any live context hands them to hipFree.  A caller that wants its memory
 * back must therefore zkhip_free every block BEFORE (or after) destroying the context -- destroy does not reclaim live blocks
 * (the shim's shared_ptr handles may outlive their context and must not dangle). */
int zkhip_malloc(zkhip_ctx *ctx, size_t bytes, void **dptr);
int zkhip_free(zkhip_ctx *ctx, void *dptr);
int zkhip_memcpy_h2d(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes);
int zkhip_memcpy_d2h(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes);
/* enqueue only (no synchronisation): for pinned host buffers, ordered with the kernels on the context's stream */
int zkhip_memcpy_h2d_async(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes);
/* the way back, enqueue only: `dst` must be page-locked (zkhip_host_alloc) and is valid after the next zkhip_sync.  The LPC shim
 * streams a precommitment's leaves to the caller's hash in slices this way: slice k + 1 is in flight while slice k is absorbed. */
int zkhip_memcpy_d2h_async(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes);
/* page-locked host memory: H2D / D2H at link speed instead of through a staging copy */
/* device-to-device copy in stream order (no synchronisation) */
int zkhip_memcpy_d2d_async(zkhip_ctx *ctx, void *dst, const void *src, size_t bytes);
/* the strided form, same device, stream order: `rows` rows of `width` bytes, row r from src + r * src_pitch to dst + r * dst_pitch
 * (width <= both pitches; the two regions must not overlap).  What the LPC shim over a device group packs a member's share of the
 * evaluations with: the 2^fri_step segments of every polynomial that one leaf owner's leaves are made of (hip/lpc.hpp). */
int zkhip_memcpy_2d_d2d_async(zkhip_ctx *ctx, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width, size_t rows);
int zkhip_host_alloc(zkhip_ctx *ctx, size_t bytes, void **hptr);
int zkhip_host_free(zkhip_ctx *ctx, void *hptr);

/* ---- resident bases: proving-key queries / SRS ------------------------------------------------
 * Replaces the `bases_begin, bases_end` iterator pair of algebra::multiexp at
 *   r1cs_gg_ppzksnark/prover.hpp:108-139 (A_query, B_query, H_query, L_query; layout proving_key.hpp:43-56)
 *   commitments/polynomial/kzg.hpp:143-148, 409-435 (params.commitment_key)
 * Upload once, reuse for every proof. */
int zkhip_bases_upload(zkhip_ctx *ctx, int curve, int group, const uint64_t *affine_xy, const uint8_t *is_infinity /* nullable */,
                       size_t n, zkhip_bases **out);
/* out[i] = scalars[i] * base (base == NULL: the standard generator), computed on the device and left
 * resident.  Replaces algebra::batch_exp / kc_batch_exp (generator.hpp:187-214,
 * knowledge_commitment_multiexp.hpp:143-208) and structured_generators_scalar_power (ipp2/srs.hpp:44-56). */
/* The same from the WIRE form of a proving key (g16/marshalling.hpp:111-112, 178-201): n compressed
 * BLS12-381 points, 48 bytes each for G1, 96 for G2 (big-endian x, G2: x.c1 then x.c0; byte 0 carries the flags
 * 0x80 compressed, 0x40 infinity, 0x20 larger y) -- the encoding pinned by the vectors of
 * r1cs_gg_ppzksnark_aggregation_conformity.cpp:932-1010.  Decoded on the device (one square root per point).
 * ZKHIP_ERR_INVALID if any encoding is malformed or not on the curve (zkhip_last_error says how many). */
int zkhip_bases_upload_compressed(zkhip_ctx *ctx, int curve, int group, const uint8_t *octets, size_t n, zkhip_bases **out);
int zkhip_bases_from_scalars(zkhip_ctx *ctx, int curve, int group, const uint64_t *base_affine_xy /* nullable */,
                             const uint64_t *scalars, size_t n, zkhip_bases **out);
/* A bases object of n_total points holding src's points at the rows d_rows[j] (a DEVICE array of src's size; NULL: the consecutive rows
 * first + j) and the point at infinity everywhere else, with its own window tables.  What it is for: queries of one proof that are
 * multiplied by the same assignment vector -- A_query, the B query's G1 half (sparse: its index list is d_rows), L_query (the auxiliary
 * part: first = n + 1) -- laid out over the SAME rows share one digit extraction and sort in zkhip_msm_batch_dev ("msm_share_sort").
 * d_rows must be STRICTLY INCREASING; a row beyond n_total or a repeated / out-of-order row raises the gather flag of
 * zkhip_device_status (a repeated row would overwrite a point). */
int zkhip_bases_spread(zkhip_ctx *ctx, const zkhip_bases *src, const uint32_t *d_rows /* nullable */, size_t first, size_t n_total,
                       zkhip_bases **out);
int zkhip_bases_download(zkhip_ctx *ctx, const zkhip_bases *b, size_t offset, size_t n, uint64_t *affine_xy, uint8_t *is_infinity);
size_t zkhip_bases_size(const zkhip_bases *b);
void zkhip_bases_free(zkhip_ctx *ctx, zkhip_bases *b);

/* ---- MSM ---------------------------------------------------------------------------------------
 * sum_{i<n} scalars[i] * bases[offset + i].
 * Replaces algebra::multiexp<multiexp_method_BDLO12>(b0, b1, s0, s1, chunks) and
 * multiexp_with_mixed_addition (prover.hpp:108-139, kzg.hpp:143-148,409-435,505-508,
 * knowledge_commitment_multiexp.hpp:107).  `chunks` has no counterpart: the device splits the work itself.
 * out_jacobian: 3 * (coordinate limbs) u64, canonical X | Y | Z of the group element; WHICH projective representative comes out is
 * not fixed (the order of additions inside a bucket follows the sort's atomics): compare points after zkhip_jacobian_to_affine,
 * as the reference's own operator== normalises before comparing.
 * The context keeps the device copy of the scalars (32 n bytes, grow-only, freed by zkhip_destroy): a prover's repeated calls neither allocate nor free, and
 * replay their launch sequence as a HIP graph. */
int zkhip_msm(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const uint64_t *scalars /* host, n x 4 */,
              uint64_t *out_jacobian /* host */);
/* Same, scalars and result resident in device memory; asynchronous on the context's stream. */
int zkhip_msm_dev(zkhip_ctx *ctx, const zkhip_bases *bases, size_t offset, size_t n, const void *d_scalars, void *d_out_jacobian);
/* `count` MSMs whose results are needed together (the queries of one Groth16 proof, the columns of a KZG batch):
 * same semantics as `count` calls of zkhip_msm_dev, but the latency-bound bucket reductions of all of them run as
 * one launch when the bases share group and window size (otherwise the calls simply run one after the other). */
int zkhip_msm_batch_dev(zkhip_ctx *ctx, size_t count, const zkhip_bases *const *bases, const size_t *offsets, const size_t *ns,
                        const void *const *d_scalars, void *const *d_out_jacobian);
/* d_out = sum of `count` Jacobian points (canonical, 3 coordinates each) resident in device memory: the
 * fold of per-GPU partial MSM results after the all-gather (RCCL has no elliptic-curve reduction). */
int zkhip_jacobian_sum_dev(zkhip_ctx *ctx, int curve, int group, const void *d_points, size_t count, void *d_out_jacobian);
/* Jacobian -> affine on the device (one inversion); is_infinity receives 0/1. */
int zkhip_jacobian_to_affine(zkhip_ctx *ctx, int curve, int group, const uint64_t *jacobian, uint64_t *affine_xy, uint8_t *is_infinity);

/* ---- NTT ---------------------------------------------------------------------------------------
 * In-place radix-2 transform of `batch` vectors of m = 2^log_m Fr elements, natural order in and out:
 *   forward:  [coset != NULL: v[j] *= coset^j;]  out[i] = sum_j v[j] omega^(ij)
 *   inverse:  out[j] = (1/m) sum_i v[i] omega^(-ij);  [coset != NULL: out[j] *= coset^(-j)]
 * Replaces evaluation_domain<F>::fft / inverse_fft and math::multiply_by_coset
 * (reductions/r1cs_to_qap.hpp:250-315; column_polynomial.hpp:53; the polynomial_dfs::coefficients /
 * resize calls of kzg.hpp:431 and basic_fri.hpp:452-455). */
int zkhip_ntt(zkhip_ctx *ctx, int curve, uint64_t *data /* host, batch x m x 4 */, size_t log_m, size_t batch,
              const uint64_t *omega /* 4 limbs */, int inverse, const uint64_t *coset_gen /* nullable, 4 limbs */);
int zkhip_ntt_dev(zkhip_ctx *ctx, int curve, void *d_data, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
                  const uint64_t *coset_gen);

/* ---- evaluation domains ------------------------------------------------------------------------------
 * What math::make_evaluation_domain<Fr>(min_size) returns (reductions/r1cs_to_qap.hpp:138-139, 229-230 call it with
 * num_constraints + num_inputs + 1, which is rarely a power of two).  crypto3-math is not part of the reference tree; the
 * family and the selection order are those of its libfqfft lineage: basic_radix2 (m = 2^k), extended_radix2
 * (m = 2^(s+1), s the field's two-adicity) and step_radix2 (m = 2^k + 2^j), tried at min_size and then at
 * big + rounded_small.  The point sets, in get_domain_element order:
 *   basic     omega^i, i < m                                    omega = primitive m-th root of unity
 *   extended  omega^i, then shift omega^i, i < m/2              omega = primitive (m/2)-th root
 *   step      omega^(2i), i < big, then omega (omega^(2 big/small))^i, i < small
 *                                                               omega = primitive (2 big)-th root (the domain's `omega` member)
 * The roots are ARGUMENTS like everywhere in this ABI (arithmetic_params<F> lives in crypto3-algebra); a root of the
 * wrong order is refused (ZKHIP_ERR_INVALID). */
enum zkhip_domain_kind { ZKHIP_DOMAIN_BASIC_RADIX2 = 0, ZKHIP_DOMAIN_EXTENDED_RADIX2 = 1, ZKHIP_DOMAIN_STEP_RADIX2 = 2 };
typedef struct zkhip_domain {
    int32_t kind;      /* zkhip_domain_kind */
    uint32_t reserved; /* 0 */
    uint64_t m;        /* number of points */
    uint64_t omega[4]; /* see above */
    uint64_t shift[4]; /* extended: detail::coset_shift<F>() (= multiplicative_generator^2); ignored otherwise */
} zkhip_domain;
/* (kind, m) make_evaluation_domain(min_size) picks over the curve's scalar field (two-adicity 32 for BLS12-381, 28 for BN254).
 * ZKHIP_ERR_RANGE when only a geometric / arithmetic sequence domain would do (beyond 2^(s+1) points: out of scope). */
int zkhip_domain_choice(int curve, size_t min_size, int *kind, size_t *m);
/* evaluation_domain::evaluate_all_lagrange_polynomials(t) (reductions/r1cs_to_qap.hpp:152-153, the key generator's QAP evaluation
 * at the trapdoor): d_out[i] = L_i(t) for every point of the domain, in get_domain_element order (m canonical Fr elements,
 * device), by the closed forms of the three radix-2 kinds with one batched inversion per 16 points.  ZKHIP_ERR_INVALID when t lies
 * in the domain (the closed forms do not cover it; a trapdoor must not lie there anyway). */
int zkhip_domain_lagrange_dev(zkhip_ctx *ctx, int curve, const zkhip_domain *dom, const uint64_t *t, void *d_out);
/* evaluation_domain<F>::fft / inverse_fft over any of the three domains, in place on `batch` vectors of dom->m Fr elements:
 *   forward:  [coset != NULL: v[j] *= coset^j;]  out[i] = sum_j v[j] x_i^j        (x_i = get_domain_element(i))
 *   inverse:  the interpolation;  [coset != NULL: out[j] *= coset^(-j)]
 * For the basic kind this is zkhip_ntt_dev. */
int zkhip_domain_fft_dev(zkhip_ctx *ctx, int curve, const zkhip_domain *dom, void *d_data, size_t batch, int inverse, const uint64_t *coset_gen);

/* ---- Groth16 witness map ------------------------------------------------------------------------
 * Resident R1CS in CSR form (three matrices A, B, C; column 0 is the constant 1, column j >= 1 is variable
 * j - 1; coefficients canonical Fr, 4 limbs each).  Replaces the constraint_system member of the proving key
 * (r1cs_gg_ppzksnark/proving_key.hpp:56; r1cs.hpp:61-64,125-133) as consumed by
 * reductions::r1cs_to_qap<F>::witness_map (reductions/r1cs_to_qap.hpp:219-325). */
typedef struct zkhip_r1cs zkhip_r1cs;
int zkhip_r1cs_upload(zkhip_ctx *ctx, int curve, size_t num_constraints, size_t num_inputs, size_t num_variables, const uint32_t *rowptr_a,
                      const uint32_t *col_a, const uint64_t *coeff_a, const uint32_t *rowptr_b, const uint32_t *col_b, const uint64_t *coeff_b,
                      const uint32_t *rowptr_c, const uint32_t *col_c, const uint64_t *coeff_c, zkhip_r1cs **out);
void zkhip_r1cs_free(zkhip_ctx *ctx, zkhip_r1cs *r);
/* The evaluation domain the witness map reduces over.  After upload it is (kind, m) = what
 * make_evaluation_domain(num_constraints + num_inputs + 1) picks (r1cs_to_qap.hpp:229-230; zkhip_domain_choice) -- a step
 * radix-2 domain for most instances.  zkhip_r1cs_set_domain installs another kind / size (m >= num_constraints + num_inputs + 1),
 * e.g. the basic domain of the next power of two for a key that was generated over that one; only kind and m are read
 * (the roots come with every witness-map call). */
int zkhip_r1cs_set_domain(zkhip_r1cs *r, int kind, size_t m);
size_t zkhip_r1cs_domain_size(const zkhip_r1cs *r);
int zkhip_r1cs_domain_kind(const zkhip_r1cs *r);
size_t zkhip_groth16_scratch_bytes(const zkhip_r1cs *r);
/* coefficients_for_H of witness_map with d1 = d2 = d3 = 0 (as prover.hpp:79-83 calls it): m + 1 Fr elements
 * written to d_h.  d_assignment = (1, primary_input, auxiliary_input), num_variables + 1 elements, device
 * resident; omega = the `omega` of the constraint system's evaluation domain (see zkhip_domain: the primitive m-th root for the
 * basic kind, the primitive (2 big)-th root for the step kind); coset_gen = the field's multiplicative generator
 * (arithmetic_params<F>::multiplicative_generator).  3 sparse mat-vecs, 7 transforms and one fused pointwise pass, all on
 * the context's stream; the result feeds zkhip_msm_dev directly.  An extended radix-2 domain also needs its shift: use
 * zkhip_groth16_witness_h_domain_dev, which takes the whole domain description (its kind and m must match the constraint
 * system's). */
int zkhip_groth16_witness_h_dev(zkhip_ctx *ctx, const zkhip_r1cs *r, const void *d_assignment, const uint64_t *omega,
                                const uint64_t *coset_gen, void *d_h, void *d_scratch);
int zkhip_groth16_witness_h_domain_dev(zkhip_ctx *ctx, const zkhip_r1cs *r, const void *d_assignment, const zkhip_domain *dom,
                                       const uint64_t *coset_gen, void *d_h, void *d_scratch);

/* d_dst[j] = d_src[d_indices[j]] on Fr elements (u32 indices): gathers the scalars of a sparse query, i.e. the
 * `*(scalar_start + scalar_position)` walk over vec.indices of kc_multiexp_with_mixed_addition
 * (knowledge_commitment_multiexp.hpp:66-100).  Zero and one scalars need no peeling on the device. */
/* d_src holds src_count elements; an index >= src_count (malformed key) gathers the zero scalar and raises the sticky
 * device status (zkhip_device_status). */
int zkhip_fr_gather_dev(zkhip_ctx *ctx, const void *d_src, size_t src_count, const void *d_indices, size_t count, void *d_dst);

/* ---- LPC / FRI polynomial helpers on top of the NTT -------------------------------------------------
 * polynomial_dfs::resize as precommit<FRI> applies it to every committed polynomial
 * (commitments/detail/polynomial/basic_fri.hpp:452-455): `batch` vectors of 2^log_n evaluations at d_in
 * (CONSUMED: left holding the coefficients) -> 2^log_out evaluations each at d_out.  omega_n / omega_out are the
 * primitive roots of the two domains.  Any pair of primitive roots is accepted; when they are NESTED (omega_out^(2^(log_out - log_n))
 * == omega_n, as roots drawn from one generator are) and the growth is at most 16-fold, only the new cosets are evaluated ("poly_coset_extend"). */
int zkhip_poly_resize_dev(zkhip_ctx *ctx, int curve, void *d_in, size_t log_n, size_t batch, const uint64_t *omega_n, void *d_out,
                          size_t log_out, const uint64_t *omega_out);
/* log_out < log_n (a SMALLER domain, for a polynomial whose degree fits it): d_out receives every 2^(log_n - log_out)-th
 * evaluation, d_in is left untouched.
 * math::polynomial_shift(f, shift, domain_size) on the evaluation vector (ph/permutation_argument.hpp:148,
 * lookup_argument.hpp:232): d_out[i] = d_in[(i + rotation) mod 2^log_size], rotation = shift * (size / domain_size), may be
 * negative; d_out must not alias d_in. */
int zkhip_poly_shift_dev(zkhip_ctx *ctx, const void *d_in, size_t log_size, int64_t rotation, void *d_out);
/* d_out[i] = prod_k d_in[k][i], i < n: the pointwise core of math::polynomial_product (ph/permutation_argument.hpp:148,
 * gates_argument.hpp:117) once every factor sits on the product's domain (zkhip_poly_resize_dev); d_in is a HOST array of
 * `count` device pointers, d_out may be one of them. */
int zkhip_fr_vec_prod_dev(zkhip_ctx *ctx, int curve, size_t count, const void *const *d_in, void *d_out, size_t n);

/* detail::fold_polynomial, DFS form (commitments/detail/polynomial/fold_polynomial.hpp:68-93):
 * d_out[i] = 1/2 [(1 + alpha omega^-i) d_f[i] + (1 - alpha omega^-i) d_f[i + size/2]], i < size/2 = 2^(log_size-1). */
int zkhip_fri_fold_dev(zkhip_ctx *ctx, int curve, const void *d_f, size_t log_size, const uint64_t *alpha, const uint64_t *omega, void *d_out);

/* The leaf layout precommit<FRI> feeds to the Merkle tree (basic_fri.hpp:456-492, FRI::m = 2): `batch` vectors of
 * 2^log_domain evaluations at d_polys -> d_out, 2^log_domain / 2^fri_step leaves of batch * 2^fri_step elements each:
 * leaf x = for every polynomial, the pairs (f[s_i], f[s_i + D/2]) in the reference's coset order.  Hashing is
 * the caller's. */
int zkhip_fri_leaves_dev(zkhip_ctx *ctx, const void *d_polys, size_t log_domain, size_t batch, size_t fri_step, void *d_out);

/* ---- DFT over group elements -------------------------------------------------------------------------------
 * evaluation_domain<Fr, G>::evaluate_all_lagrange_polynomials(powers_begin, powers_end) as the powers-of-tau result uses
 * it (commitments/detail/polynomial/powers_of_tau/result.hpp:81-94): with P_i = tau^i G the INVERSE transform gives
 * L_j(tau) G for every Lagrange polynomial of the 2^log_m-point domain.  d_jacobian: 2^log_m canonical Jacobian points
 * (3 coordinates each, Z = 0 for infinity -- the form zkhip_msm_dev writes), transformed in place:
 * out[j] = sum_i omega^(i j) P_i, or with inverse != 0: (1/m) sum_i omega^(-i j) P_i. */
int zkhip_ec_ntt_dev(zkhip_ctx *ctx, int curve, int group, void *d_jacobian, size_t log_m, const uint64_t *omega, int inverse);

/* ---- coefficient-form polynomial arithmetic (KZG opening proofs, polynomial_dfs pointwise operators) -------
 * All vectors are canonical Fr elements (4 limbs) resident on the device.
 *
 * polynomial_dfs::operator+= / -= / *= on equal domains (ph/gates_argument.hpp:119-121,
 * ph/permutation_argument.hpp:148-167): d_out[i] = d_a[i] op d_b[i], op 0 add, 1 sub, 2 mul; in place allowed. */
int zkhip_fr_vec_op_dev(zkhip_ctx *ctx, int curve, int op, const void *d_a, const void *d_b, void *d_out, size_t count);
/* d_out[i] = a d_x[i] + b d_y[i] + c (d_y and b nullable together; a, b, c canonical Fr on the host; in place allowed): the linear
 * factors the arguments multiply up -- (1 + beta)(gamma + input), (1 + beta) gamma + value + beta value(omega X)
 * (ph/lookup_argument.hpp:313, 329, 361), mask = 1 - q_last - q_blind (:161-162) -- in one pass each. */
int zkhip_fr_vec_affine_dev(zkhip_ctx *ctx, int curve, const void *d_x, const void *d_y, const uint64_t *a, const uint64_t *b, const uint64_t *c, void *d_out,
                            size_t count);
/* d_out[j] = d_a[j] d_b[j] / d_c[j] for j < count (in place allowed; entries from `count` on are not touched): the intermediate
 * polynomials of the multi-part permutation / lookup arguments, current[j] = previous[j] g[j] / h[j] over the usable rows
 * (ph/permutation_argument.hpp:196-198, lookup_argument.hpp:264-266) -- rows in chunks of 8 that share one inversion; a zero
 * denominator zeroes its chunk. */
int zkhip_fr_vec_mul_div_dev(zkhip_ctx *ctx, int curve, const void *d_a, const void *d_b, const void *d_c, void *d_out, size_t count);
/* polynomial::evaluate for `batch` polynomials (n coefficients each, `stride` elements apart) at `npoints` points
 * given on the host (eval_polys, batched_commitment.hpp:168-183): out[b * npoints + p] = poly_b(points[p]), host. */
int zkhip_poly_eval_dev(zkhip_ctx *ctx, int curve, const void *d_polys, size_t n, size_t stride, size_t batch, const uint64_t *points,
                        size_t npoints, uint64_t *out);
/* Division by (X - z), the step `f /= V`, `L /= theta_2_vanish` of kzg_v2.hpp:267, 289 performs once per root:
 * d_out[0] = f(z) (the remainder), d_out[1 .. n) = the quotient's n - 1 coefficients.  d_out may equal d_f.
 * `remainder` (nullable, host) receives f(z); passing it synchronises the stream. */
int zkhip_poly_div_linear_dev(zkhip_ctx *ctx, int curve, const void *d_f, size_t n, const uint64_t *z, void *d_out, uint64_t *remainder);
/* Exact division by the vanishing polynomial X^n - 1 of the basic n-point domain, in coefficient form: placeholder's
 * `T_consolidated = F_consolidated_normal / common_data.Z` (ph/prover.hpp:273-275).  d_f: len coefficients; d_quot receives the
 * len - n coefficients of the quotient (nothing when len <= n); d_quot must not overlap d_f.  nonzero_remainders (nullable, host;
 * passing it synchronises the stream) receives the number of non-zero coefficients of the remainder -- 0 for the exact division
 * a satisfied circuit gives. */
int zkhip_poly_div_vanishing_dev(zkhip_ctx *ctx, int curve, const void *d_f, size_t len, size_t n, void *d_quot, uint64_t *nonzero_remainders);
/* The grand product of placeholder's permutation argument (ph/permutation_argument.hpp:103-136): over n rows and k permuted columns
 *   g_i = column_i + beta S_id_i + gamma,  h_i = column_i + beta S_sigma_i + gamma   (pointwise),
 *   V_P[0] = 1,  V_P[j] = V_P[j - 1] prod_i g_i[j - 1] / prod_i h_i[j - 1].
 * d_cols / d_sid / d_ssigma: HOST arrays of k device pointers to n canonical Fr each; d_g / d_h (nullable): k x n outputs, vector i at
 * element offset i n (the g_v / h_v the argument multiplies up afterwards); d_vp: n outputs.  The reference pays one inversion per row
 * in a serial loop; here rows are taken in chunks that share an inversion and the prefix product is a three-level scan. */
int zkhip_perm_grand_product_dev(zkhip_ctx *ctx, int curve, size_t k, const void *const *d_cols, const void *const *d_sid, const void *const *d_ssigma,
                                 size_t n, const uint64_t *beta, const uint64_t *gamma, void *d_g, void *d_h, void *d_vp);
/* The g and h polynomials of the same argument (ph/permutation_argument.hpp:140-160: polynomial_product(g_v), polynomial_product(h_v)) from vectors on
 * ANY domain: d_g[j] = prod_i (column_i[j] + beta S_id_i[j] + gamma), d_h[j] likewise over S_sigma, j < n.  With the columns and the (cached)
 * permutation polynomials extended to the products' domain this is ONE pass instead of 2 k extensions, 2 k linear passes and two k-way products. */
int zkhip_perm_factor_products_dev(zkhip_ctx *ctx, int curve, size_t k, const void *const *d_cols, const void *const *d_sid, const void *const *d_ssigma, size_t n,
                                   const uint64_t *beta, const uint64_t *gamma, void *d_g, void *d_h);
/* V_L of placeholder's lookup argument (ph/lookup_argument.hpp:375-409, `compute_V_L`), the same scan over other rows:
 *   V_L[0] = 1,  V_L[k] = V_L[k - 1] g(k - 1) / h(k - 1) for 1 <= k <= usable_rows,  V_L[k] = 0 behind,
 *   g(j) = (1 + beta)^k_in prod_i (gamma + input_i[j]) prod_i ((1 + beta) gamma + value_i[j] + beta value_i[j + 1]),
 *   h(j) = prod_i ((1 + beta) gamma + sorted_i[j] + beta sorted_i[j + 1]).
 * d_input / d_value / d_sorted: HOST arrays of device pointers to n canonical Fr each (the REDUCED vectors: on the basic domain);
 * usable_rows < n (ZKHIP_ERR_RANGE otherwise); d_vl: n outputs. */
int zkhip_lookup_grand_product_dev(zkhip_ctx *ctx, int curve, size_t k_in, const void *const *d_input, size_t k_val, const void *const *d_value, size_t k_sorted,
                                   const void *const *d_sorted, size_t n, size_t usable_rows, const uint64_t *beta, const uint64_t *gamma, void *d_vl);
/* sort_polynomials of placeholder's lookup argument (ph/lookup_argument.hpp:565-638; called at :188 between the reduction of the
 * lookup inputs / values to the basic domain and `commit(LOOKUP_BATCH)`).  Over the first usable_rows entries of every vector: the
 * values of the table columns in their order (columns one after the other), every maximal run of one value replaced by as many copies
 * as the value occurs among ALL table entries and inputs -- a run of zeros by a single zero, a trailing run of zeros by nothing, and one
 * zero in front when the first value is non-zero (the reference's walk starts from a virtual zero) --, dealt over the k_in + k_val
 * output vectors, usable_rows entries each; entry usable_rows of every vector but the last repeats the head of the next; all other
 * entries are zero.  d_input / d_value / d_sorted: HOST arrays of device pointers to n canonical Fr each (the REDUCED vectors);
 * usable_rows < n.  Equality is equality of the canonical limbs.  The reference counts with an unordered_map and emits in one serial
 * walk; here: run detection + u32 scans, an open-addressing hash table over the run heads, one atomic per input, a binary search per
 * output entry.  A looked-up value that is in no table (the reference's BOOST_ASSERT, :583) raises bit 2 of the sticky status word and
 * is otherwise ignored; more emitted entries than the vectors hold (a table whose equal values are not adjacent, :559-564) raise bit 3
 * and the excess is dropped. */
int zkhip_lookup_sort_dev(zkhip_ctx *ctx, size_t k_in, const void *const *d_input, size_t k_val, const void *const *d_value, size_t n, size_t usable_rows,
                          void *const *d_sorted);
/* d_acc[j] (+)= sum_i sum_{t < taps} coeffs[i * taps + t] * poly_i[j - t] for j < acc_len (poly_i is zero outside
 * [0, lens[i])): the accumulation `f += theta_i * (f_i - U) * diffpoly` (kzg_v2.hpp:258-263) for every committed
 * polynomial in ONE pass (coeffs[i] = theta_i * diffpoly_i, a few taps), and `L += ...` (:281-288) with taps = 1.
 * d_polys / lens / coeffs are host arrays (device pointers, element counts, canonical Fr); accumulate = 0
 * overwrites d_acc. */
int zkhip_poly_lincomb_dev(zkhip_ctx *ctx, int curve, size_t count, const void *const *d_polys, const size_t *lens, const uint64_t *coeffs,
                           size_t taps, void *d_acc, size_t acc_len, int accumulate);

/* ---- the gate argument's sum over a flat program -------------------------------------------------------------------------
 * placeholder's gates argument (ph/gates_argument.hpp:93-121, 203-216) computes, on the extended domain of 2^log_size points,
 *     F = mask * sum_gates selector_g * sum_constraints theta^k * constraint(columns, rotated).
 * Which monomials a constraint is made of is decided by the reference's symbolic machinery (math::expression and its visitors: the
 * caller's, SURVEY section 2 out of scope); once flattened it is a PROGRAM over column slots, evaluated here in ONE launch, a lane per row:
 *   d_out[j] = [d_mask[j] *] ( [d_out[j] +] sum_g [slot_{sel_g}[j + rs_g] *] sum_{t in gate g} coeff_t prod_{f in term t} slot_{s_f}[(j + r_f) mod 2^log_size] )
 * -- rotations are index arithmetic on the extended domain (a rotation by k rows of the ORIGINAL n-row domain is k * 2^log_size / n here;
 * no math::polynomial_shift copies), every distinct column is read where it lies, selector and mask are multiplied in the same pass.
 * The expressions of prepare_lookup_input (ph/lookup_argument.hpp:435-496) are the same program without selectors and mask.
 * All arrays of the program are HOST memory; d_slots is a host array of n_slots device pointers to 2^log_size canonical Fr each.
 * accumulate != 0 adds to what d_out holds (a program evaluated in pieces: the mask then belongs to the LAST piece).
 * A malformed program (ranges out of order, a slot that does not exist) is refused with ZKHIP_ERR_INVALID / ZKHIP_ERR_RANGE. */
#define ZKHIP_GATE_NO_SELECTOR 0xFFFFFFFFu
typedef struct zkhip_gate_program {
    uint32_t n_gates, n_terms, n_factors, n_slots;
    const uint32_t *gate_terms;        /* n_gates + 1: gate g owns terms [gate_terms[g], gate_terms[g + 1]) */
    const uint32_t *gate_selector;     /* n_gates: slot of the gate's selector, or ZKHIP_GATE_NO_SELECTOR */
    const int32_t *gate_selector_rot;  /* n_gates: its rotation, in rows of THIS domain */
    const uint32_t *term_factors;      /* n_terms + 1: term t owns factors [term_factors[t], term_factors[t + 1]) (none: the constant coeff_t) */
    const uint32_t *factor_slot;       /* n_factors */
    const int32_t *factor_rot;         /* n_factors: rows of THIS domain, may be negative */
    const uint64_t *term_coeff;        /* n_terms x 4 limbs, canonical */
} zkhip_gate_program;
int zkhip_gate_eval_dev(zkhip_ctx *ctx, int curve, const zkhip_gate_program *prog, const void *const *d_slots, size_t log_size, const void *d_mask /* nullable */,
                        int accumulate, void *d_out);

/* ---- device group: N GPUs behind ONE caller ---------------------------------------------------------------------
 * The reference hides all of its parallelism INSIDE the call: r1cs_gg_ppzksnark_prover::process splits every multiexp into
 * `chunks = omp_get_max_threads()` pieces (r1cs_gg_ppzksnark/prover.hpp:94-99, 108-139) and kzg_commitment_scheme_v2::commit loops
 * over the batch (commitments/polynomial/kzg_v2.hpp:208-226) -- the caller sees one call and one result.  A group is the same for
 * GPUs: one zkhip_ctx per entry of device_ids (SURVEY 8b sketched this as `zkhip_init(const int *device_ids, int n_dev, ...)`),
 * all driven from ONE host thread (every entry point of this ABI is asynchronous on its context's stream), with the one exchange the
 * path has -- the partial sums of a sharded multiexp, <= 864 bytes per member for a Groth16 proof -- INSIDE the library.
 * device_ids may name a device more than once (several members on one GPU: a 1-GPU box, tests).
 * Members are ordinary contexts (zkhip_group_ctx): every other entry point of this header works on them; they are destroyed with
 * the group.  A group is not thread-safe. */
typedef struct zkhip_device_group zkhip_device_group;
int zkhip_group_init(const int *device_ids, int n_dev, zkhip_device_group **out);
void zkhip_group_destroy(zkhip_device_group *g);
int zkhip_group_size(const zkhip_device_group *g);
zkhip_ctx *zkhip_group_ctx(const zkhip_device_group *g, int member);
const char *zkhip_group_last_error(const zkhip_device_group *g);
/* How the exchanges travel.
 *   ZKHIP_GROUP_RCCL    one single-process RCCL communicator per member (ncclCommInitAll) and a grouped ncclAllGather on the members'
 *                       streams over xGMI; librccl.so is loaded on first use (dlopen), so a single-GPU caller never pays for it.
 *                       Needs pairwise distinct devices (RCCL refuses two ranks on one GPU): ZKHIP_ERR_INVALID otherwise.
 *   ZKHIP_GROUP_PEER    stream-ordered peer copies (hipMemcpyPeerAsync behind an event per source stream -- the same call between members
 *                       that share a GPU, so a one-GPU box exercises what a multi-GPU box runs).  No host synchronisation.
 *   ZKHIP_GROUP_STAGED  through one page-locked host buffer: D2H on every member, a host synchronisation, H2D -- also between members
 *                       that share a GPU (one code path on every box).  The fallback that works wherever HIP works.
 *   ZKHIP_GROUP_AUTO    (default) RCCL when the group has more than one member on pairwise distinct devices and librccl loads,
 *                       PEER otherwise.
 * zkhip_group_transport returns what AUTO resolved to (RCCL is tried at the first exchange). */
enum zkhip_group_transport_kind { ZKHIP_GROUP_AUTO = 0, ZKHIP_GROUP_RCCL = 1, ZKHIP_GROUP_PEER = 2, ZKHIP_GROUP_STAGED = 3 };
int zkhip_group_set_transport(zkhip_device_group *g, int transport);
int zkhip_group_transport(const zkhip_device_group *g);
/* All-gather in stream order: the `bytes` bytes at d_send[k] (member k's device memory) arrive at d_recv[j] + k * bytes for every
 * member j whose d_recv[j] is not NULL (a NULL entry: that member receives nothing -- the one-host-thread caller usually needs the
 * result on member 0 only).  d_send / d_recv: HOST arrays of zkhip_group_size device pointers.  Ordered after everything enqueued
 * on the members' streams so far; with RCCL / PEER nothing blocks the host.  This is the exchange SURVEY 8e names (ncclAllGather
 * of 144- / 288-byte Jacobian partial sums + a local fold: RCCL has no elliptic-curve reduction). */
int zkhip_group_all_gather(zkhip_device_group *g, const void *const *d_send, void *const *d_recv, size_t bytes);
/* d_dst (member dst) <- d_src (member src), `bytes` bytes, ordered after src's stream so far, enqueued on dst's stream.  The source
 * must stay untouched until dst's stream has passed the copy (zkhip_sync on dst, or a later exchange). */
int zkhip_group_copy(zkhip_device_group *g, int dst_member, void *d_dst, int src_member, const void *d_src, size_t bytes);
int zkhip_group_sync(zkhip_device_group *g); /* every member's stream has drained */
/* Resident bases cut by POINT RANGE over the members: member k holds points [k n / N, (k + 1) n / N) (balanced; the
 * first n mod N members hold one more) with their own window tables.  Same arguments as zkhip_bases_upload. */
typedef struct zkhip_group_bases zkhip_group_bases;
int zkhip_group_bases_upload(zkhip_device_group *g, int curve, int group, const uint64_t *affine_xy, const uint8_t *is_infinity /* nullable */, size_t n,
                             zkhip_group_bases **out);
int zkhip_group_bases_from_scalars(zkhip_device_group *g, int curve, int group, const uint64_t *base_affine_xy /* nullable */, const uint64_t *scalars, size_t n,
                                   zkhip_group_bases **out);
void zkhip_group_bases_free(zkhip_device_group *g, zkhip_group_bases *b);
size_t zkhip_group_bases_size(const zkhip_group_bases *b);
/* member k's slice as an ordinary bases object (owned by the group object); *first receives the index of its first point */
const zkhip_bases *zkhip_group_bases_member(const zkhip_group_bases *b, int member, size_t *first /* nullable */);
/* zkhip_msm over the group: algebra::multiexp<BDLO12>(b0, b1, s0, s1, chunks) with the chunks on N GPUs (prover.hpp:94-99 chooses
 * chunks = omp_get_max_threads(); here a chunk is a member's point range).  Every member receives its slice of the scalars and
 * runs the whole pipeline over its points; the partial sums are all-gathered to member 0, folded there
 * (zkhip_jacobian_sum_dev) and returned.  Same result as zkhip_msm over one device (compare in affine). */
int zkhip_group_msm(zkhip_device_group *g, const zkhip_group_bases *bases, size_t offset, size_t n, const uint64_t *scalars /* host, n x 4 */,
                    uint64_t *out_jacobian /* host */);
/* zkhip_ntt with the batch dealt over the members in contiguous ranges ("partition by polynomial, no collective"):
 * member k transforms vectors [k batch / N, (k + 1) batch / N).  Bit-identical to zkhip_ntt. */
int zkhip_group_ntt(zkhip_device_group *g, int curve, uint64_t *data /* host, batch x m x 4 */, size_t log_m, size_t batch, const uint64_t *omega, int inverse,
                    const uint64_t *coset_gen /* nullable */);

/* ---- profiling (HIP events on the context's stream around every kernel launch) ------------------ */
int zkhip_profile_enable(zkhip_ctx *ctx, int on);
int zkhip_profile_reset(zkhip_ctx *ctx);
/* Only kernels whose name starts with `prefix` are timed from now on (NULL / "": all): keeps a timed region's event traffic to
 * the one kernel whose duration is wanted. */
int zkhip_profile_filter(zkhip_ctx *ctx, const char *prefix);
/* Total milliseconds and launch count recorded for kernels whose name starts with `prefix`. */
int zkhip_profile_get(zkhip_ctx *ctx, const char *prefix, double *total_ms, uint64_t *launches);
/* Writes "name total_ms launches\n" lines; returns bytes needed. */
size_t zkhip_profile_dump(zkhip_ctx *ctx, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
