//---------------------------------------------------------------------------//
// zkhip shim: a Groth16 proving key from its wire form straight onto the MI355X.
//
// Mirrors verifier_input_deserializer_tvm<r1cs_gg_ppzksnark<bls12<381>>>::proving_key_process
// (zk/snark/systems/ppzksnark/r1cs_gg_ppzksnark/marshalling.hpp:656-760) and the pieces it is made of
// (:203-372 constraint system, :375-462 the sparse (G2, G1) query, :466-492 4-byte big-endian counts):
//   alpha_g1 | beta_g1 | beta_g2 | delta_g1 | delta_g2
//   | u32 A_count | A points | u32 B_bytes | [u32 count | count x u32 index | count x (G2 | G1) | u32 domain_size]
//   | u32 H_count | H points | u32 L_count | L points
//   | u32 primary | u32 auxiliary | u32 constraints | per constraint: u32 bytes | lc a | lc b | lc c
//   linear combination = u32 terms | terms x (u32 index | Fr, 32 bytes little-endian)
// with G1 = 48 and G2 = 96 bytes of compressed point.  The reference decompresses every point on the host (a field
// square root each) into crypto3 objects that the prover then converts again; here the query blobs go to the device
// as they are and are decoded there (zkhip_bases_upload_compressed) -- the host only walks the framing.
//---------------------------------------------------------------------------//
#ifndef ZKHIP_SHIM_MARSHALLING_HPP
#define ZKHIP_SHIM_MARSHALLING_HPP

#include <memory>
#include <vector>

#include "r1cs_gg_ppzksnark.hpp"

namespace nil {
namespace crypto3 {
namespace zk {
namespace hip {

/// a proving key whose queries live only on the device; `host` holds the five single elements and the constraint system
template <typename CurveType>
struct loaded_proving_key {
    r1cs_gg_ppzksnark_proving_key<CurveType> host;
    std::unique_ptr<r1cs_gg_ppzksnark_proving_key_hip<CurveType>> device;
};

namespace detail {
    struct byte_reader {
        const std::uint8_t *p, *end;
        const std::uint8_t *take(std::size_t n) {
            if ((std::size_t)(end - p) < n) throw std::runtime_error("proving key blob: not enough data");    // status_type::not_enough_data
            const std::uint8_t *r = p;
            p += n;
            return r;
        }
        std::size_t u32() {    // std_size_t_process: 4 bytes, big-endian
            const std::uint8_t *b = take(4);
            return ((std::size_t)b[0] << 24) | ((std::size_t)b[1] << 16) | ((std::size_t)b[2] << 8) | b[3];
        }
    };
}    // namespace detail

/// nil::marshalling::pack<endianness>(G1 point) for BLS12-381 as the KZG schemes apply it to commitments (kzg_v2.hpp:217-221): the
/// 48-byte compressed encoding (byte 0: 0x80 compressed | 0x40 infinity | 0x20 "y is the lexicographically larger root"; x big-endian)
/// -- the encoding the reference's own byte vectors pin (aggregation test, r1cs_gg_ppzksnark_aggregation_conformity.cpp:932-1010; the
/// oracle's bls12_381_compress reproduces them, tests/test_oracle_kat.py).  A ready-made `Packer` for the placeholder-facing schemes.
template <typename CurveType>
struct bls12_381_g1_packer {
    typedef curve_adapter<CurveType> adapter;
    std::vector<std::uint8_t> operator()(const typename adapter::g1_value_type &p) const {
        static_assert(adapter::g1_coord_limbs == 6, "BLS12-381: 381-bit coordinates");
        std::vector<std::uint8_t> b(48, 0);
        std::uint64_t xy[12];
        if (!p.to_affine(xy)) {
            b[0] = 0xC0;
            return b;
        }
        for (int i = 0; i < 48; ++i) b[47 - i] = (std::uint8_t)(xy[i >> 3] >> (8 * (i & 7)));
        /* y > p - y  <=>  2 y > p (y < p): compare 2 y with the modulus, most significant limb first */
        static const std::uint64_t P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                           0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
        std::uint64_t y2[7];
        std::uint64_t carry = 0;
        for (int i = 0; i < 6; ++i) {
            y2[i] = (xy[6 + i] << 1) | carry;
            carry = xy[6 + i] >> 63;
        }
        y2[6] = carry;
        bool larger = y2[6] != 0;
        if (!larger)
            for (int i = 5; i >= 0; --i)
                if (y2[i] != P[i]) {
                    larger = y2[i] > P[i];
                    break;
                }
        b[0] |= 0x80 | (larger ? 0x20 : 0);
        return b;
    }
};

/// proving_key_process: `blob` is the serialized proving key; `dom` the evaluation-domain constants (see domain_params)
template <typename CurveType>
std::unique_ptr<loaded_proving_key<CurveType>> proving_key_from_bytes(const context &ctx, const std::uint8_t *blob, std::size_t size,
                                                                       const domain_params<CurveType> &dom) {
    typedef curve_adapter<CurveType> adapter;
    constexpr std::size_t G1B = 8 * adapter::g1_coord_limbs, G2B = 8 * adapter::g2_coord_limbs, FRB = 32;
    detail::byte_reader rd {blob, blob + size};
    std::unique_ptr<loaded_proving_key<CurveType>> key(new loaded_proving_key<CurveType>());
    auto &pk = key->host;
    /* the five single elements: decoded by the same device path, then read back */
    {
        std::vector<std::uint8_t> g1s, g2s;
        auto put = [&](std::vector<std::uint8_t> &v, std::size_t n) {
            const std::uint8_t *b = rd.take(n);
            v.insert(v.end(), b, b + n);
        };
        put(g1s, G1B);    // alpha_g1
        put(g1s, G1B);    // beta_g1
        put(g2s, G2B);    // beta_g2
        put(g1s, G1B);    // delta_g1
        put(g2s, G2B);    // delta_g2
        auto d1 = device_bases<CurveType, ZKHIP_G1>::from_compressed(ctx, g1s.data(), 3);
        auto d2 = device_bases<CurveType, ZKHIP_G2>::from_compressed(ctx, g2s.data(), 2);
        pk.alpha_g1 = d1.at(0);
        pk.beta_g1 = d1.at(1);
        pk.delta_g1 = d1.at(2);
        pk.beta_g2 = d2.at(0);
        pk.delta_g2 = d2.at(1);
    }
    const std::size_t a_count = rd.u32();
    auto a_query = device_bases<CurveType, ZKHIP_G1>::from_compressed(ctx, rd.take(a_count * G1B), a_count);
    /* B query: sparse vector of (G2, G1) pairs (g2g1_knowledge_commitment_vector_process) */
    const std::size_t b_bytes = rd.u32();
    detail::byte_reader rb {rd.take(b_bytes), nullptr};
    rb.end = rb.p + b_bytes;
    const std::size_t b_count = rb.u32();
    /* count, b_count indices, b_count (g, h) pairs, domain size: checked BEFORE anything is sized from the count */
    if (b_count > (b_bytes - 8) / (4 + G2B + G1B) || b_bytes < 8) throw std::runtime_error("proving key blob: B query count exceeds its byte length");
    std::vector<std::uint32_t> b_indices(b_count);
    for (auto &i : b_indices) i = (std::uint32_t)rb.u32();
    std::vector<std::uint8_t> bg(b_count * G2B), bh(b_count * G1B);
    for (std::size_t i = 0; i < b_count; ++i) {    // element_kc on the wire: g (G2) then h (G1)
        const std::uint8_t *e = rb.take(G2B + G1B);
        std::copy(e, e + G2B, bg.begin() + i * G2B);
        std::copy(e + G2B, e + G2B + G1B, bh.begin() + i * G1B);
    }
    pk.B_query.domain_size_ = rb.u32();
    auto b_query_g = device_bases<CurveType, ZKHIP_G2>::from_compressed(ctx, bg.data(), b_count);
    auto b_query_h = device_bases<CurveType, ZKHIP_G1>::from_compressed(ctx, bh.data(), b_count);
    const std::size_t h_count = rd.u32();
    auto h_query = device_bases<CurveType, ZKHIP_G1>::from_compressed(ctx, rd.take(h_count * G1B), h_count);
    const std::size_t l_count = rd.u32();
    auto l_query = device_bases<CurveType, ZKHIP_G1>::from_compressed(ctx, rd.take(l_count * G1B), l_count);
    /* r1cs_constraint_system_process */
    auto &cs = pk.constraint_system;
    cs.primary_input_size = rd.u32();
    cs.auxiliary_input_size = rd.u32();
    const std::size_t rc_count = rd.u32();
    for (std::size_t i = 0; i < rc_count; ++i) {
        const std::size_t bytes = rd.u32();
        detail::byte_reader rc {rd.take(bytes), nullptr};
        rc.end = rc.p + bytes;
        r1cs_constraint<CurveType> c;
        linear_combination<CurveType> *lc[3] = {&c.a, &c.b, &c.c};
        for (int k = 0; k < 3; ++k) {
            const std::size_t terms = rc.u32();
            for (std::size_t t = 0; t < terms; ++t) {
                const std::size_t index = rc.u32();
                const std::uint8_t *f = rc.take(FRB);    // little-endian canonical integer
                std::uint64_t limbs[4];
                for (int w = 0; w < 4; ++w) {
                    limbs[w] = 0;
                    for (int b = 7; b >= 0; --b) limbs[w] = (limbs[w] << 8) | f[8 * w + b];
                }
                lc[k]->add_term(index, adapter::scalar_from_limbs(limbs));
            }
        }
        cs.add_constraint(c);
    }
    key->device.reset(new r1cs_gg_ppzksnark_proving_key_hip<CurveType>(ctx, pk, dom, std::move(a_query), std::move(b_query_g), std::move(b_query_h), b_indices,
                                                                      std::move(h_query), std::move(l_query)));
    return key;
}

}    // namespace hip
}    // namespace zk
}    // namespace crypto3
}    // namespace nil

#endif    // ZKHIP_SHIM_MARSHALLING_HPP
