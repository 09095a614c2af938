#!/usr/bin/env python3
"""Extract the reference's own known-answer DATA (integer literals only) into ref_kat.json.

Run in the build container only (needs /root/reference); the resulting JSON is committed and is all
that travels.  Sources (data literals, no code):
  AGG = test/systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark_aggregation_conformity.cpp
    :578-862  bls381_polynomial_test   (r_shift, 8 transcript scalars, 256 expected coefficients,
                                        kzg_challenge, expected evaluation)
    :864-930  bls381_prove_commitment_test (alpha, beta, kzg_challenge, 3 transcript scalars, r_shift,
                                        expected 2 x G2 and 2 x G1 affine points)
    :932-1010 bls381_transcript_test, serialisation part (an Fr element with its 32 little-endian bytes, a G1 point with
                                        its 48 compressed bytes, a G2 point with its 96 compressed bytes)
  test/commitment/kzg.cpp:75-103        kzg_basic_test  (alpha = 10, f = {-1, 1, 2, 3}, commit = 3209 * G)
"""
import json
import os
import re

REF = "/root/reference"
AGG = os.path.join(REF, "test/systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark_aggregation_conformity.cpp")


def hexes(text, suffix):
    return [int(h, 16) for h in re.findall(r"0x([0-9a-fA-F]+)_cppui_modular" + suffix, text)]


def main():
    lines = open(AGG).read().split("\n")

    def block(start_marker):
        i = next(k for k, l in enumerate(lines) if start_marker in l)
        j = next(k for k in range(i + 1, len(lines)) if lines[k].startswith("BOOST_AUTO_TEST_CASE"))
        return "\n".join(lines[i:j])

    poly = block("BOOST_AUTO_TEST_CASE(bls381_polynomial_test)")
    fr = hexes(poly, "255")
    # order in the file: r_shift, 8 tr, 256 coeffs, kzg_challenge, eval
    assert len(fr) == 1 + 8 + 256 + 2, len(fr)
    out = {
        "source": "NilFoundation/crypto3-zk test vectors (bellperson-derived), literals only",
        "polynomial_test": {
            "r_shift": hex(fr[0]),
            "tr": [hex(x) for x in fr[1:9]],
            "coeffs": [hex(x) for x in fr[9:265]],
            "kzg_challenge": hex(fr[265]),
            "eval": hex(fr[266]),
        },
    }
    pc = block("BOOST_AUTO_TEST_CASE(bls381_prove_commitment_test)")
    fr = hexes(pc, "255")
    fq = hexes(pc, "381")
    assert len(fr) == 7 and len(fq) == 12, (len(fr), len(fq))
    out["prove_commitment_test"] = {
        "n": 8,
        "alpha": hex(fr[0]),
        "beta": hex(fr[1]),
        "kzg_challenge": hex(fr[2]),
        "tr": [hex(x) for x in fr[3:6]],
        "r_shift": hex(fr[6]),
        # G2 points as [[x.c0, x.c1], [y.c0, y.c1]]
        "comm_v": [
            [[hex(fq[0]), hex(fq[1])], [hex(fq[2]), hex(fq[3])]],
            [[hex(fq[4]), hex(fq[5])], [hex(fq[6]), hex(fq[7])]],
        ],
        "comm_w": [[hex(fq[8]), hex(fq[9])], [hex(fq[10]), hex(fq[11])]],
    }
    out["kzg_basic_test"] = {"alpha": 10, "f": [-1, 1, 2, 3], "commit_scalar": 3209}
    # serialised forms (bincode::curve<bls12<381>>: the ZCash compressed encoding of points, little-endian scalars)
    tr = block("BOOST_AUTO_TEST_CASE(bls381_transcript_test)")
    tr = tr[: tr.index("fq12_value_type d(")]
    fr = hexes(tr, "255")
    fq = hexes(tr, "381")

    def byte_list(name):
        body = tr[tr.index(name + " = {") + len(name) + 4:]
        return [int(x) for x in re.findall(r"\d+", body[: body.index("}")])]

    assert len(fr) == 1 and len(fq) == 2 + 4, (len(fr), len(fq))
    out["serialisation_test"] = {
        "fr": hex(fr[0]), "fr_bytes": byte_list("et_a_ser"),
        "g1": [hex(fq[0]), hex(fq[1])], "g1_bytes": byte_list("et_b_ser"),
        # [[x.c0, x.c1], [y.c0, y.c1]]
        "g2": [[hex(fq[2]), hex(fq[3])], [hex(fq[4]), hex(fq[5])]], "g2_bytes": byte_list("et_c_ser"),
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kat.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
