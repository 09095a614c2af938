"""Proving-key points from their wire form (zkhip_bases_upload_compressed): the device decodes the ZCash compressed
BLS12-381 encodings the reference's serializers emit (g16/marshalling.hpp:111-112, 178-201).  Checked against the
reference's own literal vectors (AGG:932-1010, tests/golden/ref_kat.json), against the oracle on random points of
both signs incl. infinity, and on malformed input; and an MSM over a key loaded this way."""
import json
import os

import numpy as np
import pytest

import cport as cp
import pyoracle as po
from util import limbs, pt_from_limbs, pt_limbs

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kat.json")))["serialisation_test"]


def _points(b, group):
    pts, inf = b.download()
    return [pt_from_limbs(0, group, pts[i], inf[i]) for i in range(len(inf))]


def test_reference_serialisation_vectors_on_device(ctx):
    g1 = (int(KAT["g1"][0], 16), int(KAT["g1"][1], 16))
    b = ctx.upload_bases_compressed(0, 1, bytes(KAT["g1_bytes"]), 1)
    assert _points(b, 1) == [g1]
    b.free()
    g2 = tuple((int(c[0], 16), int(c[1], 16)) for c in KAT["g2"])
    b = ctx.upload_bases_compressed(0, 2, bytes(KAT["g2_bytes"]), 1)
    assert _points(b, 2) == [g2]
    b.free()


@pytest.mark.parametrize("group,n", [(1, 300), (2, 120)])
def test_compressed_round_trip_and_msm(ctx, group, n):
    G = po.BLS12_381.g1 if group == 1 else po.BLS12_381.g2
    ks = cp.random_fr(0, 91, n)
    pts, inf = cp.batch_mul(0, group, ks)
    P = [pt_from_limbs(0, group, pts[i], inf[i]) for i in range(n)]
    P[3] = None          # infinity
    P[5] = G.neg(P[4])   # both roots of one x
    P[7] = G.gen
    blob = b"".join(po.bls12_381_compress(group, p) for p in P)
    assert {bool(c[0] & 0x20) for c in (blob[i * 48 * group:(i + 1) * 48 * group] for i in range(n))} == {True, False}
    b = ctx.upload_bases_compressed(0, group, blob, n)
    assert _points(b, group) == P
    # the loaded key behaves like an uploaded one
    sc = cp.random_fr(0, 92, n)
    ref = ctx.upload_bases(0, group, np.stack([pt_limbs(0, group, p) for p in P]), np.array([p is None for p in P], dtype=np.uint8))
    a1, i1 = ctx.msm_affine(b, sc)
    a2, i2 = ctx.msm_affine(ref, sc)
    assert i1 == i2 and (a1 == a2).all()  # (Jacobian representatives may differ: compared in affine)
    b.free()
    ref.free()
    empty = ctx.upload_bases_compressed(0, group, b"", 0)
    empty.free()


@pytest.mark.parametrize("group", [1, 2])
def test_malformed_encodings_are_rejected(ctx, zk, group):
    G = po.BLS12_381.g1 if group == 1 else po.BLS12_381.g2
    good = po.bls12_381_compress(group, G.mul(G.gen, 77))
    p = po.BLS12_381.p
    # an x that is not on the curve (searched with the oracle)
    x = 5
    while True:
        cand = (bytes(47) + bytes([x])) if group == 1 else (bytes(95) + bytes([x]))
        cand = bytes([cand[0] | 0x80]) + cand[1:]
        try:
            po.bls12_381_decompress(group, cand)
            x += 1
        except ValueError:
            break
    bad = [
        bytes([good[0] & 0x7F]) + good[1:],                                   # compression flag missing
        bytes([0xC0]) + bytes(48 * group - 2) + bytes([1]),                       # infinity with a non-zero body
        bytes([0x9F]) + bytes([255]) * 47 + (good[48:] if group == 2 else b""),      # x >= p
        cand,                                                                 # not on the curve
    ]
    for enc in bad:
        assert len(enc) == 48 * group
        with pytest.raises(zk.ZkhipError):
            ctx.upload_bases_compressed(0, group, good + enc, 2)
    with pytest.raises(zk.ZkhipError):
        ctx.upload_bases_compressed(1, group, good, 1)  # BN254: no pinned wire format
