#!/usr/bin/env python3
"""Groth16 prover throughput through the C++ shim (r1cs_gg_ppzksnark_prover_hip::process): constraints/s on one GPU.
BASELINE config 4's single-GPU leg (M = 2^20, n = 10 -> m = 2^21)."""
import argparse
import ctypes
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = {0: (0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 7),
     1: (21888242871839275222246405745257275088548364400416034343698204186575808495617, 5)}


def limbs(v):
    return np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-constraints", type=int, default=20)
    ap.add_argument("--inputs", type=int, default=10)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--curve", type=int, default=0)
    ap.add_argument("--domain", choices=("ref", "basic"), default="ref",
                    help="ref: the domain make_evaluation_domain(M + n + 1) picks (step radix-2 for M = 2^20, n = 10); basic: the next power of two")
    ap.add_argument("--lanes", type=int, default=1, help="> 1: afterwards that many provers at once over the same resident key (throughput arrangement)")
    a = ap.parse_args()
    so = os.path.join(ROOT, "crypto3-zk_amd", "libzkhip_bench.so")
    if not os.path.exists(so):  # never build from here: this script runs under rocprofv3 (no compiler driver from a profiled process)
        raise SystemExit("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first" % so)
    lib = ctypes.CDLL(so)
    M = 1 << a.log_constraints
    m = 1
    while m < M + a.inputs + 1:
        m <<= 1
    lib.zkhip_bench_set_domain(0 if a.domain == "basic" else -1, ctypes.c_size_t(m if a.domain == "basic" else 0))
    lib.zkhip_bench_set_lanes(a.lanes)
    r, g = R[a.curve]
    omega = limbs(pow(g, (r - 1) // m, r))
    coset = limbs(g)
    times = np.zeros(a.steps, dtype=np.float64)
    setup = ctypes.c_double()
    prof = ctypes.create_string_buffer(16384)
    verified = ctypes.c_int(-1)
    rc = lib.zkhip_bench_groth16(0, a.curve, ctypes.c_size_t(M), ctypes.c_size_t(a.inputs), ctypes.c_uint64(1), a.steps,
                                 omega.ctypes.data_as(ctypes.c_void_p), coset.ctypes.data_as(ctypes.c_void_p),
                                 times.ctypes.data_as(ctypes.c_void_p), ctypes.byref(setup), ctypes.byref(verified), prof, ctypes.c_size_t(16384))
    assert rc == 0, rc
    info = np.zeros(8, dtype=np.uint64)
    lib.zkhip_bench_last_info(info.ctypes.data_as(ctypes.c_void_p))
    lane_info = np.zeros(4, dtype=np.float64)
    lib.zkhip_bench_last_lanes(lane_info.ctypes.data_as(ctypes.c_void_p))
    kern = {l.rsplit(" ", 2)[0]: round(float(l.rsplit(" ", 2)[1]), 3) for l in prof.value.decode().splitlines() if l}
    best = float(times[1:].min() if a.steps > 1 else times.min())
    print(json.dumps({"workload": "Groth16 prove, curve %d, 2^%d constraints, %d inputs, %s domain of %d points, 1 GPU, via C++ shim (H2D of the assignment and D2H of the 5 MSM results included)"
                      % (a.curve, a.log_constraints, a.inputs, ("basic", "extended", "step")[int(info[0])] + " radix-2", int(info[1])),
                      "ms_per_proof": [round(float(t), 3) for t in times], "constraints_per_s": round(M / best * 1e3, 1),
                      "setup_ms": round(setup.value, 1), "verified": verified.value == 1,
                      **({"lanes_over_one_key": {"lanes": int(lane_info[0]), "proofs_per_s": round(float(lane_info[1]), 2),
                                                 "ms_per_proof_seen_by_a_lane": round(float(lane_info[2]), 2), "all_equal": bool(lane_info[3] == 1)}}
                         if lane_info[0] > 1 else {}),
                      "kernel_ms_last_proof (main stream; the G2 multiexp runs on a second context)": kern}))


if __name__ == "__main__":
    main()
