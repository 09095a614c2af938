#!/usr/bin/env python3
"""Two Groth16 provers on two host threads (a context and a device key each) sharing one GPU: aggregate proofs/s against
one prover alone.  Throughput arrangement of INTEGRATION.md; every figure in DESIGN.md is one proof at a time."""
import ctypes
import json
import os
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, G = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 7


def limbs(v):
    return np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def run(lib, M, steps, out, k, barrier):
    m = 1
    while m < M + 11:
        m <<= 1
    omega, coset = limbs(pow(G, (R - 1) // m, R)), limbs(G)
    times = np.zeros(steps, dtype=np.float64)
    setup, verified = ctypes.c_double(), ctypes.c_int(-1)
    prof = ctypes.create_string_buffer(16384)
    t0 = time.perf_counter()
    rc = lib.zkhip_bench_groth16(0, 0, ctypes.c_size_t(M), ctypes.c_size_t(10), ctypes.c_uint64(1 + k), steps, omega.ctypes.data_as(ctypes.c_void_p),
                                 coset.ctypes.data_as(ctypes.c_void_p), times.ctypes.data_as(ctypes.c_void_p), ctypes.byref(setup), ctypes.byref(verified),
                                 prof, ctypes.c_size_t(16384))
    out[k] = {"rc": rc, "verified": verified.value == 1, "ms_per_proof": [round(float(t), 2) for t in times], "setup_ms": round(setup.value, 1),
              "wall_s": round(time.perf_counter() - t0, 3)}


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "crypto3-zk_amd", "libzkhip_bench.so"))
    M, steps = 1 << 20, 12
    res = {}
    for nthreads in (1, 2):
        out = [None] * nthreads
        barrier = threading.Barrier(nthreads)
        hook = ctypes.CFUNCTYPE(None)(lambda: barrier.wait())  # every prover has its key before any of them starts proving
        lib.zkhip_bench_set_after_setup(hook)
        th = [threading.Thread(target=run, args=(lib, M, steps, out, k, barrier)) for k in range(nthreads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        # steady state: proofs 3 .. steps - 2 of every prover (all provers are proving during them)
        per = [sum(o["ms_per_proof"][2:-2]) / len(o["ms_per_proof"][2:-2]) for o in out]
        res["%d prover(s)" % nthreads] = {"mean_ms_per_proof_per_prover": [round(p, 2) for p in per],
                                          "proofs_per_s": round(sum(1e3 / p for p in per), 2), "verified": all(o["verified"] for o in out)}
    print(json.dumps({"workload": "Groth16 2^20 constraints BLS12-381, one MI355X, provers on separate host threads / contexts / device keys", **res}))


if __name__ == "__main__":
    main()
