#!/usr/bin/env python3
"""What ONE member of a device group of 1 / 2 / 4 / 8 GPUs has to do for the two commitment schemes' commit(batch), measured on one GPU through the
single-context scheme classes (GPU box): python3 tools/group_member_emulation.py > profiles/rNN_group_member_emulation.json

KZG (cfg 5: 50 columns x 2^20): a member commits ceil(50 / M) columns from host memory against its replica of the key -- exactly a single-context commit of
that many columns.  LPC (16 polynomials x 2^20, D[0] = 2^21): a member uploads and extends 16 / M polynomials, lays out 16 polynomials x D / M leaf positions
(the same element count as 16 / M polynomials x D) and sends 1 / M of the leaves to the host -- a single-context commit of 16 / M polynomials moves the same
bytes through the same kernels; what it leaves out is the device-to-device exchange ((M - 1) / M of the member's extensions out, as much in).
Predictions of the group's critical path, not measurements of a group: no multi-GPU box was available to any round."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import bench

lib = bench._bench_lib()
log_n = 20
out = {"what": __doc__.split("\n\n")[0], "kzg_commit_ms": {}, "kzg_proof_eval_single_ms": None, "lpc_commit_ms": {}}
data = bench.random_scalars(np, (1 << log_n) * 50, 5).reshape(50, 1 << log_n, 4)
for members in (1, 2, 4, 8):
    cols = -(-50 // members)
    steps = 4
    ms = np.zeros(3 * steps, dtype=np.float64)
    rc = lib.zkhip_bench_kzg_scheme(0, ctypes.c_size_t(log_n), ctypes.c_size_t(cols), steps, 2, ctypes.c_size_t(10), data.ctypes.data_as(ctypes.c_void_p),
                                    ms.ctypes.data_as(ctypes.c_void_p), None)
    assert rc == 0
    m = ms.reshape(steps, 3)
    out["kzg_commit_ms"][str(members)] = {"columns_per_member": cols, "commit_ms": round(float(np.median(m[1:, 1])), 2), "proof_eval_ms_of_these_columns": round(float(np.median(m[1:, 2])), 2)}
    if members == 1:
        out["kzg_proof_eval_single_ms"] = round(float(np.median(m[1:, 2])), 2)
for members in (1, 2, 4, 8):
    cols = 16 // members
    steps = 4
    ms = np.zeros(steps, dtype=np.float64)
    root = ctypes.c_uint64(0)
    rc = lib.zkhip_bench_lpc_scheme(0, ctypes.c_size_t(log_n), ctypes.c_size_t(cols), ctypes.c_size_t(1), steps, 1, 16, ms.ctypes.data_as(ctypes.c_void_p), ctypes.byref(root))
    assert rc == 0
    exch_mb = cols * (2 << log_n) * 32 * (members - 1) / members / 1e6
    out["lpc_commit_ms"][str(members)] = {"polynomials_per_member": cols, "commit_ms": round(float(np.median(ms[1:])), 2), "leaf_bytes_per_member": cols * (2 << log_n) * 32,
                                          "exchange_out_MB_not_included": round(exch_mb, 1)}
print(json.dumps(out, indent=1))
