#!/usr/bin/env python3
"""Static VALU instruction mix of the dominant kernels, from the ISA hipcc emits for the shipped sources.

bench.py prices the VALU issue roofline with it: a wave-instruction of the 64-bit-encoded classes (VOP3:
v_mad_u64_u32, v_mul_lo_u32, 64-bit shifts and adds, three-operand adds) holds its SIMD for 4 cycles, a 32-bit-encoded VOP2 /
VOP1 instruction for 2 when another wave is there to take the next slot (4 for a lone wave) -- measured by tools/microbench2.hip
and tools/mulbench4.hip (profiles/r04_microbench2_valu_wallclock.txt, profiles/r04_mulbench4_asm_vs_cpp_13x30.txt).

Usage (dev container, no GPU needed):  python3 tools/isa_mix.py > profiles/isa_mix.json
The output is STAMPED with a hash of the kernel sources it was compiled from (`source_sha256`, = sources_hash() below); bench.py ignores a
mix whose stamp is not the tree's (a rebuilt kernel would otherwise be priced with a stale mix: ADVICE r4), and tests/test_abi.py
fails when the committed file is stale.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {"msm_bucket_acc": ("csrc/msm_bls_g1.hip", r"msm_bucket_acc_lds.*BlsFqU.*Li64ELi3ELi1"),
       "ntt_pass": ("csrc/ntt.hip", r"ntt_passIN5zkhip6BlsFrUELi2")}
VOP2 = re.compile(r"^(v_(add|sub|subrev|and|or|xor|lshlrev|lshrrev|ashrrev|mov|cndmask|min|max|addc|subb|subbrev|add_co|sub_co|not|bfrev)_[a-z0-9_]*?)(_e32|_dpp|_sdwa)?$")


def sources_hash():
    """sha256 over what the two kernels are built from: every header of crypto3-zk_amd/csrc/ and the two translation units
    (names and contents, sorted)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "crypto3-zk_amd", "csrc")
    units = {os.path.basename(src) for src, _ in SRC.values()}
    for name in sorted(os.listdir(d)):
        if name.endswith(".hpp") or name in units:
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def classify(op):
    if not op.startswith("v_"):
        return None
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "mad64"
    if op.endswith("_e64"):
        return "vop3"
    if op.endswith("_e32") or op.endswith("_dpp"):
        return "vop2"
    m = VOP2.match(op)
    # inside the asm blocks the mnemonics carry no encoding suffix: b32 logic / shifts / adds of two operands are VOP2, the b64
    # forms and everything with three operands VOP3
    if m and "b64" not in op and "u64" not in op and "i64" not in op:
        return "vop2"
    return "vop3"


def main():
    out = {}
    for key, (src, pat) in SRC.items():
        with tempfile.TemporaryDirectory() as tmp:
            s = os.path.join(tmp, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-Wno-pass-failed",
                            os.path.join(ROOT, "crypto3-zk_amd", src), "-o", s], check=True, stderr=subprocess.DEVNULL)
            text = open(s).read().splitlines()
        start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*:", l) and re.search(pat, l))
        end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
        counts = {"mad64": 0, "vop3": 0, "vop2": 0}
        for l in text[start:end]:
            m = re.match(r"^\s+([a-z_0-9]+)", l)
            if m:
                c = classify(m.group(1))
                if c:
                    counts[c] += 1
        tot = sum(counts.values())
        out[key] = {"static_valu_instructions": tot, "counts": counts, "fractions": {k: round(v / tot, 4) for k, v in counts.items()},
                    "symbol_pattern": pat, "source": src}
    out["cycles_per_wave_instruction"] = {"mad64": 4, "vop3": 4, "vop2": 2, "vop2_lone_wave": 4,
                                          "source": "tools/microbench2.hip, tools/mulbench4.hip: profiles/r04_microbench2_valu_wallclock.txt, "
                                                    "profiles/r04_mulbench4_asm_vs_cpp_13x30.txt (a 461-instruction product block issues in 4.03 "
                                                    "shader cycles per instruction from ONE wave; VOP2 at 1.0 ns against 1.73-2.2 ns for VOP3 / "
                                                    "multiply-adds at 8 waves per SIMD)"}
    out["source_sha256"] = sources_hash()
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
