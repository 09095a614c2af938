"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/zkhip.h
declares, and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(zk):
    lib = zk.load_library()
    names = _declared()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/zkhip.h but not exported by libzkhip.so"
    assert sorted(zk.zkhip.EXPORTS) == names


def test_no_cpu_fallback(zk):
    """Without a HIP device zkhip_init must fail loudly; with one it must succeed."""
    lib = zk.load_library()
    h = ctypes.c_void_p()
    rc = lib.zkhip_init(0, ctypes.byref(h))
    if rc == 0:
        lib.zkhip_destroy(h)
        pytest.skip("a GPU is present: covered by the -m gpu tests")
    assert rc == -1  # ZKHIP_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.zkhip_strerror(rc)
    with pytest.raises(zk.ZkhipError):
        zk.Context(0)


def test_device_group_has_no_cpu_fallback_either(zk):
    """zkhip_group_init without a HIP device: ZKHIP_ERR_NO_DEVICE, no handle -- the group is built from ordinary contexts and inherits their rule"""
    lib = zk.load_library()
    h = ctypes.c_void_p()
    devs = (ctypes.c_int * 2)(0, 0)
    rc = lib.zkhip_group_init(devs, 2, ctypes.byref(h))
    if rc == 0:
        lib.zkhip_group_destroy(h)
        pytest.skip("a GPU is present: covered by tests/test_gpu_group.py")
    assert rc == -1 and not h.value
    assert lib.zkhip_group_init(devs, 0, ctypes.byref(h)) == -2      # no devices: invalid, whatever the box
    with pytest.raises(zk.ZkhipError):
        zk.DeviceGroup([0, 0])


def test_product_does_not_reference_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "crypto3-zk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "pyoracle" not in text and "cport" not in text, os.path.join(dirpath, f)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libzkhip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_generated_asm_products_are_current(tmp_path):
    """crypto3-zk_amd/csrc/mont_asm.hpp (the device's Montgomery products as inline-asm blocks) is GENERATED from
    tools/gen_mont_asm.py; the committed file must be what the generator writes.  The blocks restate fu.hpp's C++ bodies term by
    term (the -m gpu suite holds them against the oracle; the host tests run the C++ bodies)."""
    import subprocess
    import sys
    out = tmp_path / "mont_asm.hpp"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_mont_asm.py"), "--product", str(out)], check=True)
    assert out.read_text() == open(os.path.join(ROOT, "crypto3-zk_amd", "csrc", "mont_asm.hpp")).read()
    text = out.read_text()
    # 14-limb product: 2 L^2 multiply-adds, L mul_lo, 2 L masks, 2 L - 1 shifts, one move
    body = text.split("struct MontAsm<14>")[1].split("static __device__")[1]
    assert body.count("v_mad_u64_u32") == 2 * 14 * 14 and body.count("v_lshrrev_b64") == 27 and body.count("v_mul_lo_u32") == 14


def test_isa_mix_is_current():
    """profiles/isa_mix.json (the static instruction mix bench.py prices the VALU roofline with) carries the hash of the kernel
    sources it was compiled from; a kernel edit without `python3 tools/isa_mix.py > profiles/isa_mix.json` would price the new
    kernel with the old mix -- bench.py then drops the priced figure, and this test says why."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("zk_isa_mix", os.path.join(ROOT, "tools", "isa_mix.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mix = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
    assert mix["source_sha256"] == mod.sources_hash(), "regenerate: python3 tools/isa_mix.py > profiles/isa_mix.json"
    for k in ("msm_bucket_acc", "ntt_pass"):
        assert abs(sum(mix[k]["fractions"].values()) - 1) < 1e-3
