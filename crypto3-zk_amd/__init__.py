"""crypto3-zk_amd -- MI355X-native MSM / NTT proving backend for NilFoundation/crypto3-zk.

The product is the C-ABI shared library `libzkhip.so` (HIP kernels for gfx950, declared in
include/zkhip.h) plus the header-only C++ shim under `include/` that mirrors the reference's call sites.
This Python package is only plumbing for tests and bench.py: a ctypes view of the C ABI.  It has NO CPU
fallback -- if the library or a GPU is missing, calls raise.
"""
from .zkhip import (  # noqa: F401
    BLS12_381,
    BN254,
    G1,
    G2,
    Bases,
    Context,
    DeviceGroup,
    GroupBases,
    R1CS,
    ZkhipError,
    build,
    coord_limbs,
    lib_path,
    load_library,
)
