import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_pkg():
    """The package directory is `crypto3-zk_amd` (not an identifier): import it as crypto3_zk_amd."""
    if "crypto3_zk_amd" in sys.modules:
        return sys.modules["crypto3_zk_amd"]
    pkg_dir = os.path.join(ROOT, "crypto3-zk_amd")
    spec = importlib.util.spec_from_file_location("crypto3_zk_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["crypto3_zk_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The shared objects are build artefacts (git-ignored): a fresh checkout compiles them once, as
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU; a few minutes)."""
    import subprocess

    need = {os.path.join(ROOT, "crypto3-zk_amd"): ["libzkhip.so", "libzkhip_hosttest.so"],
            os.path.join(ROOT, "oracle"): ["liboracle.so"],
            os.path.join(ROOT, "tests", "cpp"): ["libshimtest.so", "quadtest"]}
    for d, files in need.items():
        if not all(os.path.exists(os.path.join(d, f)) for f in files):
            subprocess.check_call(["make", "-C", d, "-j4"] + (["all"] if d.endswith("crypto3-zk_amd") else []))


@pytest.fixture(scope="session")
def zk():
    return load_pkg()


@pytest.fixture(scope="session")
def ctx(zk):
    """A zkhip context on device 0; the HIP extension must be present -- there is no fallback."""
    c = zk.Context(0)
    yield c
    c.close()
