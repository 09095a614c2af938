"""The threaded host side of the header-only shim under AddressSanitizer + UBSan and under ThreadSanitizer, and a mutation loop
over proving_key_from_bytes (untrusted input) -- CPU builds against a stub backend that computes nothing
(tests/cpp/sanitize/stub_backend.cpp); GPU sanitizers are not available on this pool.  What the driver covers is listed at the
top of tests/cpp/sanitize/sanitize_main.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build(target):
    # both drivers in one make: on a fresh checkout the two ~2-minute builds run side by side (the second test finds its binary up to date)
    subprocess.check_call(["make", "-j2", "-C", CPP, "sanitize/sanitize_asan", "sanitize/sanitize_tsan"], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, target)


def _run(exe, what, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([exe, what], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, text=True, env=env)
    return out.returncode, out.stdout


@pytest.mark.parametrize("what", ["threads", "fuzz"])
def test_host_shim_under_asan_ubsan(what):
    rc, log = _run(_build("sanitize/sanitize_asan"), what, {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert rc == 0 and "sanitize_main: ok" in log, log[-4000:]
    assert "AddressSanitizer" not in log and "runtime error" not in log and "LeakSanitizer" not in log, log[-4000:]
    if what == "fuzz":
        assert "blobs accepted" in log
    else:
        # the driver is built with ZK_PLACEHOLDER_PROFILING_ENABLED: the one reference profiler region that lies inside replaced code
        # (basic_fri.hpp:449) reports under the reference's name and format
        assert "Basic FRI Precommit time: " in log and " ms" in log


def test_host_shim_under_tsan():
    rc, log = _run(_build("sanitize/sanitize_tsan"), "threads", {"TSAN_OPTIONS": "halt_on_error=0"})
    assert rc == 0 and "sanitize_main: ok" in log, log[-4000:]
    assert "ThreadSanitizer" not in log, log[-4000:]
