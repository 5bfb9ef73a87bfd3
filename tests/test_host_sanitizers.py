"""Host sanitizer build (VERDICT r2 missing 4): the host-side pieces of the boundary -- the CPU reference
of the C ABI (oracle/cpu_ref/cipkkt_cpu.cpp) and the plain-C client (tests/c_abi/solve_qp.c) -- compiled with
-fsanitize=address,undefined and exercised by the C program and by tests/test_cpu_ref.py with the instrumented library
loaded (`python oracle/cpu_ref/build_asan.py`).  CPU only: GPU sanitizers are not available on the pool."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_asan():
    gcc = shutil.which("gcc")
    if not gcc or not shutil.which("g++"):
        return False
    p = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.isabs(p) and os.path.exists(p)


@pytest.mark.skipif(not _have_asan(), reason="gcc / libasan not available")
def test_host_side_is_clean_under_asan_and_ubsan():
    spec = importlib.util.spec_from_file_location("cip_build_asan", os.path.join(ROOT, "oracle", "cpu_ref", "build_asan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.asan_host(verbose=False) == 0


# ---------------------------------------------------------------- round 6: the PRODUCT's own host code (round-5 review, weak 7)
def _have_hostsan_toolchain():
    rt = "/opt/rocm/lib/llvm/lib/clang"
    return os.path.exists("/opt/rocm/bin/hipcc") and os.path.isdir(rt) and any(
        os.path.exists(os.path.join(rt, v, "lib", "linux", "libclang_rt.asan-x86_64.a")) for v in os.listdir(rt))


def _hostsan(kind):
    spec = importlib.util.spec_from_file_location("cip_build_hostsan", os.path.join(ROOT, "tests", "hostsan", "build_hostsan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.run(kind, verbose=False, threads=3)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    import re
    m = re.search(r"drive: (\d+) launches \((\d+) emulated\), (\d+) device allocations / (\d+) bytes still live", r.stdout)
    assert m, r.stdout[-1500:]
    launches, emulated, live, live_bytes = map(int, m.groups())
    # tens of thousands of launch sequences went through the host code; what is still allocated at the end of main() is the main
    # thread's own 4-KB host scratch (a thread_local, returned at thread exit) and nothing else: no handle, arena or pool leaked
    assert launches > 10000 and emulated > 100 and live <= 1 and live_bytes <= 4096, (launches, emulated, live, live_bytes)


@pytest.mark.skipif(not _have_hostsan_toolchain(), reason="hipcc / clang sanitizer runtimes not available")
def test_product_host_control_plane_is_clean_under_asan_and_ubsan():
    """conicip.jl_amd/csrc/*.hip compiled host-only (`hipcc --cuda-host-only`) with -fsanitize=address,undefined, linked against the
    fake HIP runtime of tests/hostsan/fake_hip.cpp ("device" memory = host memory that reads zero, launches = no-ops) and driven
    through the C ABI by tests/hostsan/drive.cpp: plugin levels (dense / CSR A, p = 0 / > 0, both routes, a chip-wide S cone), the native
    loop, cip_conicip_mixed with lock-step groups of 64 and ONE, mixed nnz, a bin of one, the thread pool, a batch refused at level 1,
    the stand-alone LDL' with a caller-owned workspace of exactly the advertised size in both solve modes, three concurrent callers.
    No sanitizer report, no leaked "device" allocation."""
    _hostsan("asan")


@pytest.mark.skipif(not _have_hostsan_toolchain(), reason="hipcc / clang sanitizer runtimes not available")
def test_product_host_control_plane_is_clean_under_tsan():
    """the same build and driver under -fsanitize=thread: the thread pools of the batch entry points, the thread-local batch contexts,
    the cached lock-step arena and the process-wide knobs under three concurrent callers"""
    _hostsan("tsan")
