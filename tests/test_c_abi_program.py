"""The drop-in boundary used from plain C: tests/c_abi/solve_qp.c is compiled with gcc against include/cipkkt.h and
the in-tree libcipkkt.so only (no Python, torch or HIP headers on that side) and run on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "conicip.jl_amd", "cipkkt")


def _compile(tmp_path):
    exe = str(tmp_path / "solve_qp")
    cmd = ["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "solve_qp.c"),
           "-L", LIBDIR, "-lcipkkt", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_program_builds_against_the_header(tmp_path):
    """CPU: the header is valid C99 and every symbol the program uses links against the built library."""
    if not os.path.exists(os.path.join(LIBDIR, "libcipkkt.so")):
        pytest.skip("libcipkkt.so not built")
    assert os.path.exists(_compile(tmp_path))


@pytest.mark.gpu
def test_c_program_runs(tmp_path):
    exe = _compile(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "conicip status 1" in r.stdout
