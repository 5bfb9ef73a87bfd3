"""TEST INFRASTRUCTURE (not product code): host sanitizer build of the C-ABI boundary's host side.

    python oracle/cpu_ref/build_asan.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def asan_host(verbose=True):
    """Host-side sanitizer build ("host -fsanitize=address build"; CPU only -- GPU sanitizers are not
    available on the pool).  Compiles the CPU reference of the C ABI (oracle/cpu_ref/cipkkt_cpu.cpp: test infrastructure)
    and the plain-C client of the ABI (tests/c_abi/solve_qp.c -DCIP_PLUGIN_LEVELS_ONLY) with -fsanitize=address,undefined,
    runs the C program against the instrumented library, and runs tests/test_cpu_ref.py with the instrumented library
    loaded into Python (LD_PRELOAD of the sanitizer runtime).  Any sanitizer report is a non-zero exit.
    Returns 0 on success."""
    import shutil
    root = os.path.abspath(os.path.join(HERE, "..", ".."))
    gxx, gcc = shutil.which("g++"), shutil.which("gcc")
    if not gxx or not gcc:
        raise RuntimeError("g++ / gcc not available")
    out = os.path.join(root, "oracle", "cpu_ref", "_build")
    os.makedirs(out, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    so = os.path.join(out, "libcipkkt_cpu_asan.so")
    exe = os.path.join(out, "solve_qp_asan")

    def run(cmd, **kw):
        if verbose:
            print(" ".join(cmd), flush=True)
        return subprocess.run(cmd, **kw)

    run([gxx, "-std=c++17", "-shared", "-fPIC"] + san + ["-o", so, os.path.join(root, "oracle", "cpu_ref", "cipkkt_cpu.cpp")], check=True)
    run([gcc, "-std=c99", "-DCIP_PLUGIN_LEVELS_ONLY"] + san + ["-I", os.path.join(root, "include"),
         os.path.join(root, "tests", "c_abi", "solve_qp.c"), so, "-lm", "-Wl,-rpath," + out, "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = run([exe], env=env, capture_output=True, text=True)
    if verbose:
        print(r.stdout[-2000:], r.stderr[-4000:])
    if r.returncode != 0:
        return r.returncode or 1
    # the same library under the Python tests of the ABI contract: the sanitizer runtime must be first in the process
    rt = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ub = subprocess.run([gcc, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=":".join(x for x in (rt, ub) if os.path.isabs(x)), CIP_CPU_REF_SO=so,
               ASAN_OPTIONS="detect_leaks=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_cpu_ref.py"), "-x", "-q", "-p", "no:cacheprovider"],
            env=env, capture_output=True, text=True, cwd=root)
    if verbose:
        print(r.stdout[-3000:], r.stderr[-3000:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(asan_host())
