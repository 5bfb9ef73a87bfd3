"""Builds libcipkkt.so (HIP, gfx950) in-tree: conicip.jl_amd/cipkkt/libcipkkt.so.

hipcc cross-compiles without a GPU, so this runs in the GPU-less build container;
the resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "cipkkt", "libcipkkt.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["gemm_f64.hip", "ldlt.hip", "solve.hip", "diag.hip", "cones.hip", "sdp.hip", "sdp_large.hip", "vecops.hip", "assemble.hip", "api.hip", "driver.hip", "lockstep.hip", "batch.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "cipkkt.h"))
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not _newer(op, [sp] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn.strip():
                    print(warn)
    if force or jobs or not os.path.exists(OUT):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"])
    return OUT


def asan_host(verbose=True):
    """Host-side sanitizer build ("host -fsanitize=address build"; CPU only -- GPU sanitizers are not
    available on the pool).  Compiles the CPU reference of the C ABI (oracle/cpu_ref/cipkkt_cpu.cpp: test infrastructure)
    and the plain-C client of the ABI (tests/c_abi/solve_qp.c -DCIP_PLUGIN_LEVELS_ONLY) with -fsanitize=address,undefined,
    runs the C program against the instrumented library, and runs tests/test_cpu_ref.py with the instrumented library
    loaded into Python (LD_PRELOAD of the sanitizer runtime).  Any sanitizer report is a non-zero exit.
    Returns 0 on success."""
    import shutil
    root = os.path.abspath(os.path.join(HERE, ".."))
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
    if "--asan-host" in sys.argv:
        sys.exit(asan_host())
    print(build(force="--force" in sys.argv, verbose=True))
