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
SOURCES = ["gemm_f64.hip", "ldlt.hip", "diag.hip", "cones.hip", "sdp.hip", "sdp_large.hip", "vecops.hip", "assemble.hip", "api.hip", "driver.hip", "lockstep.hip", "batch.hip"]
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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
