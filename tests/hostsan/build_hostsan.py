"""TEST INFRASTRUCTURE (not product code): sanitizer builds of the PRODUCT's host control plane, CPU only.

    python tests/hostsan/build_hostsan.py [asan|tsan]

Every translation unit of conicip.jl_amd/csrc is compiled with `hipcc --cuda-host-only` (its host half: handle and arena
management, the lock-step driver, the thread pools, the native interior-point loop, the launch sequences) under
-fsanitize=address,undefined or -fsanitize=thread, and linked -- instead of libamdhip64 -- against tests/hostsan/fake_hip.cpp, a fake
HIP runtime whose "device" memory is host memory and whose kernel launches are no-ops.  tests/hostsan/drive.cpp then drives the C ABI
through it (plugin levels, cip_conicip, cip_conicip_mixed with groups of 1 and 65, mixed nnz, refused S cones, concurrent callers).
GPU sanitizers are not available on the pool; the kernels themselves are covered by the parity tests on the GPU.
Outputs go to tests/hostsan/_build (git-ignored)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "conicip.jl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANGXX = os.environ.get("CIP_CLANGXX", "/opt/rocm/lib/llvm/bin/clang++")


def _sources():
    import importlib.util                                        # the product's own source list
    spec = importlib.util.spec_from_file_location("cipkkt_build", os.path.join(ROOT, "conicip.jl_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SOURCES


def build(kind="asan", verbose=False):
    """Returns (driver executable, environment for running it)."""
    out = os.path.join(HERE, "_build", kind)
    os.makedirs(out, exist_ok=True)
    san = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "tsan": ["-fsanitize=thread"]}[kind]
    common = ["-O1", "-g", "-std=c++17", "-fPIC", "-fno-omit-frame-pointer"] + san

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("command failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    objs = []
    for src in _sources():
        o = os.path.join(out, src.replace(".hip", ".o"))
        sp = os.path.join(CSRC, src)
        deps = [sp] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "cipkkt.h")]
        if not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in deps):
            run([HIPCC, "--cuda-host-only", "--offload-arch=gfx950", "-Wno-unused-function", "-w"] + common + ["-c", sp, "-o", o])
        objs.append(o)
    # every host-only translation unit refers to its (absent) device image by a hashed symbol: give each one a few dummy bytes
    undef = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
    fat = sorted(set(re.findall(r"__hip_fatbin_[0-9a-f]+", undef)))
    stub = os.path.join(out, "fatbin_stubs.c")
    with open(stub, "w") as f:
        for s in fat:
            f.write("const char %s[64] = {0};\n" % s)
    fake = os.path.join(out, "fake_hip.o")
    run([CLANGXX, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + common + ["-c", os.path.join(HERE, "fake_hip.cpp"), "-o", fake])
    stubo = os.path.join(out, "fatbin_stubs.o")
    run([CLANGXX.replace("clang++", "clang"), "-fPIC", "-c", stub, "-o", stubo])
    so = os.path.join(out, "libcipkkt_host_%s.so" % kind)
    run([CLANGXX, "-shared"] + san + ["-o", so] + objs + [fake, stubo, "-ldl", "-lpthread"])
    exe = os.path.join(out, "drive_%s" % kind)
    run([CLANGXX, "-I", os.path.join(ROOT, "include")] + common + [os.path.join(HERE, "drive.cpp"), so, "-Wl,-rpath," + out, "-lpthread", "-o", exe])
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    if kind == "asan":
        env.update(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23:detect_stack_use_after_return=1",
                   UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    else:
        env.update(TSAN_OPTIONS="halt_on_error=1:exitcode=24:second_deadlock_stack=1")
    return exe, env


def run(kind="asan", verbose=True, threads=3):
    exe, env = build(kind, verbose=verbose)
    r = subprocess.run([exe, str(threads)], env=env, capture_output=True, text=True, timeout=1500)
    if verbose:
        print(r.stdout[-3000:], r.stderr[-6000:])
    return r


if __name__ == "__main__":
    kinds = sys.argv[1:] or ["asan", "tsan"]
    rc = 0
    for k in kinds:
        rc = rc or run(k).returncode
    sys.exit(rc)
