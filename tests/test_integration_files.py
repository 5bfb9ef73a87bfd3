"""The reference-side binding (integration/ConicIPHIP: a Julia package) cannot be executed in this image --
no Julia -- so it is checked mechanically against the C ABI it binds: every `ccall` names a function declared in
include/cipkkt.h with the same number of arguments, the Julia `CipProblem` / `CipOptions` / `CipResult` structs mirror `cip_problem` / `cip_options` / `cip_result` field for field (order, names, C types),
the package files are complete, and integration/moi_kktsolver.patch applies to the reference's src/MOI_wrapper.jl (when the
reference tree is present: in the build container, not on the GPU box)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "integration", "ConicIPHIP")
JL = os.path.join(PKG, "src", "ConicIPHIP.jl")
HEADER = os.path.join(ROOT, "include", "cipkkt.h")


def _header():
    src = open(HEADER).read()
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _c_prototypes():
    """name -> number of parameters, for every function declared in the header."""
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char \*)\s*(cip_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", _header(), flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def _split_top_level(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


def _julia_ccalls():
    """(symbol, number of argument TYPES in the ccall's type tuple) for every ccall of the package."""
    src = open(JL).read()
    out = []
    for m in re.finditer(r"ccall\(_sym\(:(cip_[A-Za-z0-9_]+)\),\s*(\w+),\s*\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        types = src[i:j - 1]
        out.append((m.group(1), len([t for t in _split_top_level(types) if t])))
    return out


def test_package_files_are_complete():
    toml = open(os.path.join(PKG, "Project.toml")).read()
    for key in ('name = "ConicIPHIP"', "uuid =", "ConicIP =", "Libdl =", "SparseArrays =", "LinearAlgebra ="):
        assert key in toml, key
    src = open(JL).read()
    assert re.search(r"^module ConicIPHIP", src, flags=re.M) and src.rstrip().endswith("end # module")
    for name in ("kktsolver_hip", "kktsolver_hip_full3x3", "kktsolver_2x2_hip", "__init__", "Libdl.find_library", "Libdl.dlopen"):
        assert name in src, name
    assert os.path.exists(os.path.join(PKG, "test", "runtests.jl"))
    # balanced block structure (a cheap parse check: every `function` / `struct` / `if` / `for` / `begin` / `module` closes)
    code = re.sub(r'"""(.|\n)*?"""', "", src)
    code = re.sub(r"#.*", "", code)
    code = re.sub(r'"(\\.|[^"\\])*"', '""', code)
    # block openers at the start of a statement (comprehension `for`s and ternaries do not open blocks), `x = if ...`, and
    # the `GC.@preserve ... begin` form
    opens = len(re.findall(r"^\s*(?:mutable\s+)?(function|struct|if|for|begin|module|while|let)\b", code, flags=re.M))
    opens += len(re.findall(r"=\s*if\b", code)) + len(re.findall(r"\S[^\n]*\bbegin\s*$", code, flags=re.M))
    closes = len(re.findall(r"^\s*end\b", code, flags=re.M))
    assert opens == closes, (opens, closes)


def _c_param_classes():
    """name -> ['int' | 'double' | 'ptr', ...] for every function declared in the header"""
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char \*)\s*(cip_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", _header(), flags=re.S):
        args = m.group(2).strip()
        cls = []
        for a in ([] if args in ("", "void") else args.split(",")):
            a = a.strip()
            cls.append("ptr" if "*" in a else "double" if re.match(r"(const\s+)?double\b", a) else "int")
        out[m.group(1)] = cls
    return out


def test_every_ccall_argument_has_the_headers_type_class():
    """beyond the count: an `int` parameter is bound as Cint, a `double` as Cdouble, every pointer as Ptr{..} / Ref{..}"""
    src = open(JL).read()
    classes = _c_param_classes()
    n = 0
    for m in re.finditer(r"ccall\(_sym\(:(cip_[A-Za-z0-9_]+)\),\s*(\w+),\s*\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        types = [t for t in _split_top_level(src[i:j - 1]) if t]
        got = ["int" if t == "Cint" else "double" if t in ("Cdouble", "Float64") else "ptr" if re.match(r"(Ptr|Ref)\{", t) else t
               for t in types]
        assert got == classes[m.group(1)], (m.group(1), got, classes[m.group(1)])
        n += 1
    assert n >= 10


def test_every_ccall_matches_the_header():
    protos = _c_prototypes()
    calls = _julia_ccalls()
    assert {c[0] for c in calls} >= {"cip_create", "cip_create_ex", "cip_set_scaling_packed", "cip_factor", "cip_solve3x3",
                                     "cip_solve2x2", "cip_destroy", "cip_last_error", "cip_conicip", "cip_conicip_mixed"}
    for name, nargs in calls:
        assert name in protos, "%s is not declared in include/cipkkt.h" % name
        assert protos[name] == nargs, "%s: %d argument types in the ccall, %d parameters in the header" % (name, nargs, protos[name])


def _c_struct_fields(name):
    """[(field, julia type class)] of `typedef struct <name> {...}`: 'Cint', 'Cdouble' or 'Ptr{Cint}' / 'Ptr{Float64}'."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), _header(), flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        base = re.match(r"(const\s+)?(int|double)\b", decl)
        assert base, decl
        scalar = {"int": "Cint", "double": "Cdouble"}[base.group(2)]
        elem = {"int": "Cint", "double": "Float64"}[base.group(2)]
        rest = decl[base.end():]
        for piece in rest.split(","):
            piece = piece.strip()
            fname = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", piece)[0]
            out.append((fname, "Ptr{%s}" % elem if "*" in piece else scalar))
    return out


def _julia_struct_fields(name):
    src = open(JL).read()
    jbody = re.search(r"^struct %s\b[^\n]*\n(.*?)\n^end" % name, src, flags=re.S | re.M).group(1)
    jbody = re.sub(r"#.*", "", jbody)
    return re.findall(r"([A-Za-z_][A-Za-z0-9_]*)::(Ptr\{\w+\}|Cint|Cdouble|Float64)", jbody)


@pytest.mark.parametrize("cname, jname", [("cip_problem", "CipProblem"), ("cip_options", "CipOptions"), ("cip_result", "CipResult")])
def test_julia_structs_mirror_the_c_structs(cname, jname):
    """field ORDER, NAMES and C TYPES (int <-> Cint, double <-> Cdouble, T* <-> Ptr{T}): Julia lays an immutable struct of
    such fields out as the C compiler does, so these three facts are the whole ABI of the struct"""
    c = _c_struct_fields(cname)
    j = [(n, "Cdouble" if t == "Float64" else t) for n, t in _julia_struct_fields(jname)]
    assert [f[0] for f in j] == [f[0] for f in c], (j, c)
    for (jn, jt), (cn, ct) in zip(j, c):
        assert jt == ct, "%s.%s: Julia %s, C %s" % (jname, jn, jt, ct)


def test_struct_sizes_match_the_python_binding():
    """the ctypes mirrors the GPU tests drive (cipkkt/_lib.py) have the same fields in the same order as the header: the Julia
    mirrors are then checked against the same source of truth"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "conicip.jl_amd"))
    from cipkkt import _lib as L
    for cname, ct in (("cip_problem", L.CipProblem), ("cip_options", L.CipOptions), ("cip_result", L.CipResult)):
        assert [f[0] for f in ct._fields_] == [f[0] for f in _c_struct_fields(cname)], cname


def test_native_loop_and_batch_are_bound():
    """SURVEY 8 f1 from the reference side: `conicIP_hip` binds cip_conicip, `conicIP_hip_batch` binds cip_conicip_mixed; both
    return ConicIP.Solution with the reference's keyword names and defaults (src/ConicIP.jl:498-509)"""
    src = open(JL).read()
    calls = {c[0] for c in _julia_ccalls()}
    assert {"cip_conicip", "cip_conicip_mixed"} <= calls
    sig = re.search(r"function conicIP_hip\((.*?)\)\n", src, flags=re.S).group(1)
    for kw in ("optTol = 1e-6", "DTB = 0.01", "verbose = true", "maxRefinementSteps = 3", "maxIters = 100", "cache_nestodd = false",
               "infeasTol = optTol", "refinementThreshold = optTol / 1e7", "G = spzeros(0, length(c))", "d = zeros(0)"):
        assert kw in sig, kw
    assert "ConicIP.Solution(y, w, v, _STATUS[r.status + 1]" in src
    # status codes of the header in the order of the Julia tuple
    h = _header()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define CIP_STATUS_(\w+)\s+(\d+)", h)}
    tup = re.search(r"const _STATUS = \((.*?)\)", src).group(1)
    names = [t.strip().lstrip(":").upper() for t in tup.split(",")]
    assert [codes[nm] for nm in names] == list(range(len(names))), (names, codes)
    # Diagonal / Id(n) goes over as CSR, not as a dense identity (src/ConicIP.jl:18)
    assert "_as_csr_source(A::Diagonal) = sparse(A)" in src
    for name in ("conicIP_hip", "conicIP_hip_batch", "preprocess_conicIP_hip", "CipOptions", "CipResult"):
        assert re.search(r"^export .*\b%s\b" % name, src, flags=re.M), name
    tests = open(os.path.join(PKG, "test", "runtests.jl")).read()
    for name in ("conicIP_hip(", "conicIP_hip_batch(", "preprocess_conicIP_hip(", "Id(n)"):
        assert name in tests, name


@pytest.mark.skipif(not os.path.exists("/root/reference/src/MOI_wrapper.jl") or shutil.which("patch") is None,
                    reason="reference tree / patch(1) not present (GPU box)")
def test_moi_patch_applies_to_the_reference(tmp_path):
    work = tmp_path / "ConicIP" / "src"
    work.mkdir(parents=True)
    shutil.copy("/root/reference/src/MOI_wrapper.jl", work / "MOI_wrapper.jl")
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "integration", "moi_kktsolver.patch")],
                       cwd=tmp_path / "ConicIP", capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["patch", "-p1", "-i", os.path.join(ROOT, "integration", "moi_kktsolver.patch")],
                       cwd=tmp_path / "ConicIP", capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = (work / "MOI_wrapper.jl").read_text()
    assert "kktsolver = dest.kktsolver" in out and 'attr.name == "kktsolver"' in out and "kktsolver::Any" in out
    assert "dest.sol = dest.solve(Q, c_int, A, b, cone_dims, G, d;" in out and "solve::Any" in out


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "ConicIP.jl")), reason="reference tree not present (GPU box)")
def test_every_conicip_identifier_the_julia_shim_uses_exists_in_the_reference():
    """integration/ConicIPHIP cannot be executed here (no julia in the image): a second static check beside the ccall / struct
    ones -- every name the shim takes from ConicIP is defined in the reference where the shim's comments say: `Block` with a
    `Blocks` vector (src/blockmatrices.jl:35-41), `VecCongurance` with its `R` (src/ConicIP.jl:35), `ord` (:85), `Solution` with
    the eleven fields the shim fills positionally (:384-398), `imcols` / `preprocess_conicIP` (src/preprocessor.jl), `pivot` and
    `kktsolver_2x2` (src/kktsolvers.jl), the keyword names and defaults of `conicIP` (src/ConicIP.jl:498-509)."""
    rd = lambda f: open(os.path.join(REF_SRC, f), encoding="utf-8").read()
    core, blocks, pre, kkt = rd("ConicIP.jl"), rd("blockmatrices.jl"), rd("preprocessor.jl"), rd("kktsolvers.jl")
    shim = open(os.path.join(ROOT, "integration", "ConicIPHIP", "src", "ConicIPHIP.jl"), encoding="utf-8").read()
    assert re.search(r"mutable struct Block\b", blocks) and re.search(r"Blocks::Vector\{BlockElem\}", blocks)
    assert re.search(r"mutable struct VecCongurance; R :: Matrix; end", core)
    assert re.search(r"^ord\(x\) = ", core, re.M)
    m = re.search(r"mutable struct Solution\n(.*?)\nend", core, re.S)
    fields = re.findall(r"^\s*(\w+)\s*::", m.group(1), re.M)
    assert fields == ["y", "w", "v", "status", "Iter", "Mu", "prFeas", "duFeas", "muFeas", "pobj", "dobj"]
    # the shim builds Solution positionally with exactly these eleven values, in this order
    call = re.search(r"ConicIP\.Solution\(y, w, v, _STATUS\[r\.status \+ 1\], Int\(r\.iter\), r\.mu, r\.prFeas, r\.duFeas, r\.muFeas, r\.pobj, r\.dobj\)", shim)
    assert call, "the shim's Solution constructor call changed: re-check it against src/ConicIP.jl:384-398"
    assert re.search(r"^function imcols\(A, b", pre, re.M) and re.search(r"^function preprocess_conicIP\(Q, c", pre, re.M)
    assert re.search(r"^function pivot", kkt, re.M) or re.search(r"^pivot\(", kkt, re.M) or "pivotgen" in kkt
    assert re.search(r"function kktsolver_2x2", kkt)
    # every `ConicIP.<name>` of the shim is a name the reference defines somewhere in src/
    everything = core + blocks + pre + kkt + rd("MOI_wrapper.jl")
    for name in sorted(set(re.findall(r"ConicIP\.(\w+)", shim))):
        if name in ("jl",):                                   # "ConicIP.jl:123" in comments
            continue
        assert re.search(r"\b%s\b" % re.escape(name), everything), "ConicIP.%s is not a name of the reference" % name
    # keyword names and defaults of conicIP the shim repeats (src/ConicIP.jl:498-509)
    sig = core[core.index("function conicIP("):core.index("function conicIP(") + 1500]
    for kw, default in (("optTol", "1e-6"), ("DTB", "0.01"), ("maxRefinementSteps", "3"), ("maxIters", "100")):
        assert re.search(r"%s\s*=\s*%s" % (kw, re.escape(default)), sig), kw
        assert re.search(r"%s\s*=\s*%s" % (kw, re.escape(default)), shim), kw
