"""The reference-side binding (integration/ConicIPHIP: a Julia package) cannot be executed in this image --
no Julia -- so it is checked mechanically against the C ABI it binds: every `ccall` names a function declared in
include/cipkkt.h with the same number of arguments, the Julia `CipProblem` struct mirrors `cip_problem` field for field,
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


def test_every_ccall_matches_the_header():
    protos = _c_prototypes()
    calls = _julia_ccalls()
    assert {c[0] for c in calls} >= {"cip_create", "cip_create_ex", "cip_set_scaling_packed", "cip_factor", "cip_solve3x3",
                                     "cip_solve2x2", "cip_destroy", "cip_last_error"}
    for name, nargs in calls:
        assert name in protos, "%s is not declared in include/cipkkt.h" % name
        assert protos[name] == nargs, "%s: %d argument types in the ccall, %d parameters in the header" % (name, nargs, protos[name])


def test_julia_struct_mirrors_cip_problem():
    h = _header()
    body = re.search(r"typedef struct cip_problem \{(.*?)\} cip_problem;", h, flags=re.S).group(1)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        base_is_ptr = "*" in decl
        for piece in decl.split(","):
            name = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", piece.strip())[0]
            c_fields.append((name, "*" in piece or (base_is_ptr and piece is decl)))
    c_names = [f[0] for f in c_fields]
    src = open(JL).read()
    jbody = re.search(r"struct CipProblem[^\n]*\n(.*?)\nend", src, flags=re.S).group(1)
    j_fields = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)::(Ptr\{\w+\}|Cint)", jbody)
    assert [f[0] for f in j_fields] == c_names, ([f[0] for f in j_fields], c_names)
    for (jn, jt), (cn, _) in zip(j_fields, c_fields):
        decl = re.search(r"([^;{]*\b%s\b)" % cn, body).group(1)
        is_ptr = "*" in decl.split(cn)[0].split(",")[-1] or re.search(r"\*\s*%s\b" % cn, decl) is not None
        assert jt.startswith("Ptr") == bool(is_ptr), (jn, jt, decl)


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
