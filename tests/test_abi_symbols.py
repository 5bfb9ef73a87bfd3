"""CPU-side checks of the drop-in boundary: the C-ABI library is built, loads, and
exports every symbol include/cipkkt.h declares; the ctypes table covers them all.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cipkkt.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(cip_[A-Za-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_plugin_levels():
    fns = header_functions()
    for must in ("cip_create", "cip_set_scaling_packed", "cip_factor", "cip_solve3x3", "cip_destroy"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    import cipkkt
    from cipkkt import _lib
    assert os.path.exists(_lib.LIB_PATH), "libcipkkt.so is not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [f for f in header_functions() if not hasattr(lib, f)]
    assert not missing, "declared in include/cipkkt.h but not exported: %s" % missing


def test_ctypes_table_matches_header():
    from cipkkt import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_no_cpu_fallback_without_gpu():
    import numpy as np
    import torch
    import cipkkt
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        cipkkt.KKTSystem(np.eye(3), np.eye(3), None, [("R", 3)])


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "conicip.jl_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt:
                    bad.append(f)
    assert not bad, "product files reference the oracle: %s" % bad


def test_driver_argument_checks_need_no_gpu():
    """bad input raises before touching the GPU (test/runtests.jl:507-523)."""
    import numpy as np
    import scipy.sparse as sp
    import cipkkt
    n = 10
    with pytest.raises(Exception):
        cipkkt.conicIP(np.zeros((n, n)), np.arange(1.0, n + 1), sp.identity(n + 2, format="csr"),
                       np.zeros(n), [("R", n)])
