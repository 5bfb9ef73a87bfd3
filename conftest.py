"""Repo-root pytest bootstrap: make the product package (``conicip.jl_amd/cipkkt``)
importable as ``cipkkt`` and the oracle importable as ``oracle``."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "conicip.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
