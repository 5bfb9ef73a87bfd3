#!/usr/bin/env python3
"""Extracts the NUMERIC DATA of the reference's "Miles's counterexamples" (test/testdata.jl:109-150: the literal
arrays c, b, con_cones, var_cones and the COO triplets I, J, V of A) into tests/golden/miles_problems.json.
Run in the build container (needs /root/reference); the JSON travels, the reference does not.  Only data is
extracted -- the format converter the reference's tests use (mpb_to_conicip) is restated in tests/problems.py."""
import json
import os
import re

SRC = "/root/reference/test/testdata.jl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miles_problems.json")


def parse_cones(text):
    out = []
    for m in re.finditer(r"\(:(\w+),\s*\[([^\]]*)\]\)", text):
        out.append([m.group(1), [int(x) for x in m.group(2).split(",") if x.strip()]])
    return out


def main():
    src = open(SRC).read()
    problems = {}
    for k in (1, 2, 3):
        body = src[src.index("function miles_problem_%d()" % k):]
        body = body[:body.index("\nend")]
        rec = {}
        for name in ("c", "b", "I", "J", "V"):
            m = re.search(r"^\s*%s = \[(.*)\]\s*$" % name, body, re.M)
            vals = [x for x in m.group(1).split(",") if x.strip()]
            rec[name] = [int(x) for x in vals] if name in ("I", "J") else [float(x) for x in vals]
        for name in ("con_cones", "var_cones"):
            m = re.search(r"^\s*%s = \[(.*)\]\s*$" % name, body, re.M)
            rec[name] = parse_cones(m.group(1))
        assert len(rec["I"]) == len(rec["J"]) == len(rec["V"])
        problems["miles_problem_%d" % k] = rec
    json.dump(problems, open(OUT, "w"))
    for k, v in problems.items():
        print(k, "n =", len(v["c"]), "rows =", len(v["b"]), "nnz =", len(v["V"]), [(t, len(i)) for t, i in v["con_cones"]],
              [(t, len(i)) for t, i in v["var_cones"]])


if __name__ == "__main__":
    main()
