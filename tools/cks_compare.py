"""Compares the CIP_LG_CHECKSUM tables of repeated runs (stderr of tools/sdp640_repeat.py): first differing (call, stage) per run."""
import sys, collections
runs, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("cks call 0:") and cur: runs.append(cur); cur = []
    if line.startswith("cks call"): cur.append(line.split(":", 1)[1].split("|")[0].split())
if cur: runs.append(cur)
ref = runs[0]
print(len(runs), "runs,", len(ref), "calls each")
for k, r in enumerate(runs[1:], 1):
    d = [(c, q) for c in range(min(len(ref), len(r))) for q in range(len(ref[c])) if ref[c][q] != r[c][q]]
    if d or len(r) != len(ref): print("run", k, "calls", len(r), "first differences (call, stage):", d[:6])
