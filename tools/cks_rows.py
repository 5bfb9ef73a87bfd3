"""Rows of a CIP_LG_CHECKSUM table that differ from row 0 (every call had the same input): which stages, per call."""
import sys
lines = [l for l in open(sys.argv[1]) if l.startswith("cks call")]
rows = [l.split(":", 1)[1].split("|")[0].split() for l in lines]
print(len(rows), "calls")
for c, r in enumerate(rows[1:], 1):
    d = [q for q in range(len(r)) if r[q] != rows[0][q]]
    if d: print("call", c, "stages that differ:", d, "|", lines[c].split("|", 1)[1].strip() if "|" in lines[c] else "")
print("call 0 |", lines[0].split("|", 1)[1].strip() if "|" in lines[0] else "")
