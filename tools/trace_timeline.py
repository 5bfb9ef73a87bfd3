"""Print the kernel timeline of one factorisation from a rocprofv3 kernel trace (csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
# last factorisation: find last k_diag_inverse_batched, walk back to the previous one
idx = [i for i, n in enumerate(names) if n.startswith("k_diag_inverse_batched")]
lo, hi = idx[-2] + 1, idx[-1]
t0 = int(rows[lo]["Start_Timestamp"])
first = [i for i in range(lo, hi) if names[i].startswith("k_ldlt_diag128")][0]
t0 = int(rows[first]["Start_Timestamp"])
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 80
for i in range(first, min(hi, first + nmax)):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q{r.get('Queue_Id','?'):>3} grid {r['Grid_Size_X']:>8} {names[i][:28]}")
print("total us", (int(rows[hi]["End_Timestamp"]) - t0) / 1e3)
