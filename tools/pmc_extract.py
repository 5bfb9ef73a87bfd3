"""Reduce a rocprofv3 counter_collection.csv to the rows of the LDL' kernels and of the solves' gemv (one line per dispatch and counter).
usage: pmc_extract.py <counter_collection.csv> <out.csv>"""
import csv, sys
KEEP = ("k_ldlt_trailing_64", "k_gemm_nt_64", "k_ldlt_panel", "k_ldlt_diag128_v2", "k_ldlt_diag_upd", "k_trsm_subst", "k_gemv_t")
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Counter_Name", "Counter_Value", "Duration_ns"])
    for r in rows:
        name = r["Kernel_Name"].replace("void ", "")
        if not name.startswith(KEEP):
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) if "End_Timestamp" in r and r["End_Timestamp"] else ""
        w.writerow([r["Dispatch_Id"], name, r["Grid_Size"], r["Counter_Name"], r["Counter_Value"], dur])
