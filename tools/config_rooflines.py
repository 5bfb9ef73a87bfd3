#!/usr/bin/env python3
"""Machine-readable roofline records of the secondary configurations (round-3 review, "missing" 5): BASELINE configs 3, 4 and
the per-rank shard of config 5, from the `rocprofv3 --kernel-trace --stats` summaries tools/profile_r4.sh collects.

usage: python tools/config_rooflines.py <dir with c3_kernel_stats.csv, c4_kernel_stats.csv, c5_b8_kernel_stats.csv> > rooflines.json

For every dominant kernel: bound, algorithmic work per launch (Schur formation m n^2 flop on the lower half,
trailing update r (r + 1) K, congruences 2 x 2 rp^3 per column), average launch duration from the profile, achieved rate,
peak (MI355X_MICROARCH.md: fp64 MFMA 78.6 TFLOP/s, HBM 8 TB/s) and the fraction.  Latency-bound chains (the one-sided Jacobi of
the large-S NT scaling, the panel launches) carry `bound: "latency"` with their per-round / per-launch time instead."""
import csv
import json
import os
import sys

PEAK_TF, PEAK_GBS = 78.6, 8000.0


def stats(path):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "")
        e = out.setdefault(name, dict(calls=0, total_ns=0.0))
        e["calls"] += int(r["Calls"]); e["total_ns"] += float(r["TotalDurationNs"])
    for e in out.values():
        e["avg_us"] = e["total_ns"] / max(e["calls"], 1) / 1e3
    return out


def mfma(kernel, st, flops_per_launch, what):
    if kernel not in st:
        return None
    e = st[kernel]
    tf = flops_per_launch / (e["avg_us"] * 1e-6) / 1e12
    return dict(kernel=kernel, role=what, bound="mfma", launches=e["calls"], avg_launch_us=round(e["avg_us"], 2),
                algorithmic_flops_per_launch=flops_per_launch, achieved=round(tf, 2), peak=PEAK_TF, unit="TFLOP/s", frac=round(tf / PEAK_TF, 3))


def latency(kernel, st, what, **extra):
    if kernel not in st:
        return None
    e = st[kernel]
    return dict(kernel=kernel, role=what, bound="latency", launches=e["calls"], avg_launch_us=round(e["avg_us"], 2), **extra)


def share(st):
    tot = sum(e["total_ns"] for e in st.values())
    top = sorted(st.items(), key=lambda kv: -kv[1]["total_ns"])[:8]
    return [dict(kernel=k, pct=round(100 * e["total_ns"] / tot, 1), calls=e["calls"], avg_us=round(e["avg_us"], 1)) for k, e in top]


def main(d):
    out = {}
    # ---- config 3: SOCP n = 4096, 512 x Q(8) (m = 4096), p = 512, dense A -> Schur order 4608
    st = stats(os.path.join(d, "c3_kernel_stats.csv"))
    if st:
        n, m = 4096, 4096
        syrk3 = next((k for k in st if k.startswith("k_syrkq_64")), "k_syrkq_64")          # (templated on the operand path since round 4)
        recs = [mfma(syrk3, st, float(m) * n * n, "Schur formation S = Q + (A'F^-1)(A'F^-1)' (src/kktsolvers.jl:33-34, :290)"),
                latency("k_ldlt_panel<true>", st, "LDL' panel launch (diag + TRSM + in-block update)"),
                latency("k_scale_At", st, "A'F^-1 for 512 Q(8) cones: O(mn) beside the SYRK", hbm_bytes_per_launch=3.0 * 8 * m * n)]
        out["c3_socp_n4096"] = dict(workload="SOCP n=4096, 512 x Q(8), p=512, dense A; Schur order 4608", dominant=[r for r in recs if r], time_share=share(st))
    # ---- config 4: one S cone of matrix order 256, n = 1024, p = 16
    st = stats(os.path.join(d, "c4_kernel_stats.csv"))
    if st:
        r, n, m = 256, 1024, 32896
        jac = None
        for k in st:
            if k.startswith("k_lg_jacobi"):
                jac = k
        syrk = "k_syrk_splitk_128" if "k_syrk_splitk_128" in st else ("k_syrk_splitk_64" if "k_syrk_splitk_64" in st else "k_syrkq_64")
        recs = [mfma(syrk, st, float(m) * n * n, "Schur formation with the scaled A' (32896 x 1024): split-K images + fixed-order reduction"),
                latency(jac, st, "one-sided block Jacobi of svd(Lz'Ls) (src/ConicIP.jl:204): serial rotation rounds, 255 per sweep at order 256",
                        rounds_per_sweep=255, note="time per launch / (sweeps x 255) = time per rotation round; 16 workgroups of 8-column blocks, wave sums by DPP") if jac else None,
                latency("k_lg_lanczos1", st, "max-step: extreme eigenvalue by Lanczos in one workgroup (src/ConicIP.jl:272-303)")]
        # the batched congruences A'F^-1: per factorisation n columns x 2 GEMMs x 2 rp^3 flop
        g = st.get("k_gemm_nt_64_batched")
        if g:
            recs.append(dict(kernel="k_gemm_nt_64_batched", role="batched congruences Rinv X Rinv' for A'F^-1 (all 1024 columns per launch, two launches per factorisation; the second on the lower tiles) and the S-cone LDL' block inverses (small launches under the same name)",
                             bound="mfma", launches=g["calls"], avg_launch_us=round(g["avg_us"], 2),
                             note="flops per factorisation of the congruences alone: %.3g (n x 2 x 2 rp^3)" % (n * 2 * 2.0 * r ** 3)))
        recs.append(latency("k_gemm_nt_small", st, "single 256^3 products of apply / max-step / NT scaling: one 16x16 tile per workgroup, k split over the waves"))
        recs.append(latency("k_lg_vecm_cols", st, "vecm of the 1024 congruence results into the columns of W' = A'F^-1 (64 entries x 64 matrices per workgroup through LDS)",
                            hbm_bytes_per_launch=8.0 * (n * r * (r + 1) / 2) * 2))
        out["c4_sdp_r256"] = dict(workload='SDP: one ("S", 32896) cone (matrix order 256), n=1024, p=16', dominant=[x for x in recs if x], time_share=share(st))
    # ---- config 5, the per-rank shard at 8 GPUs: 8 problems of order 2048 in lock-step
    st = stats(os.path.join(d, "c5_b8_kernel_stats.csv"))
    if st:
        B, N = 8, 2048
        recs = []
        for k, K_, r_ in (("k_ldlt_trailing_64<4>", 512, N - 512), ("k_ldlt_trailing_64<0>", 512, None)):
            if k in st and r_:
                recs.append(mfma(k, st, B * float(r_) * (r_ + 1) * K_, "first trailing update of the 8 problems (C operand read from Q: lazy copy), r = 1536, K = 512"))
            elif k in st:
                e = st[k]
                recs.append(dict(kernel=k, role="later trailing updates (r = 1024, 512)", bound="mfma", launches=e["calls"], avg_launch_us=round(e["avg_us"], 2),
                                 note="two shapes share the kernel name: B x (1024 x 1025 + 512 x 513) x 512 flop per pair of launches = %.3g" % (B * (1024 * 1025 + 512 * 513) * 512.0)))
        recs.append(latency("k_ldlt_panel<true>", st, "panel launch carrying 8 problems in blockIdx.z"))
        recs.append(latency("k_gemv_t", st, "triangular sweeps + Q y: one launch per block step for the 8 problems"))
        out["c5_shard_8x2048"] = dict(workload="8 dense QPs n=2048 in lock-step (config 5's per-rank shard on 8 GPUs)", dominant=[x for x in recs if x], time_share=share(st))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
