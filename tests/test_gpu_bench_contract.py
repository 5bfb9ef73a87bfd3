"""bench.py prints ONE JSON line with the fields the driver and the judge read (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, plus the `roofline` and
`cpu_baseline` objects).  Run here at a reduced order so that the CPU leg takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "1024", "--steps", "3", "--warmup", "1", "--no-c5"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6          # value = steps per second
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert d["converge"]["status"] == "Optimal"


def _json_line(r):
    if r.returncode != 0:                                   # (in full: pytest abbreviates long assertion operands)
        print("---- stdout tail\n" + r.stdout[-3000:] + "\n---- stderr tail\n" + r.stderr[-12000:])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_flag_is_honoured():
    """`--gpus N` must start N ranks or fail loudly (it was parsed and ignored).  On a box with fewer GPUs than
    asked for the launcher path refuses before touching the GPU; a WORLD_SIZE that disagrees with --gpus is an error too."""
    import torch
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])


def test_config5_line_through_rccl_on_one_rank():
    """The N > 1 code path of bench.py -- config 5 in lock-step, process group on the nccl (= RCCL) backend, the SUM / MAX
    all-reduces, `ranks_seen` -- driven with ONE rank (CIP_BENCH_FORCE_DIST=1), launched by torch.distributed.run exactly
    as the driver launches it."""
    env = dict(os.environ, CIP_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--workload", "c5", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    d = _json_line(r)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["batch"]["n_problems"] == 64 and d["batch"]["n_optimal"] == 64
    assert d["value"] > 0 and abs(d["value"] - d["batch"]["n_factor"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and rf["launches"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    # identical iteration counts to the oracle over the whole batch (tests/golden/fullsize_trajectories.json)
    assert d["iters_cpu"] == d["iters_gpu"], (d["iters_cpu"], d["iters_gpu"])


def test_two_ranks_on_one_gpu():
    """The N > 1 path with TWO ranks on real kernels (no 8-GPU node is available to the builder): `torch.distributed.run
    --nproc-per-node 2 bench.py --gpus 2 --workload c5` with CIP_BENCH_SHARE_GPU=1 -- both ranks on cuda:0, process group on
    gloo (RCCL refuses two ranks per device).  Exercises rank 1's shard generation (problems 1, 3, 5, ...), the
    rank-0-alone pass behind the barrier, the SUM / MAX reductions over two ranks and the per-rank busy times.  Not a
    scaling measurement: the ranks share the chip."""
    env = dict(os.environ, CIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29733", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--workload", "c5", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    d = _json_line(r)                                       # exactly one line: rank 1 prints nothing
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["ranks_share_one_gpu"] is True
    assert d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["batch"]["n_problems"] == 64 and d["batch"]["n_optimal"] == 64        # both shards, reduced with SUM
    assert d["iters_cpu"] == d["iters_gpu"], (d["iters_cpu"], d["iters_gpu"])      # the oracle's counts over the whole batch
    assert d["batch"]["n_factor"] == 627                                           # = the one-GPU run's (tests/test_gpu_lockstep.py)
    assert "c5_single_gpu" in d and d["c5_single_gpu"]["n_problems"] == 64 and d["c5_single_gpu"]["n_optimal"] == 64
    assert d["speedup_vs_c5_single_gpu"] > 0
    lo, hi = d["batch"]["rank_busy_ms_min"], d["batch"]["rank_busy_ms_max"]
    assert 0 < lo <= hi <= d["ms_per_step"] * 1.001 + 1.0
    assert abs(d["value"] - d["batch"]["n_factor"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1


def test_default_line_carries_the_round5_objects():
    """plugin_boundary (host-pointer level-2 / level-3 timings + the reference-shaped host loop through the plugin), the
    profiling-cost A/B and the explicit measured-vs-replayed flags -- at a reduced order so that the CPU legs take seconds;
    `secondary` (configs 3 and 4) only runs at the headline size and is covered by the driver's own bench run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "1024", "--steps", "3", "--warmup", "1", "--no-c5",
                        "--no-live-pmc"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    d = _json_line(r)
    pb = d["plugin_boundary"]
    assert pb["level2_ms"] > 0 and pb["level3_ms"] > 0 and pb["kkt_solves_per_s"] > 0
    assert abs(pb["kkt_solves_per_s"] - 1e3 / (pb["level2_ms"] + pb["solves_per_factor"] * pb["level3_ms"])) < 1e-6 * pb["kkt_solves_per_s"]
    hl = pb["host_loop"]
    assert hl["status"] == "Optimal" and hl["iters"] == d["converge"]["iters"] == hl["iters_native"], hl
    assert hl["wall_s"] > 0 and hl["native_loop_wall_s"] == d["converge"]["wall_s"]
    assert d["roofline"]["traffic_measured_live"] is False and d["roofline_solve"]["traffic_measured_live"] is False
    pc = d["profiling_cost"]
    assert pc["ms_per_step_with_trailing_events"] == d["ms_per_step"] and pc["ms_per_step_without"] > 0
    cb = d["cpu_baseline"]
    assert cb["strong_cpu_variant"] in cb["strong_cpu_variants"] and "potrf" in cb["strong_cpu_variants"]
    assert cb["strong_cpu_value"] > 0


def test_eight_ranks_on_one_gpu():
    """World size 8 -- the size the north-star's batched configuration is quoted at -- without an 8-GPU node: eight ranks of
    `bench.py --gpus 8 --workload c5` on cuda:0 (CIP_BENCH_SHARE_GPU=1, process group on gloo), launched by torch.distributed.run
    as the driver launches the real thing.  Every rank generates and solves its own shard (problem i -> rank i mod 8: 8 x 8), the
    SUM / MAX reductions run over eight ranks: all 64 problems Optimal, the iteration and factorisation totals are the one-GPU run's
    and the oracle's.  Not a scaling measurement: the ranks share the chip."""
    env = dict(os.environ, CIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29737", os.path.join(ROOT, "bench.py"),
                        "--gpus", "8", "--workload", "c5", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    d = _json_line(r)                                       # exactly one line: ranks 1..7 print nothing
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["ranks_share_one_gpu"] is True
    assert d["shard_sizes"] == [8] * 8
    assert d["batch"]["n_problems"] == 64 and d["batch"]["n_optimal"] == 64
    assert d["batch"]["iters"] == 563 and d["iters_gpu"] == d["iters_cpu"] == 563
    assert d["batch"]["n_factor"] == 627
    assert d["c5_single_gpu"]["n_problems"] == 64 and d["c5_single_gpu"]["n_factor"] == 627
    lo, hi = d["batch"]["rank_busy_ms_min"], d["batch"]["rank_busy_ms_max"]
    assert 0 < lo <= hi <= d["ms_per_step"] * 1.001 + 1.0
    assert abs(d["value"] - d["batch"]["n_factor"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_headline_size_line_carries_the_literal_3x3_route_and_the_shard_passes():
    """Round-5 review: what is claimed must be in the driver-timed line.  At the headline size (no CPU legs, no PMC children: those
    are the driver's own run) the default line carries `secondary.full3x3` -- the literal 3x3 route of src/kktsolvers.jl:254-257 at
    N = 16384: ms per factorisation, whole-factor and trailing-update TFLOP/s, the iteration count of the Schur route and the oracle
    -- and `c5_shards`: lock-step passes over rank 0's shard of the 8 / 4 / 2 / 1-GPU job (8 / 16 / 32 / 64 problems)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-plugin-boundary"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    d = _json_line(r)
    f3 = d["secondary"]["full3x3"]
    assert f3["kkt_order"] == 16384 and f3["status"] == "Optimal"
    assert f3["iters"] == d["converge"]["iters"] == f3["iters_cpu"] == d["iters_cpu"]
    assert f3["ms_ldlt_factor"] > 0 and 0 < f3["ldlt_tflops_whole_factor"] < 78.6
    rt = f3["roofline_trailing"]
    assert rt["bound"] == "mfma" and 0 < rt["frac"] < 1 and rt["launches"] > 0
    # recompute: the events' flops are the trailing updates' share of N^3 / 3
    assert rt["algorithmic_flops_per_launch"] * rt["launches"] < f3["algorithmic_flops_per_factor"]
    sh = d["c5_shards"]
    assert [sh[k]["problems"] for k in ("8", "16", "32", "64")] == [8, 16, 32, 64]
    assert all(sh[k]["n_optimal"] == sh[k]["problems"] and sh[k]["ms_per_pass"] > 0 for k in ("8", "16", "32", "64"))
    assert sh["64"]["n_factor"] == 627 and sh["8"]["ms_per_pass"] < sh["64"]["ms_per_pass"]
    assert abs(sh["projected_speedup_8_gpus"] - sh["64"]["ms_per_pass"] / sh["8"]["ms_per_pass"]) < 1e-9
    for name in ("c3", "c4"):
        assert d["secondary"][name]["status"] == "Optimal" and d["secondary"][name]["iters"] == d["secondary"][name]["iters_cpu"]
