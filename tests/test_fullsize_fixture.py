"""tests/golden/fullsize_trajectories.json is what the GPU tests compare the BASELINE-size runs with (north-star:
"identical iteration count to convergence").  Here, without a GPU: the fixture is complete, self-consistent, and
REPRODUCIBLE -- re-running the oracle on one of its problems (config 5, problem 3: n = 2048, a few seconds) gives the
stored record again."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _load():
    with open(os.path.join(HERE, "golden", "fullsize_trajectories.json")) as f:
        return json.load(f)


def test_fixture_is_complete():
    d = _load()
    c2 = d["c2_n8192_seed1234"]
    assert c2["n"] == 8192 and c2["status"] == "Optimal" and c2["kktsolver"] == "pivot(kktsolver_2x2)"
    assert len(c2["trace"]) == c2["Iter"] and c2["n_factor"] == c2["Iter"] + 1 and len(c2["y"]) == 64 == len(c2["idx_y"])
    # every non-terminating iteration took a step
    assert all(r["alpha"] is not None and 0 < r["alpha"] <= 1 for r in c2["trace"][:-1]) and c2["trace"][-1]["alpha"] is None
    mus = [r["mu"] for r in c2["trace"]]
    assert all(b < a for a, b in zip(mus[1:], mus[2:]))          # mu decreases once the loop is under way
    c3 = d["c3_socp_seed11"]
    assert c3["status"] == "Optimal" and c3["kktsolver"] == "kktsolver_qr" and len(c3["trace"]) == c3["Iter"]
    c4 = d["c4_sdp_r256_seed5"]
    assert (c4["status"], c4["r"], c4["n"], c4["p"]) == ("Optimal", 256, 1024, 16) and c4["kktsolver"] == "kktsolver_schur_exact"
    assert len(c4["trace"]) == c4["Iter"] and len(c4["v"]) == 64
    c5 = d["c5_n2048_seed4000"]["problems"]
    assert sorted(map(int, c5)) == list(range(64)) and all(p["status"] == "Optimal" for p in c5.values())


def test_fixture_record_is_reproducible_by_the_oracle():
    import make_fullsize_fixtures as mk
    want = _load()["c5_n2048_seed4000"]["problems"]["3"]
    got = mk.run_c2(2048, 4000 + 3)
    assert (got["status"], got["Iter"], got["n_factor"], got["n_solve"]) == (want["status"], want["Iter"], want["n_factor"], want["n_solve"])
    for a, b in zip(got["trace"], want["trace"]):
        assert abs(a["mu"] - b["mu"]) <= 1e-9 * abs(b["mu"])
    np.testing.assert_allclose(got["y"], want["y"], rtol=1e-9, atol=1e-12)
    assert got["idx_y"] == want["idx_y"]
