"""The randomised differential runs (tools/fuzz_parity.py, tools/fuzz_cbf.py) with FIXED seeds inside the driver-run suite (the packed Hessian's uncleared
last word sat under ~250 green tests for a round and was found by a hand-run fuzz only).  HIP == oracle through the C-ABI on random configurations -- every buffer after
every launch for the env step; minimisers, safe actions, convergence flags, groups and margins for the CBF module -- plus what a stale-memory bug breaks first: a second
solve of the same problem must return the same bits, and the iteration counts of the device must be the oracle's.  About a minute on the GPU box."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_fuzz_parity_64_configurations_fixed_seed():
    import fuzz_parity as fp

    rng = np.random.default_rng(20260930)
    steps = diff = 0
    for k in range(64):
        tag, n, d = fp.one_case(rng, k)  # (raises on any mask / index mismatch or a float beyond the tolerance, naming the configuration)
        steps += n
        diff += d
        assert d == 0, f"{d} differing non-observation fp32 words in {tag}"
    assert steps > 50_000 and diff == 0


def _run_cbf(seed, n_cases, cpm_agents, forced=()):
    import fuzz_cbf as fc

    fc.CPM_MAX_AGENTS = cpm_agents
    fc.CHECK_REPEAT[0] = True
    fc.UNCONVERGED[0] = 0
    for key in fc.STATS:
        fc.STATS[key] = 0
    rng = np.random.default_rng(seed)
    kinds = [0, 0, 0]
    try:
        for k in range(n_cases):
            fc.FORCE_N[0] = forced[k] if k < len(forced) else 0
            B, solve, grouping, du, dm = fc.one_case(rng, k)
            kinds[0 if (solve and not grouping) else (1 if grouping else 2)] += 1
    finally:
        fc.FORCE_N[0] = 0
        fc.CHECK_REPEAT[0] = False
        fc.CPM_MAX_AGENTS = 16
    return kinds, dict(fc.STATS), fc.UNCONVERGED[0]


def test_fuzz_cbf_40_configurations_up_to_16_vehicles():
    kinds, stats, unconverged = _run_cbf(7_2026, 40, 16)
    assert unconverged == 0 and min(kinds) > 0, (kinds, unconverged)
    assert stats["repeat_checked"] == stats["solves"] > 0
    # the device walks the oracle's iterations (same held sets, same line searches): the counts agree solve by solve -- up to the rare solve where the two sides'
    # different summation orders decide a line-search acceptance at the rounding noise of F differently (observed: 1 of 2517)
    assert stats["iter_mismatch"] <= stats["solves"] // 200, stats


def test_fuzz_cbf_24_configurations_17_to_64_vehicles_with_odd_counts():
    """Beyond 16 vehicles: two / one lanes per vehicle in the register path, the <BIG> instantiation beyond 32 -- the forced odd counts put the packed triangle's
    last word (N (2N + 1) words: odd for odd N) under the repeat check that the round-4 bug failed."""
    kinds, stats, unconverged = _run_cbf(8_2026, 24, 64, forced=(41, 33, 63, 17, 47, 35))
    assert stats["repeat_checked"] == stats["solves"] > 0
    # (crammed scenes may hit the iteration limit -- on both sides, for the same envs: one_case checks that; their iteration counts are the limit on both sides)
    assert stats["iter_mismatch"] <= stats["solves"] // 200, stats
