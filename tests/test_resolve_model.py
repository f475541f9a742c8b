"""The speculative-round formulation of the projection-search claim loop equals the in-order loop (CPU, no GPU)."""
import numpy as np
import pytest

from resolve_model import random_problem, sequential, speculative


@pytest.mark.parametrize("seed", range(12))
def test_rounds_equal_in_order_loop(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20, 160))
    nq = int(rng.integers(1, 200))
    cands, obs, pre = random_problem(rng, n, nq, density=[0.05, 0.2, 0.6][seed % 3], p_obs=[1.0, 0.7, 0.3][(seed // 3) % 3],
                                     junk=[0.1, 0.5][seed % 2])
    for use_second in (True, False):
        th = 100
        ref = sequential(cands, obs, n, th, 0.8, use_second, pre)
        stats = {}
        got = speculative(cands, obs, n, th, 0.8, use_second, pre, stats)
        assert got[0] == ref[0]
        assert got[1] == ref[1]
        # slot_obs only matters where a slot was taken
        assert [o for o, s in zip(got[2], got[1]) if s >= 0] == [o for o, s in zip(ref[2], ref[1]) if s >= 0]


def test_rounds_are_few():
    rng = np.random.default_rng(99)
    cands, obs, pre = random_problem(rng, 150, 192, 0.2, 1.0, 0.2)
    stats = {}
    speculative(cands, obs, 150, 100, 0.8, True, pre, stats)
    assert stats["rounds"] < 192  # fewer rounds than queries: the point of the formulation
    assert stats.get("refresh_batches", 0) <= stats.get("refreshed", 0)
