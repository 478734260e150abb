"""CPU, build container only: the C oracle against the LIVE reference on random cases
(tests/golden/ref_fuzz.py).  Skipped wherever /root/reference is absent (the GPU box)."""
import os
import sys

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.reference_available(),
                                reason="the reference is only mounted in the build container")


@pytest.mark.parametrize("first", [0, 700])
def test_oracle_matches_live_reference_on_random_cases(first):
    import ref_fuzz
    for case in range(first, first + 24):
        ref_fuzz.run_case(case)


@pytest.mark.parametrize("first", [0, 500])
def test_oracle_matches_live_reference_ads_market(first):
    """random digital-ads markets run by the reference's own example module (gen_goldens_ads.run_ads):
    budgets of all three numpy kinds, both auction strategies, connectivity < 1, actions above 1."""
    import numpy as np
    import gen_goldens_ads as gga
    from oracle import OracleEnv
    from test_oracle_vs_goldens import replay_ads
    for case in range(first, first + 12):
        rng = np.random.RandomState(9000 + case)
        n = int(rng.randint(1, 9))
        themes = sorted(rng.choice(gga.THEMES, n).tolist(), key=gga.THEMES.index)
        budgets = []
        for _ in range(n):
            r = rng.rand()
            if r < 0.4:
                budgets.append(float(rng.choice([0.5, 0.75, 1.0, 1.3, 2.0, 3.1])))
            elif r < 0.7:
                lo = float(rng.uniform(0.3, 1.0))
                budgets.append(("clipped", lo, lo + 1.0, lo + 0.1, lo + 0.9))
            else:
                budgets.append(("uniform", 0.4, float(rng.uniform(0.5, 2.5))))
        rates = None if rng.rand() < 0.5 else tuple(float(x) for x in rng.choice([1.0, 0.9, 0.7, 0.5], 3))
        g = gga.run_ads(None, themes, budgets, int(rng.randint(2, 16)), int(rng.randint(8, 60)), seed=case,
                        strategy=("second" if rng.rand() < 0.5 else "first"), rates=rates,
                        act_hi=float(rng.choice([1.0, 1.2, 2.0])))
        replay_ads(g, lambda spec: OracleEnv(spec))


def test_oracle_matches_live_reference_ads_auction_ties():
    """equal budgets and grid actions: many equal bids, so the winner / second bid are decided by the
    stable sort's arrival order (digital_ads_market.py:498-516)."""
    import numpy as np
    import gen_goldens_ads as gga
    from oracle import OracleEnv
    from test_oracle_vs_goldens import replay_ads
    for case in range(8):
        rng = np.random.RandomState(19000 + case)
        k = int(rng.randint(2, 21))
        themes = sorted(rng.choice(gga.THEMES, k).tolist(), key=gga.THEMES.index)
        budgets = [[1.0] * k, [float(rng.choice([1.0, 2.0])) for _ in range(k)], [("clipped", 0.5, 1.5, 1.0, 1.0)] * k][case % 3]
        g = gga.run_ads(None, themes, budgets, int(rng.randint(4, 30)), int(rng.randint(20, 90)), seed=case,
                        strategy=("second" if case % 2 else "first"), p_uniform=0.3)
        assert (g["step_wins"].sum() > 0)
        replay_ads(g, lambda spec: OracleEnv(spec))
