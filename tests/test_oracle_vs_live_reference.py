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
