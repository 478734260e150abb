"""GPU: random differential cases (tests/fuzz_cases.py: steps on both engines; tests/fuzz_rollouts.py: the fused rollout
kernels), the HIP path vs the CPU oracle, bit-exact."""
import pytest

from fuzz_cases import run_case
from fuzz_rollouts import run_rollout_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [0, 40, 80, 1320])
def test_random_differential_cases(first):
    for case in range(first, first + 40):
        run_case(case)


@pytest.mark.parametrize("first", [0, 100, 7000])
def test_random_differential_rollouts(first):
    assert sum(run_rollout_case(case) for case in range(first, first + 100)) > 0
