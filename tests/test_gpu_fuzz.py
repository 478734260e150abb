"""GPU: random differential cases (tests/fuzz_cases.py), the HIP path vs the CPU oracle, bit-exact."""
import pytest

from fuzz_cases import run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [0, 40, 80, 1320])
def test_random_differential_cases(first):
    for case in range(first, first + 40):
        run_case(case)
