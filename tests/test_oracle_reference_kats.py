"""CPU: the reference's own known-answer tests (values ported in kats.py) pin the C oracle."""
import pytest

from kats import ALL_KATS
from oracle import OracleEnv


@pytest.mark.parametrize("kat", ALL_KATS, ids=lambda f: f.__name__)
def test_oracle_kat(kat):
    kat(lambda spec: OracleEnv(spec))
