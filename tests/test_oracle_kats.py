"""Pins the CPU oracle against every known-answer test the reference holds for
the step path (tests/test_solvers.cu, tests/test_links.cu).  CPU only."""
import pytest

import kats


@pytest.mark.parametrize("name,fn", kats.ALL, ids=[k for k, _ in kats.ALL])
def test_oracle_kat(oracle, name, fn):
    fn(oracle)
