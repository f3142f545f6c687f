"""The same known-answer tests on the HIP engine (MI355X)."""
import pytest

import kats

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,fn", kats.ALL, ids=[k for k, _ in kats.ALL])
def test_device_kat(device, name, fn):
    fn(device)
