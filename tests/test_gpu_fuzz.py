"""GPU: a short randomised parity sweep of the coupled step (the long one: scripts/fuzz_forces.py, profiles/r02_k_*)."""
import pytest

import fuzz_util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [5003, 5018, 5042, 5077, 5113, 5150, 5201, 5333])
def test_random_packing_is_bit_equal_to_the_oracle(pkg, po, seed):
    desc, ok, tab, gat, _ = fuzz_util.run_case(pkg, po, seed)
    assert ok is not None, desc
    assert ok, desc
    assert tab > 0, desc        # the link-sum table served grains (not everything fell to the gather queue)
