"""GPU: the HIP path (through the C ABI) directly against the golden vectors dumped from the
unmodified reference. Exact equality."""
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(gu.CASES))
def test_hip_matches_reference_dump(pkg, name):
    sim = gu.GpuAdapter(pkg, name)
    res = gu.run_case(sim, name)
    gu.compare(name, res, grain_cols=list(range(9)))
