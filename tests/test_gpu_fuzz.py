"""A short, seeded slice of the randomised parity sweep (tests/fuzz_parity.py) inside the GPU suite: random extractor
configurations / image statistics and random window-search problems, HIP through the C ABI vs the oracle, bit-exact."""
import pytest

import fuzz_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_slice(pkg, oracle, synth, seed):
    msgs = []
    bad = fuzz_parity.run(pkg, oracle, synth, 20, seed, log=msgs.append)
    assert bad == 0, "\n".join(msgs)
