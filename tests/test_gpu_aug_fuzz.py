"""GPU: differential fuzz of the augmentation call against the oracle (scripts/fuzz_aug.py): random source / crop sizes (4-aligned and not:
fused and staged flow), scale ranges with up- and down-scaling by up to 2, 1 - 4 op slots, both datasets, with and without the per-pool
statistics cache -- every output value equal."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.parametrize("seed", [7, 11, 20260929])
def test_random_calls_equal_the_oracle(seed):
    import fuzz_aug
    assert fuzz_aug.run(60, seed, verbose=False) == 0
