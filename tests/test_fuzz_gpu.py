"""Randomised parity sweep of the convolution entry points over every dispatch branch (scripts/fuzz_conv.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_conv_shapes_match_torch(seed):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_conv
    bad, n = fuzz_conv.run(30, seed, verbose=False)
    assert n > 10 and bad == 0


def test_small_map_convolutions_every_combination():
    """conv_small_body (64 x 32 tiles, 32 x 16 on the 4x4 maps) and whatever the dispatch falls back to around it: forward, data gradient
    and weight gradient of every (map, batch, sources, width, upsample) combination of scripts/fuzz_conv.py small_map_cases()."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_conv
    cases = fuzz_conv.small_map_cases()
    bad, n = fuzz_conv.run(0, 0, verbose=False, cases=cases)
    assert n == len(cases) and n > 60 and bad == 0


@pytest.mark.parametrize("seed", [21])
def test_random_groupnorm_shapes_match_torch(seed):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_gn
    assert fuzz_gn.run(30, seed, verbose=False) == 0
