"""End-to-end use of the boundary the way the reference's optimisation loop uses a shape
(mi.render -> backward -> opt.step -> params.update, src/python/python/util.py:185-232,356-523):
authored stand-in for BASELINE.json configs[4] (examples/inverse_heights.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_adam_recovers_heights(hf):
    import inverse_heights
    hist, err, wall = inverse_heights.run(grid=64, film=128, spp=1, steps=60, lr=0.02, verbose=False)
    assert hist[-1] < 0.1 * hist[0], (hist[0], hist[-1])      # the loss drops by > 10x
    assert err < 0.12                                          # start: mean |0.5 - h*| ~ 0.17
