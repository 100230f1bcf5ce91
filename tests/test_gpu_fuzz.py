"""GPU: a short run of the randomised shape sweep (tools/fuzz_parity.py): random image sizes, batches, keypoint budgets,
ragged LightGlue batches and stream-mode shapes against the oracle.  SuperPoint bit-exact, match lists identical
(flips are tolerated only where the stated score tolerance allows them: within tol of the 0.1 filter, or two best
row / column probabilities closer than 2 tol -- tools/fuzz_parity.py:borderline), scores within 5e-4."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_fuzz_parity_short(seed):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    # bounded by a case count, not by the clock: every box -- fast, slow or busy -- runs the same 50 cases of the seed (a 12 s budget reached 35 to 70 of them)
    assert fz.main(seconds=300.0, seed=seed, max_cases=50) == 0
