"""A 20-second, fixed-seed slice of tools/fuzz_parity.py (the randomised HIP-vs-oracle sweep over every entry point: warps, the fused
tile kernel, the blenders in all precisions and cycles, mask preparation, the seam finder, the linear pair blend, whole pairs through
PairStitcher) under -m gpu.  The long soaks are kept as JSON summaries under profiles/."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_slice_20s(gpu):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import fuzz_parity
    out = fuzz_parity.run(20.0, 20260928, verbose=True)
    assert out["mismatches"] == 0, out["failing_seeds"]
    assert out["cases"] >= 100 and all(v["cases"] > 0 for v in out["per_family"].values()), out["per_family"]


# seeds the soaks have tripped over, kept as cases of their own (family, seed):
#   case_many_tiles 20260988784363 - round 5: 35 tiles in mode 2, two bands; one column strip of the cycle took k_collapse_gather (16-byte level-1
#   records, produced again on its columns), its neighbour read a shared tile's planar level 1 from feed() behind it
REGRESSIONS = [("case_many_tiles", 20260988784363)]


@pytest.mark.parametrize("family,seed", REGRESSIONS)
def test_fuzz_regression_seed(gpu, family, seed):
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import fuzz_parity
    fuzz_parity.G.load()
    getattr(fuzz_parity, family)(np.random.default_rng(seed))
