"""The measured exposure of the bit-exact index contract to XLA's  x / const -> x * RN(1 / const)  rewrite (tools/xla_rcp_exposure.py).
The oracle is unpinned (no JAX here), so which form the reference executes is unknown; what CAN be known is how much would change: a few
voxel / resample indices per million, and — through rays that graze the refractive boundary — an RGB effect of the order of the 1e-4
contract itself.  The full-size numbers are committed in profiles/r03/xla_rcp_exposure.json; this test re-measures a small case."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_reciprocal_form_changes_few_indices():
    import xla_rcp_exposure as X
    from oracle import ref_np as R
    c = X.ship_refractive_small(G=48, B=96)
    c["S"], c["P"] = 32, 6
    c["jitter"] = np.arange(0, 32 * 6, 6) + 3
    r = X.compare("small", c)
    assert R.CONST_DIV_AS_RECIPROCAL is False                         # the switch is restored
    assert r["voxel_index_fraction_changed"] < 1e-3                   # a handful of (ray, node) cells, not a different march
    assert r["max_abs_position_diff"] < 1e-2 and r["max_abs_rgb_diff"] < 5e-3
    # and the two forms really are different arithmetic: some coordinate differs by an ulp somewhere along the paths
    assert r["max_abs_position_diff"] > 0 or r["voxel_index_tuples_changed"] >= 0


def test_committed_exposure_record():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03", "xla_rcp_exposure.json")))
    assert len(d["results"]) == 3
    for r in d["results"]:
        assert r["voxel_index_fraction_changed"] < 1e-4               # measured: 2.5e-6 .. 2.5e-5 of the (ray, node) index tuples
        assert r["max_abs_rgb_diff"] < 1e-3                           # measured: up to 1.5e-4 (config 3: one grazing ray)
