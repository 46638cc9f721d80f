"""CPU check of the headline-workload fixture (tests/golden/c3_herest.npz): it loads, the comparison rule of tests/c3_herest.py accepts
the reference against itself (8-way merge vs one process), and the whole-set counts recorded when it was made say what DESIGN.md quotes."""
import json

import numpy as np

import c3_herest as c3


def test_fixture_reference_self_consistency():
    z = np.load(c3.GOLDEN, allow_pickle=False)
    assert z["states"].shape == (128,) and z["mean1"].shape == (2048, 39) and z["var8"].shape == (2048, 39)
    r1 = dict(mean=z["mean1"], var=z["var1"], compWeight=z["w1"], transP=z["trans1"])
    r8 = dict(mean=z["mean8"], var=z["var8"], compWeight=z["w8"], transP=z["trans8"])
    r = c3.compare(r8, r1, r8, z["occ"].astype(np.float64))
    for k in ("mean", "var", "weight"):
        assert r[k]["n_fail"] == 0
    assert r["var"]["worst_rel"] < 1e-4 and r["mean"]["worst_rel"] < 1e-5
    whole = json.loads(str(z["whole_set_self"]))
    # the reference reproduces itself to 1e-4 on all but a handful of the set's 2.9 M variances with two frames or more
    assert whole["var"]["n"] > 2_800_000 and whole["var"]["n_self_above_1e4"] <= 3 and whole["mean"]["n_self_above_1e4"] == 0
    assert "average log prob per frame = -6.182121e+01" in str(z["log1"])
