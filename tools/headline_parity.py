#!/usr/bin/env python3
"""Which arithmetic choice costs how much parity at the headline size: one EM iteration of bench.py's shard per score mode, the
re-estimated model against the reference's (tests/golden/c3_herest.npz: 2 048 sampled Gaussians), report per mode.
    python tools/headline_parity.py [modes...]      modes: bit 1 fp32 matrix-core scores, 2 fast LAdd, 4 bf16 x 3 scores"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import c3_herest as c3  # noqa: E402
from htk_amd import capi  # noqa: E402
import test_gpu_headline_parity as T  # noqa: E402

if __name__ == "__main__":
    modes = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 6]
    z = np.load(c3.GOLDEN, allow_pickle=False)
    g = (z["states"][:, None].astype(np.int64) * c3.M + np.arange(c3.M)[None, :]).reshape(-1)
    r1 = dict(mean=z["mean1"], var=z["var1"], compWeight=z["w1"], transP=z["trans1"])
    r8 = dict(mean=z["mean8"], var=z["var8"], compWeight=z["w8"], transP=z["trans8"])
    s, pk = c3.workload()
    for mode in modes:
        p, a, stats, pr = T._hip_model(capi, s, pk, mode)
        got = dict(mean=p["mean"][g], var=p["var"][g], compWeight=p["compWeight"][g], transP=p["transP"])
        r = c3.compare(got, r1, r8, z["occ"].astype(np.float64))
        print("mode %d: mean worst %.2e p9999 %.2e n>1e-4 %d | var worst %.2e p9999 %.2e n>1e-4 %d | weight worst %.2e n>1e-4 %d | avg logprob/frame %.9f" % (
            mode, r["mean"]["worst_rel"], r["mean"]["p9999_rel"], r["mean"]["n_above_1e4"], r["var"]["worst_rel"], r["var"]["p9999_rel"], r["var"]["n_above_1e4"],
            r["weight"]["worst_rel"], r["weight"]["n_above_1e4"], a["totalPr"] / a["totalT"]), flush=True)
