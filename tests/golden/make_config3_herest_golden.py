#!/usr/bin/env python3
"""Known answer for the HEADLINE workload (bench.py's rank-0 shard: 5 000 tied states x 16 mixtures, 1 250 utterances of 500 frames,
HERest -m 3 -v 0.01) from the reference's HERest (oracle/_ref): the model one process writes, the model eight `-p k` processes and a
`-p 0` merge write (the reference's OWN run-to-run difference: float accumulators, another summation order), the Gaussians'
occupancies from the eight dumps, and HERest's summary lines.

Committed: tests/golden/c3_herest.npz -- a seeded sample of 128 tied states (2 048 Gaussians) with both models' means, variances and
weights, the occupancies, the transition matrix, plus the counts of the whole set (how many of its 3.1 M entries the reference itself
reproduces to 1e-4).  The model set and the data are regenerated from their seeds by the tests (tests/c3_herest.py: workload()).
    python tests/golden/make_config3_herest_golden.py       (needs oracle/_ref; ~3 minutes, ~0.5 GB of scratch under /tmp)"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import c3_herest as c3  # noqa: E402

N_SAMPLE_STATES = 128


def main():
    t0 = time.time()
    s, pk = c3.workload()
    print("workload %.1f s" % (time.time() - t0))
    with tempfile.TemporaryDirectory(prefix="c3herest_") as d:
        c3.write_files(d, s, pk)
        print("files %.1f s" % (time.time() - t0))
        o1, log1, _ = c3.run_reference(d, c3.NU, 1)
        print("1 process %.1f s" % (time.time() - t0))
        o8, log8, accs = c3.run_reference(d, c3.NU, 8)
        print("8-way %.1f s" % (time.time() - t0))
        r1, r8 = c3.read_model(os.path.join(o1, "MMF"), pk), c3.read_model(os.path.join(o8, "MMF"), pk)
        vec = c3.load_accs(pk, accs)
    from htk_amd import capi
    lay = capi.accs_layout(pk)
    G = int(pk["numGauss"])
    occ = vec[lay.muOcc:lay.muOcc + G].copy()
    keep = lambda log: [l.strip() for l in log.splitlines() if "average log prob" in l or "floored variance" in l or "WARNING" in l]
    print(keep(log1), keep(log8))
    whole = c3.compare(r8, r1, r8, occ)            # the reference against itself: got = 8-way, ref = 1 process
    print(json.dumps(whole))
    rng = np.random.default_rng(20261002)
    stOcc = occ.reshape(c3.NS, c3.M).sum(1)
    cand = np.where(stOcc >= 50.0)[0]
    states = np.sort(rng.choice(cand, N_SAMPLE_STATES, replace=False)).astype(np.int32)
    g = (states[:, None] * c3.M + np.arange(c3.M)[None, :]).reshape(-1)
    np.savez_compressed(c3.GOLDEN, states=states, occ=occ[g].astype(np.float32),
                        mean1=r1["mean"][g], var1=r1["var"][g], w1=r1["compWeight"][g], trans1=r1["transP"],
                        mean8=r8["mean"][g], var8=r8["var"][g], w8=r8["compWeight"][g], trans8=r8["transP"],
                        init_mean_sum=np.float64(pk["mean"].astype(np.float64).sum()), x_sum=np.float64(sum(float(x.astype(np.float64).sum()) for x in s.feats)),
                        log1="\n".join(keep(log1)), log8="\n".join(keep(log8)), whole_set_self=json.dumps(whole),
                        total_occ=np.float64(occ.sum()), n_occ_ge2=np.int64((occ >= 2.0).sum()))
    print("wrote", c3.GOLDEN, os.path.getsize(c3.GOLDEN), "bytes; %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
