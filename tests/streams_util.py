"""Helpers of the multi-stream tests: the demo's stream sets (tests/golden/make_streams_golden.py) and random multi-stream sets."""
import os

import numpy as np

DEMO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo")


def demo_utterances(native, oracle, mmf):
    """The demo's training files as (MFCC_E_D table, model sequence) pairs (TARGETKIND = MFCC_E_D: deltas added at load)."""
    utts = []
    for f in sorted(os.listdir(os.path.join(DEMO, "train"))):
        if not f.endswith(".mfc"):
            continue
        X, _, _ = native.parm_read(os.path.join(DEMO, "train", f))
        labs = native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab")))
        utts.append(dict(feat=oracle.parm_qualify(X, hasD=True), seq=np.array([mmf.logical[n] for n, _, _, _ in labs], np.int32)))
    return utts


def make_multistream(pk, widths, rng, max_mix=3, single=()):
    """A random multi-stream set on the topology of the single-stream packed set `pk`: streams of the given widths (consecutive pieces
    of the row), per (state, stream) 1..max_mix components (streams listed in `single` always 1), Gaussians in undivided rows as
    include/htk_amd.h describes them (mean 0 / variance inf outside the stream)."""
    D, S, NS = int(pk["vecSize"]), int(pk["numStates"]), len(widths)
    assert sum(widths) == D
    dimStream = np.repeat(np.arange(NS), widths).astype(np.int32)
    off, wt, cg, mean, var = [0], [], [], [], []
    base_mean = np.asarray(pk["mean"], np.float32).reshape(-1, D)
    base_var = np.asarray(pk["var"], np.float32).reshape(-1, D)
    sco = np.asarray(pk["stateCompOff"])
    for s in range(S):
        g0 = int(pk["compGauss"][sco[s]])
        for k in range(NS):
            M = 1 if k in single else int(rng.integers(1, max_mix + 1))
            w = rng.random(M).astype(np.float32) + 0.2
            w /= w.sum()
            for m in range(M):
                mu = np.zeros(D, np.float32); va = np.full(D, np.inf, np.float32)
                sel = dimStream == k
                mu[sel] = base_mean[g0][sel] + rng.normal(0, 0.7, int(sel.sum())).astype(np.float32)
                va[sel] = base_var[g0][sel] * np.float32(rng.uniform(0.7, 1.5))
                cg.append(len(mean)); mean.append(mu); var.append(va); wt.append(w[m])
            off.append(len(wt))
    out = dict(pk)
    out.update(numStreams=NS, dimStream=dimStream, stateCompOff=np.array(off, np.int32), compWeight=np.array(wt, np.float32),
               compGauss=np.array(cg, np.int32), mean=np.array(mean, np.float32), var=np.array(var, np.float32), gconst=None,
               numComp=len(wt), numGauss=len(mean))
    return out
