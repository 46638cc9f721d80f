#!/usr/bin/env python3
"""The headline parity figures of ONE build and scoring mode: one EM iteration of bench.py's shard (tests/c3_herest.py) through the C ABI,
the re-estimated model against the model the reference's HERest writes on this box from the same files -- every entry -- and the block
scores of the mode against float64 arithmetic.  The reference's run (one process + 8-way, ~2 min) is cached in /tmp so that a sweep over
builds (tools/r06_parvar.sh) pays it once.
    python tools/headline_live.py <name> <scoreMode> [out.json]"""
import json
import os
import pickle
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import c3_herest as c3  # noqa: E402
from htk_amd import capi  # noqa: E402
import test_gpu_headline_parity as T  # noqa: E402

CACHE = os.environ.get("C3LIVE_CACHE", "/tmp/c3live.pkl")


def live_reference(s, pk):
    if os.path.exists(CACHE):
        return pickle.load(open(CACHE, "rb"))
    with tempfile.TemporaryDirectory(prefix="c3herest_") as d:
        c3.write_files(d, s, pk)
        o1, log1, _ = c3.run_reference(d, c3.NU, 1)
        o8, log8, accs = c3.run_reference(d, c3.NU, 8)
        r1, r8 = c3.read_model(os.path.join(o1, "MMF"), pk), c3.read_model(os.path.join(o8, "MMF"), pk)
        vec = c3.load_accs(pk, accs)
    pickle.dump((r1, r8, vec), open(CACHE, "wb"), protocol=4)
    return r1, r8, vec


def score_truth(mode):
    """rms deviation of the mode's block scores from float64 arithmetic and from the exact mode (the reference's floats), best 8 states per frame"""
    from htk_amd import synth
    S, M, Tn = 300, 16, 1500
    s = synth.generate_fast(S, M, 60, 6, 300, seed=1000, model_seed=3)
    pk = s.packed()
    model = capi.Model(pk)
    X = np.concatenate(s.feats)[:Tn].astype(np.float32)
    st = np.arange(S, dtype=np.int32)
    D = pk["vecSize"]
    mean = pk["mean"].astype(np.float64); var = pk["var"].astype(np.float32)
    ivar = (np.float32(1.0) / var).astype(np.float64)
    gconst = (D * np.log(2 * np.pi) + np.log(var.astype(np.float64)).sum(1))
    w = pk["compWeight"].astype(np.float64)
    off = pk["stateCompOff"]; cg = pk["compGauss"]
    truth = np.empty((Tn, S))
    Xd = X.astype(np.float64)
    for j in range(S):
        c = np.arange(off[j], off[j + 1]); g = cg[c]
        d = Xd[:, None, :] - mean[g][None]
        lp = np.log(w[c])[None] - 0.5 * (gconst[g][None] + (d * d * ivar[g][None]).sum(2))
        mx = lp.max(1); truth[:, j] = mx + np.log(np.exp(lp - mx[:, None]).sum(1))
    top = np.argsort(-truth, axis=1)[:, :8]
    rows = np.arange(Tn)[:, None]
    ref = model.outp_block(X, st, 0).astype(np.float64)
    sm = mode & (capi.SCORE_MFMA | capi.SCORE_BF16 | capi.SCORE_F16)
    got = model.outp_block(X, st, sm).astype(np.float64)
    e, er = (got - truth)[rows, top], (got - ref)[rows, top]
    return dict(rms_vs_float64=float(np.sqrt((e ** 2).mean())), mean_vs_float64=float(e.mean()), max_vs_float64=float(np.abs(e).max()),
                rms_vs_exact_mode=float(np.sqrt((er ** 2).mean())), exact_mode_rms_vs_float64=float(np.sqrt(((ref - truth)[rows, top] ** 2).mean())))


if __name__ == "__main__":
    name, mode = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "headline_live_%s.json" % name)
    rep = dict(name=name, scoreMode=mode, scores=score_truth(mode))
    print(name, "scores", json.dumps(rep["scores"]), flush=True)
    s, pk = c3.workload()
    r1, r8, vec = live_reference(s, pk)
    p, a, stats, pr = T._hip_model(capi, s, pk, mode)
    lay = capi.accs_layout(pk)
    G = int(pk["numGauss"])
    occ = vec[lay.muOcc:lay.muOcc + G]
    r = c3.compare(p, r1, r8, occ, init_mean=pk["mean"])
    rep["model"] = r
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)
    print(name, "var n>1e-4 %d worst %.3g | low-occ n>1e-4 %d worst %.3g | mean worst %.3g | weight worst %.3g | n_fail var %d low %d" % (
        r["var"]["n_above_1e4"], r["var"]["worst_rel"], r["var_low_occ"]["n_above_1e4"], r["var_low_occ"]["worst_rel"],
        r["mean"]["worst_rel"], r["weight"]["worst_rel"], r["var"]["n_fail"], r["var_low_occ"]["n_fail"]), flush=True)
