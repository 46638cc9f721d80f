"""Tolerance of the MFMA scoring path: re-estimated parameters (exact scores vs MFMA scores) on a golden case and on a
larger synthetic set.  Run on the GPU box: python tools/mfma_tolerance.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from htk_amd import capi, synth
from util import batch_arrays, load_case


def run(pk, utts, mode):
    model = capi.Model(pk)
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = capi.DevArray(X)
    fb = capi.ForwardBackward(model)
    acc = capi.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(capi.fb_config(scoreMode=mode), acc)
    pr, st = fb.results()
    a = acc.download()
    model.update(acc, a["vec"], minEgs=1, singleProcess=True)
    return pr, a, model.get_params()


def report(name, pk, utts):
    pr0, a0, p0 = run(pk, utts, 0)
    pr1, a1, p1 = run(pk, utts, 1)
    print("==", name, "utts", len(utts))
    print(" pr rel max", np.max(np.abs(pr1 - pr0) / np.abs(pr0)))
    occ = a0["muOcc"]
    for k in ("muOcc", "wtOcc", "trOcc"):
        d = np.abs(a1[k] - a0[k]); ok = np.abs(a0[k]) > 1e-3
        print(" %s rel max %.3g" % (k, np.max(d[ok] / np.abs(a0[k][ok]))))
    sigma = np.sqrt(p0["var"])
    dm = np.abs(p1["mean"] - p0["mean"]) / np.maximum(np.abs(p0["mean"]), sigma)
    dv = np.abs(p1["var"] - p0["var"]) / p0["var"]
    G, D = p0["mean"].shape if p0["mean"].ndim == 2 else (len(occ), len(p0["mean"]) // len(occ))
    dm = dm.reshape(G, D).max(1); dv = dv.reshape(G, D).max(1)
    for lo, hi in ((0, 1), (1, 5), (5, 20), (20, 1e9)):
        sel = (occ >= lo) & (occ < hi)
        if sel.any():
            print(" occ [%g,%g): n=%d  mean rel max %.3g  var rel max %.3g" % (lo, hi, sel.sum(), dm[sel].max(), dv[sel].max()))
    print(" weights abs max %.3g" % np.max(np.abs(p1["compWeight"] - p0["compWeight"])))


if __name__ == "__main__":
    for n in ("fb_small", "fb_topo"):
        c = load_case(n)
        report(n, c["pk"], c["utts"])
    s = synth.generate_fast(200, 8, 300, 400, 300, seed=5, model_seed=6)
    utts = [dict(feat=f, seq=np.asarray(q, np.int32)) for f, q in zip(s.feats, s.seqs)]
    report("synth 200x8, 400 utts", s.packed(), utts)
