"""First-contact GPU probe: scoring + forward-backward vs the oracle on a small synthetic set."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htk_amd import synth, capi
from oracle import pyoracle as po

def main(NS=60, M=4, NP=40, NU=6, T=150, seed=5, general=False):
    s = synth.generate(NS, M, NP, NU, T, seed)
    pk = s.packed()
    om = po.Model(pk)
    gm = capi.Model(pk)
    prep = gm.get_prepared()
    print("prep ivar eq", np.array_equal(prep["ivar"], om.ivar), "gconst eq", np.array_equal(prep["gconst"], om.gconst),
          "logwt eq", np.array_equal(prep["compLogWt"], om.compLogWt), "minDur", prep["minDur"])
    X = np.concatenate(s.feats)
    states = np.arange(NS, dtype=np.int32)
    t0 = time.time(); ref = om.score_block(X, states); t1 = time.time()
    got = gm.outp_block(X, states); t2 = time.time()
    neq = int((ref != got).sum())
    print("outp_block: %d/%d mismatching floats; max abs diff %g (oracle %.2fs, gpu call %.2fs)" % (neq, ref.size, np.abs(ref - got).max(), t1 - t0, t2 - t1))
    # forward-backward
    frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int32)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
    labs = np.concatenate(s.seqs).astype(np.int32)
    dX = capi.DevArray(X)
    fb = capi.ForwardBackward(gm, debug=True, force_general=general)
    print('=== path:', 'general (workgroup per utterance)' if general else 'wave per utterance')
    acc = capi.Accs(gm)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    cfg = capi.fb_config()
    fb.execute(cfg, acc)
    pr, st = fb.results()
    print("status", st, "frame_states", fb.frame_states(), "ktimes", fb.kernel_times())
    oacc = po.Accs(om); ocfg = po.fb_cfg()
    nev = 0
    for u in range(NU):
        rc, opr, d = po.fb_utt(om, ocfg, s.feats[u], s.seqs[u], oacc, dump=True)
        nev += d["nEval"]
        g = fb.trellis(u)
        print("utt %d rc=%d pr oracle %.10f gpu %.10f rel %.2e" % (u, rc, opr, pr[u], abs(opr - pr[u]) / abs(opr)))
        for k in ("qLo", "qHi", "aLo", "aHi"):
            if not np.array_equal(d[k], g[k]): print("   beam mismatch", k, d[k][:10], g[k][:10])
        for k in ("beta", "alpha", "outp"):
            a, b = d[k], g[k]
            if k == "outp":
                mask = ~np.isnan(a)   # oracle evaluates in-beam only; gpu scores everything
                print("   outp bit-equal in beam:", np.array_equal(a[mask], b[mask]))
                continue
            na, nb = np.isnan(a), np.isnan(b)
            if k == "beta": print("   beta nan pattern equal:", np.array_equal(na, nb))
            mask = ~na & ~nb & (a > -1e9)
            rel = np.abs(a[mask] - b[mask]) / np.maximum(1.0, np.abs(a[mask]))
            print("   %s max rel %.3e (n=%d)" % (k, rel.max() if rel.size else 0, mask.sum()))
    ga = acc.download()
    print("nEval gpu %d oracle %d" % (ga["nEval"], nev), "totalPr", ga["totalPr"], "nUttDone", ga["nUttDone"])
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs"):
        a = np.asarray(getattr(oacc, k), np.float64).reshape(-1); b = ga[k].reshape(-1)
        den = np.maximum(np.abs(a), 1e-3)
        print("   acc %-6s max abs %.3e max rel %.3e  sum %.6f vs %.6f" % (k, np.abs(a - b).max(), (np.abs(a - b) / den).max(), a.sum(), b.sum()))

if __name__ == "__main__":
    main()
    main(general=True)
    if len(sys.argv) > 1: main(1000, 8, 2000, 8, 500, 1)
