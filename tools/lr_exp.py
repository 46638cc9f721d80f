"""Ablations of the lean left-to-right kernels on the bench workload (a build with HTKAMD_LR_DEFS=-DLR_EXP_BUILD=1; results of the
ablated passes are wrong by design, only kernel times are read).   python tools/lr_exp.py "0 1 2 4 8 16" [lean mask]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from htk_amd import capi, synth

exps = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split()]
s = synth.generate_fast(5000, 16, 6000, 1250, 500, seed=1000, model_seed=3)
model = capi.Model(s.packed()); accs = capi.Accs(model); fb = capi.ForwardBackward(model)
cfg = capi.fb_config(scoreMode=6)
X = np.concatenate(s.feats)
frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int32)
labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
labs = np.concatenate(s.seqs).astype(np.int32)
dX = capi.DevArray(X)
fb.prepare(dX.ptr.value, frameOff, labOff, labs, None)
for rnd in range(2):
    for e in exps:
        os.environ["HTKAMD_LR_EXP"] = str(e)
        ts = []
        for it in range(6):
            accs.zero(None)
            fb.execute(cfg, accs, None)
            try:
                fb.results(None)
            except Exception as ex:      # an ablated pass may fail its own checks
                pass
            if it >= 1:
                ts.append(fb.kernel_times5() if hasattr(fb, "kernel_times5") else fb.kernel_times())
        t = np.median(np.array(ts), axis=0) * 1e3
        print("exp %2d round %d: score %.3f beta %.3f alpha %.3f stats %.3f mix %.3f" % ((e, rnd) + tuple(t[:5])), flush=True)
