"""Host-side breakdown of one bench step (prepare / execute launch / results wait)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from htk_amd import capi, synth

s = synth.generate_fast(5000, 16, 6000, 1250, 500, seed=1000, model_seed=3)
model = capi.Model(s.packed()); accs = capi.Accs(model); fb = capi.ForwardBackward(model)
cfg = capi.fb_config(scoreMode=1)
X = np.concatenate(s.feats)
frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int32)
labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
labs = np.concatenate(s.seqs).astype(np.int32)
dX = capi.DevArray(X)
tt = np.zeros(4)
for it in range(7):
    t0 = time.perf_counter(); accs.zero(None)
    t1 = time.perf_counter(); fb.prepare(dX.ptr.value, frameOff, labOff, labs, None)
    t2 = time.perf_counter(); fb.execute(cfg, accs, None)
    t3 = time.perf_counter(); fb.results(None)
    t4 = time.perf_counter()
    if it >= 2: tt += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
print("ms: zero %.3f prepare %.3f execute(launch) %.3f results(wait) %.3f" % tuple(tt / 5 * 1e3))
print("kernels ms", [round(x * 1e3, 3) for x in fb.kernel_times()])
