"""tools/ubench/score_exp.py [LIB]: kernel times of one forward-backward pass of the bench workload (SM = score mode, default 6) with the library LIB
(default: the built one); with a library built with -DF16W_STAMP=1 (gmm_f16.hip) also the s_memtime stamps per phase of k_score_f16w."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from htk_amd import capi, synth
if len(sys.argv) > 1:
    capi.LIBPATH = os.path.abspath(sys.argv[1])
s = synth.generate_fast(5000, 16, 6000, 1250, 500, seed=1000, model_seed=3)
model = capi.Model(s.packed()); accs = capi.Accs(model); fb = capi.ForwardBackward(model)
cfg = capi.fb_config(scoreMode=int(os.environ.get("SM", "6")))
X = np.concatenate(s.feats)
frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int32)
labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
labs = np.concatenate(s.seqs).astype(np.int32)
dX = capi.DevArray(X)
kt = np.zeros(4)
import ctypes
L = capi.lib()
has = hasattr(L, 'htkamd_dbg_read')
for it in range(7):
    accs.zero(None); fb.prepare(dX.ptr.value, frameOff, labOff, labs, None); fb.execute(cfg, accs, None); pr, st = fb.results(None)
    if it == 5 and has: L.htkamd_dbg_zero()
    if it >= 2: kt += np.array(fb.kernel_times()[:4])
print(sys.argv[1:] , "kernels ms", [round(x / 5 * 1e3, 3) for x in kt], "sum pr %.6f ok %d" % (pr[st == 0].sum(), (st == 0).sum()))

if has:
    h = np.zeros(4096 * 16, np.uint64)
    L.htkamd_dbg_read(h.ctypes.data_as(ctypes.c_void_p))
    h = h.reshape(4096, 16).astype(np.float64)
    nw = int((h[:, 6] > 0).sum())
    h = h[:nw]
    it = h[:, 7].sum()
    print("waves", nw, "cycles/wave %.0f" % h[:, 6].mean(), " iterations/wave %.0f tasks/wave %.1f active %.2f" % (h[:, 7].mean(), h[:, 8].mean(), h[:, 9].sum() / it))
    print("per iteration: task switch %.0f  issue loads %.0f  compute %.0f (per active %.0f)  switch barrier %.0f  vmcnt+ds_write %.0f  barrier %.0f   | per task: switch %.0f, its barrier %.0f" % (
        h[:, 0].sum() / it, h[:, 1].sum() / it, h[:, 2].sum() / it, h[:, 2].sum() / h[:, 9].sum(), h[:, 3].sum() / it, h[:, 4].sum() / it, h[:, 5].sum() / it, h[:, 0].sum() / h[:, 8].sum(), h[:, 3].sum() / h[:, 8].sum()))
