"""tools/ubench/score_cmp.py S M T [LIB]: block scores of the bf16 x 3 mode against the exact mode on a synthetic set of S states x M mixtures, T frames."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from htk_amd import capi, synth
if len(sys.argv) > 4: capi.LIBPATH = os.path.abspath(sys.argv[4])
S, M = int(sys.argv[1]), int(sys.argv[2]); T = int(sys.argv[3])
s = synth.generate_fast(S, M, 60, 2, T, seed=5, model_seed=3)
model = capi.Model(s.packed())
X = np.concatenate(s.feats)[:T]
st = np.arange(S, dtype=np.int32)
ex = model.outp_block(X, st, 0)
bf = model.outp_block(X, st, 4)
d = np.abs(ex - bf)
print("rms diff %.3g  mean %.3g  p99.9 %.3g" % (np.sqrt((d.astype(np.float64)**2).mean()), d.mean(), np.quantile(d, 0.999)), "score range", ex.min(), ex.max())
print("max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "shape", d.shape)
bad = np.argwhere(d > 1e-3)
print("bad count", len(bad))
if len(bad):
    print("bad frames", sorted(set(bad[:, 0]))[:40])
    print("bad states", sorted(set(bad[:, 1]))[:40])
