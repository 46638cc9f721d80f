# tools/ubench/score_truth.py: the scoring kernels against float64 arithmetic on the same fp32 parameters, on the states that matter (the best 8 of a frame)
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from htk_amd import capi, synth
S, M, T = 300, 16, 1500
s = synth.generate_fast(S, M, 60, 6, 300, seed=1000, model_seed=3)
pk = s.packed()
model = capi.Model(pk)
X = np.concatenate(s.feats)[:T].astype(np.float32)
st = np.arange(S, dtype=np.int32)
D = pk["vecSize"]
mean = pk["mean"].astype(np.float64); var = pk["var"].astype(np.float32)
ivar = (np.float32(1.0) / var).astype(np.float64)          # ConvDiagC's float inverse
gconst = (D * np.log(2 * np.pi) + np.log(var.astype(np.float64)).sum(1))
w = pk["compWeight"].astype(np.float64)
off = pk["stateCompOff"]; cg = pk["compGauss"]
truth = np.empty((T, S))
Xd = X.astype(np.float64)
for j in range(S):
    c = np.arange(off[j], off[j + 1]); g = cg[c]
    d = Xd[:, None, :] - mean[g][None]
    lp = np.log(w[c])[None] - 0.5 * (gconst[g][None] + (d * d * ivar[g][None]).sum(2))
    mx = lp.max(1); truth[:, j] = mx + np.log(np.exp(lp - mx[:, None]).sum(1))
top = np.argsort(-truth, axis=1)[:, :8]
rows = np.arange(T)[:, None]
print("best-8 scores: mean %.1f; all: mean %.1f" % (truth[rows, top].mean(), truth.mean()))
for name, mode in (("exact", 0), ("mfma f32", 1), ("bf16 x 3", 4), ("f16 x 2", 32)):
    got = model.outp_block(X, st, mode).astype(np.float64)
    e = (got - truth)[rows, top]
    ea = got - truth
    print("%-9s best-8: rms %.3g  mean %+.3g  max %.3g   | all: rms %.3g mean %+.3g" % (name, np.sqrt((e ** 2).mean()), e.mean(), np.abs(e).max(), np.sqrt((ea ** 2).mean()), ea.mean()))
# the part of a kernel's deviation from the reference's float arithmetic that is the same in every frame of a state (it does not average
# out along a path): per state, the mean over the frames where the state is among the best 8
ref = model.outp_block(X, st, 0).astype(np.float64)
mask = np.zeros((T, S), bool); mask[rows, top] = True
cnt = mask.sum(0)
ok = cnt >= 20
for name, mode in (("bf16 x 3", 4), ("f16 x 2", 32)):
    got = model.outp_block(X, st, mode).astype(np.float64)
    dv = np.where(mask, got - ref, 0.0)
    bias = dv.sum(0)[ok] / cnt[ok]
    resid = np.sqrt((np.where(mask, (got - ref) ** 2, 0.0).sum(0)[ok] / cnt[ok]))
    print("%-9s vs the float reference, %d states with >= 20 relevant frames: per-state mean deviation rms %.3g (max %.3g); per-frame deviation rms %.3g" % (name, ok.sum(), np.sqrt((bias ** 2).mean()), np.abs(bias).max(), np.sqrt((resid ** 2).mean())))
