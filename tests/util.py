"""Shared helpers of the test-suite (fixtures loading, tolerance rules)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFDIR = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref")


def load_case(name: str) -> dict:
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    pk = {k[3:]: z[k] for k in z.files if k.startswith("pk_")}
    for k in ("vecSize", "numStates", "numComp", "numGauss", "numTrans", "numPhys"):
        pk[k] = int(pk[k])
    pk.setdefault("gconst", None)
    n = int(z["nUtt"])
    utts = []
    for u in range(n):
        d = dict(seq=z["seq_%d" % u], feat=z["feat_%d" % u], ok=int(z["ok_%d" % u]))
        for k in ("pr", "qLo", "qHi", "aLo", "aHi", "beta", "alpha", "outp", "occ"):
            key = "%s_%d" % (k, u)
            if key in z.files:
                d[k] = z[key]
        utts.append(d)
    acc = {k[4:]: z[k] for k in z.files if k.startswith("acc_")}
    upd = {k[4:]: z[k] for k in z.files if k.startswith("upd_")}
    t = str(z["tflag"]).replace("-t", "").split()
    prune = dict(pruneInit=float(t[0]), pruneInc=float(t[1]) if len(t) > 1 else 0.0,
                 pruneLim=float(t[2]) if len(t) > 1 else float(t[0])) if t else {}
    return dict(pk=pk, utts=utts, acc=acc, upd=upd, prune=prune, log=str(z["herest_log"]))


def batch_arrays(utts):
    X = np.concatenate([u["feat"] for u in utts]).astype(np.float32)
    frameOff = np.concatenate([[0], np.cumsum([u["feat"].shape[0] for u in utts])]).astype(np.int32)
    labOff = np.concatenate([[0], np.cumsum([len(u["seq"]) for u in utts])]).astype(np.int32)
    labs = np.concatenate([u["seq"] for u in utts]).astype(np.int32)
    return X, frameOff, labOff, labs


def eq_nan(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def mmf_fmt(a):
    """What the reference's text MMF shows of a float: '%e' (7 significant digits), read back as float32."""
    return np.array([float("%e" % x) for x in np.asarray(a, np.float64).reshape(-1)], np.float32)


def trans_as_saved(logtp, transN, transOff):
    """PutTransMat (HModel.c:2877-2912): exp in double -> float, rows re-normalised in float, last row zero."""
    out = []
    for t, N in enumerate(transN):
        m = np.asarray(logtp[transOff[t]:transOff[t + 1]], np.float32).reshape(N, N)
        v = np.where(m < -0.5e10, 0.0, np.exp(m.astype(np.float64))).astype(np.float32)
        for i in range(N - 1):
            s = np.float32(0.0)
            for j in range(N):
                s = np.float32(s + v[i, j])
            v[i] = (v[i] / s).astype(np.float32)
        v[N - 1] = 0
        out.append(v.reshape(-1))
    return np.concatenate(out)


def acc_close(got, ref, what, rtol=1e-4, floor=1e-3):
    """Accumulators: |d| <= rtol * max(|ref|, floor) -- the reference itself differs by ~4e-6 relative between a
    single process and an 8-way -p merge, so 1e-4 with an absolute floor is the parity bar."""
    got = np.asarray(got, np.float64).reshape(-1); ref = np.asarray(ref, np.float64).reshape(-1)
    err = np.abs(got - ref); lim = rtol * np.maximum(np.abs(ref), floor)
    bad = np.where(err > lim)[0]
    assert len(bad) == 0, "%s: %d of %d outside tolerance, worst |d|=%g at ref=%g" % (
        what, len(bad), len(ref), err[bad].max(), ref[bad[np.argmax(err[bad])]])
