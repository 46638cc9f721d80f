#!/usr/bin/env python3
"""Viterbi training of ONE model from labelled tokens (HTKTools/HInit.c), single-Gaussian states, on the device.

EstimateModel (HInit.c:1192-1245): uniform segmentation of every token into the emitting states (UCollectData :505-531) gives the
first means / variances; then passes of Viterbi alignment (ViterbiAlign :792 -- here the library's batch alignment,
htkamd_viterbi_align, one single-model utterance per token) and re-estimation from the aligned frames (UpdateCounts :878,
UpMeans / UpVars / UpTrans :1021-1097) until the average log probability per token changes by less than epsilon.
Mixtures with more than one component need HInit's clustering and are not covered.

    from examples.hinit_model import hinit
    pk, history = hinit(capi, mmf, "S", tables, labels)       # pk: the packed model set with model "S" estimated
"""
import numpy as np

from examples.hrest_model import segments_of

LZERO = -1.0e10
MINLARG = 2.45e-308


def _estimate(frames_by_state, old_mean, min_var):
    """UpMeans / UpVars on the frames aligned to one state (sums relative to the old mean, HInit.c:964-968,1021-1056)."""
    X = np.concatenate(frames_by_state).astype(np.float64)
    occ = X.shape[0]
    mu = (X - old_mean).sum(0)
    va = ((X - old_mean) ** 2).sum(0)
    new_mean = old_mean + mu / occ
    d = new_mean - old_mean
    var = np.maximum(va / occ - d * d, min_var)
    return new_mean, var


def hinit(capi, mmf, name, tables, labels, max_iter=20, epsilon=1.0e-4, min_var=1.0e-2, min_seg=3):
    pk = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in mmf.packed().items()}
    h = mmf.logical[name]
    states = [int(s) for s in pk["hmmState"][pk["hmmStateOff"][h]:pk["hmmStateOff"][h + 1]]]
    gauss = []
    for s in states:
        c0, c1 = int(pk["stateCompOff"][s]), int(pk["stateCompOff"][s + 1])
        if c1 - c0 != 1:
            raise ValueError("hinit: state with %d mixture components (only single Gaussians are covered)" % (c1 - c0))
        gauss.append(int(pk["compGauss"][c0]))
    ne = len(states)
    N = ne + 2
    t = int(pk["hmmTrans"][h]); toff = int(pk["transOff"][t])
    segs = segments_of(name, tables, labels, ne)
    if len(segs) < min_seg:
        raise ValueError("HInit: only %d training tokens for %s (-m %d)" % (len(segs), name, min_seg))
    # ---- uniform segmentation (UCollectData): frame j (1-based) of a token of length L goes to state int((j-1)/(L/(N-2))) + 2
    by_state = [[] for _ in range(ne)]
    for X in segs:
        per = np.float32(X.shape[0]) / np.float32(ne)
        idx = (np.arange(X.shape[0], dtype=np.float32) / per).astype(np.int32)
        for j in range(ne):
            if (idx == j).any():
                by_state[j].append(X[idx == j])
    for j, g in enumerate(gauss):
        X = np.concatenate(by_state[j]).astype(np.float64)
        pk["mean"][g] = X.mean(0)
        pk["var"][g] = np.maximum((X ** 2).mean(0) - X.mean(0) ** 2, min_var)       # FlatCluster with one cluster
    model = capi.Model(pk)
    model.set_params(mean=pk["mean"], var=pk["var"])
    X = np.ascontiguousarray(np.concatenate(segs), np.float32)
    frameOff = np.concatenate([[0], np.cumsum([s.shape[0] for s in segs])]).astype(np.int32)
    labOff = np.arange(len(segs) + 1, dtype=np.int32)
    seq = np.full(len(segs), h, np.int32)
    dX = capi.DevArray(X)
    vit = capi.Viterbi(model)
    history, total, it, converged = [], np.float32(LZERO), 0, False
    while not converged and it < max_iter:
        it += 1
        al = vit.align(dX.ptr.value, frameOff, labOff, seq)
        newP = np.float32(0.0)
        for a in al:
            if a["status"] != capi.UTT_OK:
                raise ValueError("HInit: no path found in a token of %s" % name)
            newP = np.float32(newP + np.float32(a["total"]))
        newP = np.float32(newP / np.float32(len(segs)))
        delta = np.float32(newP - total)
        converged = it > 1 and abs(float(delta)) < epsilon
        if not converged:
            # UpdateCounts: frames of each state, transition counts along the state sequence (entry = 1, exit = N)
            by_state = [[] for _ in range(ne)]
            tran = np.zeros((N + 1, N + 1)); occ = np.zeros(N + 1)
            for a, S in zip(al, segs):
                last = 1
                for j in range(ne):
                    st, en = int(a["segStart"][j]), int(a["segEnd"][j])     # frames [st, en) of the token
                    if st < 0:
                        continue
                    by_state[j].append(S[st:en])
                    n = en - st
                    occ[last] += 1; tran[last][j + 2] += 1                  # entering state j+2
                    occ[j + 2] += n - 1; tran[j + 2][j + 2] += n - 1        # staying in it
                    last = j + 2
                occ[last] += 1; tran[last][N] += 1
            for j, g in enumerate(gauss):
                pk["mean"][g], pk["var"][g] = _estimate(by_state[j], pk["mean"][g].astype(np.float64), min_var)
            tp = pk["transP"][toff:toff + N * N].reshape(N, N)
            for i in range(1, N):                                           # UpTrans: rows renormalised, logs
                row = (tran[i, 2:N + 1] / occ[i]).astype(np.float32)
                s = np.float32(row.sum(dtype=np.float32))
                tp[i - 1, 0] = LZERO
                for j in range(2, N + 1):
                    x = np.float32(row[j - 2] / s)
                    tp[i - 1, j - 1] = LZERO if x < MINLARG else np.float32(np.log(np.float64(max(float(x), MINLARG))))
            model.set_params(mean=pk["mean"], var=pk["var"], transP=pk["transP"])
        total = newP
        history.append(float(newP))
    return pk, model, history, converged
