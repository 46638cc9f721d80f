#!/usr/bin/env python3
"""Viterbi training of ONE model from labelled tokens (HTKTools/HInit.c), Gaussian mixture states, on the device.

EstimateModel (HInit.c:1192-1245): uniform segmentation of every token into the emitting states (UCollectData :505-531) and, per
state, FlatCluster (HTrain.c:763-803: split the cluster of largest average cost, k-means until the cost settles) give the first
weights / means / variances; then passes of Viterbi alignment (ViterbiAlign :792 -- here the library's batch alignment,
htkamd_viterbi_align_mode with SOutP / DOutP arithmetic, one single-model utterance per token), the most likely component of the
aligned state for every frame (FindBestMixes :738 -- the per-Gaussian MOutP scores come from htkamd_outp_block_mode over one
single-Gaussian scoring state per component) and re-estimation from those hard assignments (UpdateCounts :878, UpWeights / UpMeans /
UpVars / UpTrans :994-1097) until the average log probability per token changes by less than epsilon.

    from examples.hinit_model import hinit
    pk, model, history, converged = hinit(capi, mmf, "S", tables, labels)       # pk: the packed model set with model "S" estimated
"""
import numpy as np

from examples.hrest_model import segments_of

LZERO = -1.0e10
MINLARG = 2.45e-308
MIN_CLUST_SIZE, MAX_CLUST_ITER = 3, 10                 # HTrain.c:69-70


def _estimate(frames, old_mean, min_var):
    """UpMeans / UpVars on the frames given to one component (sums relative to the old mean, HInit.c:964-968,1021-1056)."""
    X = np.concatenate(frames).astype(np.float64)
    occ = X.shape[0]
    mu = (X - old_mean).sum(0)
    va = ((X - old_mean) ** 2).sum(0)
    new_mean = old_mean + mu / occ
    d = new_mean - old_mean
    var = np.maximum(va / occ - d * d, min_var)
    return new_mean, var


def flat_cluster(X, nc):
    """FlatCluster (HTrain.c:763-803) with Euclidean distance (NULLC) and diagonal cluster covariances: returns sizes [nc], centres
    [nc, D] (float32) and variances [nc, D] (float32).  Arithmetic as the reference's: float differences, double sums of squares in
    dimension order, float costs and centre sums in item order."""
    X = np.ascontiguousarray(X, np.float32)
    n, D = X.shape
    if n < nc:
        raise ValueError("InitClustering: only %d items for %d clusters" % (n, nc))
    f32 = np.float32
    cmap = np.zeros(n, np.int64)
    ctr = np.zeros((nc, D), f32); csize = np.zeros(nc, np.int64); ave = np.zeros(nc, f32)
    csize[0] = n; ave[0] = f32(1.0)

    def find_centres(k):
        for c in range(k):
            rows = X[cmap == c]
            tot = np.cumsum(rows, axis=0, dtype=f32)[-1] if rows.shape[0] else np.zeros(D, f32)
            with np.errstate(all="ignore"):
                ctr[c] = tot / f32(csize[c])

    def perturb(c, c2):
        v = ctr[c].copy()
        x = np.abs(v.astype(np.float64) * 0.01).astype(f32)
        x = np.where(x < f32(0.0001), f32(0.0001), x)
        ctr[c] = v + x; ctr[c2] = v - x

    def allocate(k):
        d = X[:, None, :] - ctr[None, :k, :]                                  # float differences
        dist = np.sqrt(np.cumsum(d.astype(np.float64) ** 2, axis=2)[:, :, -1]).astype(f32)
        best = np.argmin(dist, axis=1)                                        # first minimum, as the strict `d < min` scan keeps
        mn = dist[np.arange(n), best]
        total = np.cumsum(mn, dtype=f32)[-1]
        for c in range(k):
            sel = mn[best == c]
            csize[c] = sel.shape[0]
            ave[c] = np.cumsum(sel, dtype=f32)[-1] if sel.shape[0] else f32(0.0)
        cmap[:] = best
        for c in range(k):
            if csize[c] < MIN_CLUST_SIZE:
                return c + 1, total
            ave[c] = ave[c] / f32(csize[c])
        return 0, total

    find_centres(1)
    for c in range(2, nc + 1):
        big, mx = 0, ave[0]                                                   # BiggestCluster
        for k in range(1, c - 1):
            if ave[k] > mx:
                mx = ave[k]
                if csize[k] >= MIN_CLUST_SIZE:
                    big = k
        perturb(big, c - 1)
        old = f32(1e10)
        it = 0
        while True:
            repair = 0
            ce, new = allocate(c)
            while ce != 0:
                repair += 1
                if repair > c:
                    break
                full, mxs = 0, csize[0]                                       # FullestCluster
                for k in range(c):
                    if csize[k] > mxs:
                        mxs = csize[k]; full = k
                perturb(full, ce - 1)
                ce, new = allocate(c)
            if ce != 0:
                raise ValueError("FlatCluster: failed to make %d clusters" % c)
            converged = it >= MAX_CLUST_ITER or f32(f32(old - new) / old) < f32(0.001)
            find_centres(c); old = new
            it += 1
            if converged:
                break
    var = np.zeros((nc, D), f32)
    for c in range(nc):
        rows = X[cmap == c]
        d = (rows - ctr[c]).astype(np.float64)
        var[c] = (np.cumsum(d * d, axis=0)[-1] / float(csize[c])).astype(f32)
    return csize.copy(), ctr, var


def hinit(capi, mmf, name, tables, labels, max_iter=20, epsilon=1.0e-4, min_var=1.0e-2, min_seg=3):
    pk = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in mmf.packed().items()}
    h = mmf.logical[name]
    states = [int(s) for s in pk["hmmState"][pk["hmmStateOff"][h]:pk["hmmStateOff"][h + 1]]]
    comps = [list(range(int(pk["stateCompOff"][s]), int(pk["stateCompOff"][s + 1]))) for s in states]      # component indices per state
    gauss = [[int(pk["compGauss"][c]) for c in cs] for cs in comps]
    ne = len(states)
    N = ne + 2
    t = int(pk["hmmTrans"][h]); toff = int(pk["transOff"][t])
    segs = segments_of(name, tables, labels, ne)
    if len(segs) < min_seg:
        raise ValueError("HInit: only %d training tokens for %s (-m %d)" % (len(segs), name, min_seg))
    # one extra single-Gaussian scoring state per Gaussian of the set (no model uses them): MOutP of a component = score of its state
    S0, C0, G = int(pk["numStates"]), int(pk["numComp"]), int(pk["numGauss"])
    pk["stateCompOff"] = np.concatenate([pk["stateCompOff"], C0 + 1 + np.arange(G)]).astype(np.int32)
    pk["compGauss"] = np.concatenate([pk["compGauss"], np.arange(G)]).astype(np.int32)
    pk["compWeight"] = np.concatenate([pk["compWeight"], np.ones(G, np.float32)]).astype(np.float32)
    pk["numStates"], pk["numComp"] = S0 + G, C0 + G
    # ---- uniform segmentation (UCollectData): frame j (1-based) of a token of length L goes to state int((j-1)/(L/(N-2))) + 2
    by_state = [[] for _ in range(ne)]
    for X in segs:
        per = np.float32(X.shape[0]) / np.float32(ne)
        idx = (np.arange(X.shape[0], dtype=np.float32) / per).astype(np.int32)
        for j in range(ne):
            if (idx == j).any():
                by_state[j].append(X[idx == j])
    for j in range(ne):
        X = np.concatenate(by_state[j])
        size, ctr, var = flat_cluster(X, len(gauss[j]))                      # UniformSegment HInit.c:533-600
        for m, g in enumerate(gauss[j]):
            if len(gauss[j]) > 1:
                pk["compWeight"][comps[j][m]] = np.float32(size[m]) / np.float32(X.shape[0])
            pk["mean"][g] = ctr[m]
            pk["var"][g] = np.maximum(var[m], np.float32(min_var))
    model = capi.Model(pk)
    model.set_params(mean=pk["mean"], var=pk["var"], compWeight=pk["compWeight"])
    X = np.ascontiguousarray(np.concatenate(segs), np.float32)
    frameOff = np.concatenate([[0], np.cumsum([s.shape[0] for s in segs])]).astype(np.int32)
    labOff = np.arange(len(segs) + 1, dtype=np.int32)
    seq = np.full(len(segs), h, np.int32)
    dX = capi.DevArray(X)
    vit = capi.Viterbi(model)
    mode = capi.SCORE_SOUTP | capi.SCORE_DIAGC                                # OutP on a DIAGC set: SOutP over DOutP
    mixed = [j for j in range(ne) if len(gauss[j]) > 1]
    history, total, it, converged = [], np.float32(LZERO), 0, False
    while not converged and it < max_iter:
        it += 1
        al = vit.align(dX.ptr.value, frameOff, labOff, seq, scoreMode=mode)
        newP = np.float32(0.0)
        for a in al:
            if a["status"] != capi.UTT_OK:
                raise ValueError("HInit: no path found in a token of %s" % name)
            newP = np.float32(newP + np.float32(a["total"]))
        newP = np.float32(newP / np.float32(len(segs)))
        delta = np.float32(newP - total)
        converged = it > 1 and abs(float(delta)) < epsilon
        if not converged:
            # FindBestMixes: per-Gaussian scores of every frame for the components of the mixture states (first maximum wins)
            gsc = {}
            if mixed:
                glist = np.array(sorted({g for j in mixed for g in gauss[j]}), np.int32)
                sc = model.outp_block(X, (S0 + glist).astype(np.int32), mode=mode)                   # [frames, len(glist)]
                gsc = {int(g): sc[:, k] for k, g in enumerate(glist)}
            # UpdateCounts: frames of each state / component, transition counts along the state sequence (entry = 1, exit = N)
            by_comp = [[[] for _ in gauss[j]] for j in range(ne)]
            tran = np.zeros((N + 1, N + 1)); occ = np.zeros(N + 1)
            for u, (a, S) in enumerate(zip(al, segs)):
                last = 1
                for j in range(ne):
                    st, en = int(a["segStart"][j]), int(a["segEnd"][j])     # frames [st, en) of the token
                    if st < 0:
                        continue
                    if len(gauss[j]) == 1:
                        by_comp[j][0].append(S[st:en])
                    else:
                        f0 = int(frameOff[u])
                        P = np.stack([gsc[g][f0 + st:f0 + en] for g in gauss[j]], axis=1)
                        if not (P.max(axis=1) > np.float32(LZERO)).all():
                            raise ValueError("FindBestMixes: no best mix")
                        best = np.argmax(P, axis=1)
                        for m in range(len(gauss[j])):
                            if (best == m).any():
                                by_comp[j][m].append(S[st:en][best == m])
                    n = en - st
                    occ[last] += 1; tran[last][j + 2] += 1                  # entering state j+2
                    occ[j + 2] += n - 1; tran[j + 2][j + 2] += n - 1        # staying in it
                    last = j + 2
                occ[last] += 1; tran[last][N] += 1
            for j in range(ne):
                cnt = [sum(x.shape[0] for x in by_comp[j][m]) for m in range(len(gauss[j]))]
                for m, g in enumerate(gauss[j]):
                    if cnt[m] == 0:
                        raise ValueError("UpMeans: zero occ i=%d/s=1/m=%d" % (j + 2, m + 1))    # HError 2127
                    if len(gauss[j]) > 1:
                        pk["compWeight"][comps[j][m]] = np.float32(cnt[m]) / np.float32(sum(cnt))
                    pk["mean"][g], pk["var"][g] = _estimate(by_comp[j][m], pk["mean"][g].astype(np.float64), min_var)
            tp = pk["transP"][toff:toff + N * N].reshape(N, N)
            for i in range(1, N):                                           # UpTrans: rows renormalised, logs
                row = (tran[i, 2:N + 1] / occ[i]).astype(np.float32)
                s = np.float32(row.sum(dtype=np.float32))
                tp[i - 1, 0] = LZERO
                for j in range(2, N + 1):
                    x = np.float32(row[j - 2] / s)
                    tp[i - 1, j - 1] = LZERO if x < MINLARG else np.float32(np.log(np.float64(max(float(x), MINLARG))))
            model.set_params(mean=pk["mean"], var=pk["var"], transP=pk["transP"], compWeight=pk["compWeight"])
        total = newP
        history.append(float(newP))
    return pk, model, history, converged
