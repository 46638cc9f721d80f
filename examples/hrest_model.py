#!/usr/bin/env python3
"""Isolated-unit Baum-Welch re-estimation of ONE model (HTKTools/HRest.c) on the device.

Every training token of the model -- each label `name` of the transcriptions, frames floor(start/rate) .. floor(end/rate)
(LoadSegment HTrain.c:210), tokens shorter than the model's emitting states dropped (HRest.c:499) -- is an utterance whose
transcription is that one model; the forward-backward batch, the accumulators and the update are the library's (htkamd_fb_*,
htkamd_model_update with HRest's row-normalised transitions), iterated like ReEstimateModel (HRest.c:1309-1356) until the average
log probability per token changes by less than epsilon or maxIter passes are done.

    from examples.hrest_model import hrest
    model, history = hrest(capi, mmf, "S", tables, labels)     # tables: per file [T x D] features, labels: per file label list
"""
import numpy as np


def segments_of(name, tables, labels, n_emitting, frame_dur=100000, seg_reject=True):
    segs = []
    for X, labs in zip(tables, labels):
        for lab, start, end, _ in labs:
            if lab != name:
                continue
            st, en = int(start // frame_dur), int(end // frame_dur)
            en = min(en, X.shape[0] - 1)
            if st <= en and (en - st + 1 >= n_emitting or not seg_reject):
                segs.append(X[st:en + 1])
    return segs


def hrest(capi, mmf, name, tables, labels, max_iter=20, epsilon=1.0e-4, min_var=0.0, mix_weight_floor=0.0, min_seg=3, model=None):
    pk = mmf.packed()
    h = mmf.logical[name]
    n_emit = int(pk["hmmStateOff"][h + 1] - pk["hmmStateOff"][h])
    segs = segments_of(name, tables, labels, n_emit)
    if len(segs) < min_seg:
        raise ValueError("HRest: only %d training tokens for %s (-m %d)" % (len(segs), name, min_seg))
    model = model or capi.Model(pk)
    X = np.ascontiguousarray(np.concatenate(segs), np.float32)
    frameOff = np.concatenate([[0], np.cumsum([s.shape[0] for s in segs])]).astype(np.int32)
    labOff = np.arange(len(segs) + 1, dtype=np.int32)
    seq = np.full(len(segs), h, np.int32)
    dX = capi.DevArray(X)
    fb = capi.ForwardBackward(model)
    acc = capi.Accs(model)
    # HRest prunes nothing: no beam, and every state posterior counts (HERest's MINFORPROB cut is pushed out of the way)
    cfg = capi.fb_config(minFrwdP=700.0)
    history, old = [], np.float32(-1.0e10)
    for it in range(1, max_iter + 1):
        acc.zero()
        fb.prepare(dX.ptr.value, frameOff, labOff, seq)
        fb.execute(cfg, acc)
        pr, st = fb.results()
        ok = st == capi.UTT_OK
        if not ok.any():
            raise ValueError("HRest: no usable training token for %s" % name)
        a = acc.download()
        model.update(acc, a["vec"], minEgs=1, minVar=min_var, mixWeightFloor=mix_weight_floor, rowNormalise=True, singleProcess=True)
        new = np.float32(0.0)                                    # LogFloat newP, summed token by token (HRest.c:1311,1328)
        for p in pr[ok]:
            new = np.float32(new + np.float32(p))
        new = np.float32(new / np.float32(ok.sum()))
        history.append((float(new), int(ok.sum())))
        delta, old = np.float32(new - old), new
        if abs(float(delta)) < epsilon:
            break
    return model, history
