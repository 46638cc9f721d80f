"""Randomised parity sweep of the HIP path against the oracle (run on the GPU box; not part of the test-suite because of its
run time): forward-backward (pr, beams, accumulators) with random topologies / pruning / ragged batches, forced alignment
with random beams, network decoding over random word networks; multi-stream and tied-mixture sets.   python tests/fuzz_parity.py [iterations] [seed] [families, e.g. mfcc,quals]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from htk_amd import capi, synth  # noqa: E402
import pyoracle  # noqa: E402
from util import batch_arrays  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))) if a.size else 0.0


def fuzz_fb(rng, it):
    if rng.random() < 0.5:
        pk, names, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=int(rng.choice([13, 26, 39])), NU=int(rng.integers(2, 7)))
    else:
        big = rng.random() < 0.08                                   # now and then: long chains (up to 64 models, and beyond: general kernels)
        s = synth.generate(int(rng.integers(20, 80)), int(rng.integers(1, 13)), int(rng.integers(10, 40)), int(rng.integers(2, 6)),
                           int(rng.integers(600, 1000)) if big else int(rng.integers(30, 140)), int(rng.integers(1, 10**6)), D=int(rng.choice([13, 20, 39])))
        pk, seqs, feats = s.packed(), s.seqs, s.feats
    feats = [f[: max(3, len(f) - int(rng.integers(0, 10)))] for f in feats]          # ragged
    if rng.random() < 0.25:                                         # long chains: 2 / 4 / 8 wavefronts per utterance
        nq, nx = [], []
        for _ in range(int(rng.integers(2, 6))):
            tgt = int(rng.choice([rng.integers(55, 75), rng.integers(120, 136), rng.integers(65, 260), rng.integers(250, 290), rng.integers(300, 440)]))
            q, x = [], []
            while sum(len(a) for a in q) < tgt:
                k = int(rng.integers(0, len(seqs)))
                q.append(np.asarray(seqs[k], np.int32)); x.append(feats[k])
            nq.append(np.concatenate(q)); nx.append(np.concatenate(x))
        seqs, feats = nq, nx
    prune = {}
    r = rng.random()
    if r < 0.35:
        prune = dict(pruneInit=float(rng.uniform(20, 200)), pruneInc=0.0, pruneLim=0.0)
        prune["pruneLim"] = prune["pruneInit"]
    elif r < 0.6:
        prune = dict(pruneInit=float(rng.uniform(1, 30)), pruneInc=float(rng.uniform(5, 40)), pruneLim=float(rng.uniform(60, 300)))
    general = bool(rng.random() < 0.3) and max(len(q) for q in seqs) <= 150     # the general kernels hold up to 1024 model states per utterance
    mode = int(rng.random() < 0.3 and pk["vecSize"] in (13, 26, 39))
    if rng.random() < 0.25:
        mode |= 2                                                   # fast LAdd of the recursions (tolerance class)
    if rng.random() < 0.3 and (mode & 1):                           # (this draw once selected the removed scaled-linear mode)
        mode = (mode & ~1) | (4 if it % 2 else 32)                  # bf16 x 3 / fp16 x 2 matrix-core scores instead of the fp32 ones (the latter with |2: bench.py's default)
    model = capi.Model(pk); om = pyoracle.Model(pk)
    utts = [dict(seq=np.asarray(q, np.int32), feat=x) for q, x in zip(seqs, feats)]
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = capi.DevArray(X)
    fb = capi.ForwardBackward(model, debug=False, force_general=general, no_state_path=bool(rng.random() < 0.35))
    acc = capi.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    extra = {}
    if rng.random() < 0.3:
        extra["minFrwdP"] = float(rng.choice([3.0, 5.0, 20.0, 40.0]))
    if rng.random() < 0.3:
        extra["uFlags"] = int(rng.integers(1, 16))
    if rng.random() < 0.35:                                            # the pass in two phases, the states in random ranges in random order (htkamd_fb_execute_begin / _mix)
        if fb.execute_begin(capi.fb_config(scoreMode=mode, **prune, **extra), acc):
            S_ = int(pk["numStates"])
            cuts = sorted(set([0, S_] + [int(x) for x in rng.integers(0, S_ + 1, size=int(rng.integers(0, 4)))]))
            parts = list(zip(cuts[:-1], cuts[1:]))
            for k_ in rng.permutation(len(parts)):
                fb.execute_mix(parts[k_][0], parts[k_][1])
    else:
        fb.execute(capi.fb_config(scoreMode=mode, **prune, **extra), acc)
    pr, st = fb.results()
    a = acc.download()
    oacc = pyoracle.Accs(om)
    ocfg = pyoracle.fb_cfg(**prune, **extra)
    bad = []
    aborted = False
    for u, ut in enumerate(utts):
        rc, opr, _ = pyoracle.fb_utt(om, ocfg, ut["feat"], ut["seq"], oacc)
        ok_o = rc == 1
        aborted |= rc == -7390                                      # the reference's HERest stops here (fatal error in the alpha pass), having
                                                                    # accumulated part of the utterance: there is no accumulator state to compare
        if (st[u] == 1) != ok_o:
            bad.append("status u%d gpu %d oracle rc %d" % (u, st[u], rc))
        elif ok_o and abs(pr[u] - opr) > (1e-6 if mode else 1e-10) * abs(opr) and not (mode & 2 and prune):
            # (under a beam the tolerance-class LAdd may prune one model more or less than the reference: pr then moves by the pruned mass)
            bad.append("pr u%d %r vs %r" % (u, pr[u], opr))
    tol = 5e-4 if mode else 1e-4          # MFMA scores: posteriors near the MINFORPROB cut move by a few 1e-4 of small occupancies
    for k in ("muOcc", "wtOcc", "trOcc", "tr", "wt"):
        e = 0.0 if aborted else rel(a[k], getattr(oacc, k))
        if e > tol:
            bad.append("%s rel %.3g" % (k, e))
    if bad:
        print("FB  it %d general=%s mode=%d prune=%s extra=%s: %s" % (it, general, mode, prune, extra, "; ".join(bad)))
        import pickle
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)             # keep the case for a closer look
        pickle.dump(dict(pk=pk, utts=utts, prune=prune, extra=extra, general=general, mode=mode, bad=bad),
                    open(os.path.join(ROOT, "gpurun_out", "fuzz_fail_fb_%d.pkl" % it), "wb"))
    return not bad


def _random_widths(rng, D):
    S = int(rng.integers(2, 5))
    cuts = sorted(rng.choice(np.arange(1, D), size=S - 1, replace=False).tolist())
    return [b - a for a, b in zip([0] + cuts, cuts + [D])]


def _as_tied_mixture(pk, rng):
    """every (state, stream) lists its stream's pool (the components of state 0) with weights of its own"""
    NS, S = int(pk.get("numStreams", 1) or 1), int(pk["numStates"])
    sco, cg = np.asarray(pk["stateCompOff"]), np.asarray(pk["compGauss"])
    pools = [cg[sco[k]:sco[k + 1]] for k in range(NS)]
    pools = [p if len(p) > 1 else np.concatenate([p, cg[sco[NS + k]:sco[NS + k] + 1]]) for k, p in enumerate(pools)]      # a pool holds >= 2 Gaussians
    off, wt, ncg = [0], [], []
    for s_ in range(S):
        for k in range(NS):
            w = rng.random(len(pools[k])).astype(np.float32) + 0.05
            if rng.random() < 0.3:
                w[int(rng.integers(0, len(w)))] = 0.0                    # a pruned entry
            w /= w.sum()
            wt += list(w); ncg += list(pools[k]); off.append(len(wt))
    out = dict(pk)
    used = np.unique(np.concatenate(pools))
    remap = -np.ones(int(pk["numGauss"]), np.int64); remap[used] = np.arange(len(used))
    out.update(hsKind=1, stateCompOff=np.array(off, np.int32), compWeight=np.array(wt, np.float32), compGauss=remap[np.array(ncg)].astype(np.int32),
               mean=np.asarray(pk["mean"]).reshape(int(pk["numGauss"]), -1)[used], var=np.asarray(pk["var"]).reshape(int(pk["numGauss"]), -1)[used],
               gconst=None, numComp=len(wt), numGauss=len(used))
    return out


def fuzz_streams(rng, it):
    """Multi-stream and tied-mixture sets against the oracle: random stream widths, streams of one and of several
    components, tied-mixture pools, every recursion path, pruning, update flags."""
    import streams_util as su
    D = int(rng.choice([8, 13, 20, 26]))
    if rng.random() < 0.5:
        pk, names, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=D, NU=int(rng.integers(2, 6)))
    else:
        s = synth.generate(int(rng.integers(8, 30)), 1, int(rng.integers(6, 20)), int(rng.integers(2, 6)), int(rng.integers(30, 120)), int(rng.integers(1, 10**6)), D=D)
        pk, seqs, feats = s.packed(), s.seqs, s.feats
    kind = rng.choice(["ms", "ms", "tm1", "tmS"])
    if kind != "tm1":
        widths = _random_widths(rng, D)
        single = tuple(k for k in range(len(widths)) if rng.random() < 0.3)
        pk = su.make_multistream(pk, widths, rng, max_mix=int(rng.integers(1, 6)), single=single)
    else:
        pk = dict(pk)
    if kind.startswith("tm"):
        if kind == "tm1":                                            # one stream: give the states mixtures first
            pk = su.make_multistream(pk, [D], rng, max_mix=int(rng.integers(2, 7)))
            pk["numStreams"] = 1; pk["dimStream"] = None
            pk["var"] = np.where(np.isfinite(pk["var"]), pk["var"], 1.0)
        pk = _as_tied_mixture(pk, rng)
    prune = {}
    if rng.random() < 0.4:
        prune = dict(pruneInit=float(rng.uniform(30, 200)), pruneInc=0.0, pruneLim=0.0); prune["pruneLim"] = prune["pruneInit"]
    extra = {}
    if rng.random() < 0.3:
        extra["minFrwdP"] = float(rng.choice([5.0, 20.0]))
    if rng.random() < 0.3:
        extra["uFlags"] = int(rng.integers(1, 16))
    path = rng.choice(["state", "wave", "general"])
    # a third of the plain multi-stream cases with the reference's own second-visit arithmetic (htkamd_model_set_compat): against the oracle
    # WITHOUT ms_intended, retries of StepBack included
    compat = kind == "ms" and rng.random() < 0.33
    if compat and prune and rng.random() < 0.5:
        prune["pruneInc"] = float(rng.uniform(20, 80)); prune["pruneLim"] = prune["pruneInit"] + 4 * prune["pruneInc"]
    model = capi.Model(pk); om = pyoracle.Model(pk, ms_intended=not compat)
    if compat:
        model.set_compat(capi.COMPAT_STREAM_REVISIT)
    utts = [dict(seq=np.asarray(q, np.int32), feat=x[: max(3, len(x) - int(rng.integers(0, 6)))]) for q, x in zip(seqs, feats)]
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = capi.DevArray(X)
    fb = capi.ForwardBackward(model, debug=False, force_general=(path == "general"), no_state_path=(path == "wave"))
    acc = capi.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(capi.fb_config(**prune, **extra), acc)
    pr, st = fb.results()
    a = acc.download()
    oacc = pyoracle.Accs(om)
    bad, aborted = [], False
    for u, ut in enumerate(utts):
        rc, opr, _ = pyoracle.fb_utt(om, pyoracle.fb_cfg(**prune, **extra), ut["feat"], ut["seq"], oacc)
        aborted |= rc == -7390
        if (st[u] == 1) != (rc == 1):
            bad.append("status u%d gpu %d oracle rc %d" % (u, st[u], rc))
        elif rc == 1 and abs(pr[u] - opr) > 1e-7 * abs(opr):
            bad.append("pr u%d %r vs %r" % (u, pr[u], opr))
    for k in ("muOcc", "vaOcc", "wtOcc", "trOcc", "tr", "wt"):
        e = 0.0 if aborted else rel(a[k], getattr(oacc, k))
        if e > 1e-4:
            bad.append("%s rel %.3g" % (k, e))
    if bad:
        print("STREAMS it %d kind=%s path=%s compat=%s prune=%s extra=%s: %s" % (it, kind, path, compat, prune, extra, "; ".join(bad)))
        import pickle
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        pickle.dump(dict(pk=pk, utts=utts, prune=prune, extra=extra, path=str(path), bad=bad), open(os.path.join(ROOT, "gpurun_out", "fuzz_fail_streams_%d.pkl" % it), "wb"))
    return not bad


def fuzz_align(rng, it):
    if rng.random() < 0.3:                                          # mixed topologies incl. a tee model
        from types import SimpleNamespace
        pk, _, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=int(rng.choice([13, 39])), NU=int(rng.integers(2, 6)))
        s = SimpleNamespace(seqs=seqs, feats=feats)
    else:
        s = synth.generate(int(rng.integers(20, 80)), int(rng.integers(1, 5)), int(rng.integers(10, 40)), int(rng.integers(2, 6)),
                           int(rng.integers(900, 1100)) if rng.random() < 0.05 else int(rng.integers(40, 150)), int(rng.integers(1, 10**6)), D=int(rng.choice([13, 39])))
        pk = s.packed()
    if rng.random() < 0.2:                                          # long chains: 2 / 4 / 8 wavefronts per utterance, workgroup kernel beyond 512 models
        nq, nx = [], []
        for _ in range(int(rng.integers(2, 5))):
            tgt = int(rng.choice([rng.integers(55, 75), rng.integers(120, 136), rng.integers(65, 260), rng.integers(250, 560)]))
            q, x = [], []
            while sum(len(a) for a in q) < tgt:
                k = int(rng.integers(0, len(s.seqs)))
                q.append(np.asarray(s.seqs[k], np.int32)); x.append(s.feats[k])
            nq.append(np.concatenate(q)); nx.append(np.concatenate(x))
        s.seqs, s.feats = nq, nx
    beam = float(rng.choice([1.0e10, rng.uniform(5, 80)]))
    model = capi.Model(pk); om = pyoracle.Model(pk)
    utts = [dict(seq=np.asarray(q, np.int32), feat=x) for q, x in zip(s.seqs, s.feats)]
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = capi.DevArray(X)                              # must outlive the call
    got = capi.Viterbi(model).align(dX.ptr.value, frameOff, labOff, labs, genBeam=beam)
    ok = True
    for u, g in enumerate(got):
        r = pyoracle.viterbi_align(om, s.feats[u], utts[u]["seq"], genBeam=beam)
        if r is None:
            ok &= g["status"] == 0
            continue
        v = g["segStart"] >= 0
        good = bool(g["status"] == 1 and g["total"] == r["total"] and int(v.sum()) == int(r["n"]) and
                    np.array_equal(g["segStart"][v], r["start"][: r["n"]]) and np.array_equal(g["segScore"][v], r["score"][: r["n"]]))
        if not good:
            print("   u%d status %d total %r vs %r, n %d vs %d" % (u, g["status"], g["total"], r["total"], int(v.sum()), int(r["n"])))
        ok &= good
    if not ok:
        print("ALIGN it %d beam %g mismatch" % (it, beam))
    return ok


def fuzz_decode(rng, it, tmp):
    d = os.path.join(tmp, "d%d" % it); os.makedirs(d, exist_ok=True)
    tee = None
    if rng.random() < 0.3:                                          # mixed topologies with the tee model "sp" closing some pronunciations
        from types import SimpleNamespace
        pkt, names, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=13, NU=int(rng.integers(2, 5)))
        s = SimpleNamespace(seqs=seqs, feats=feats)
        synth.write_mmf_packed(os.path.join(d, "MMF"), pkt, names)
        NP = len(names); tee = names.index("sp")
    else:
        NP = int(rng.integers(6, 30))
        s = synth.generate(int(rng.integers(20, 60)), int(rng.integers(1, 4)), NP, int(rng.integers(2, 5)), int(rng.integers(40, 120)), int(rng.integers(1, 10**6)), D=13)
        synth.write_mmf(os.path.join(d, "MMF"), s, kind="USER")
        names = ["p%d" % i for i in range(NP)]
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    V = int(rng.integers(3, 12))
    words = []
    with open(os.path.join(d, "dict"), "w") as f:
        for w in range(V):
            for v in range(int(rng.integers(1, 3))):
                if tee is None:
                    ph = [names[int(k)] for k in rng.integers(0, NP, size=int(rng.integers(1, 4)))]
                else:                                                 # real models, optionally closed by the tee model (never two tee models in a row)
                    real = [k for k in range(NP) if k != tee]
                    ph = [names[int(rng.choice(real))] for _ in range(int(rng.integers(1, 4)))] + (["sp"] if rng.random() < 0.6 else [])
                pp = "" if rng.random() < 0.5 else "%.2f " % rng.uniform(0.1, 1.0)
                f.write("W%d %s%s\n" % (w, pp, " ".join(ph)))
            words.append("W%d" % w)
    arcs = []
    for w in range(V):
        arcs.append((0, 1 + w, float(np.log(1.0 / V))))
        for k in rng.choice(V, size=min(V, 3), replace=False):
            arcs.append((1 + w, 1 + int(k), float(np.log(rng.uniform(0.05, 0.5)))))
        arcs.append((1 + w, V + 1, float(np.log(0.1))))
    with open(os.path.join(d, "net.slf"), "w") as f:
        f.write("VERSION=1.0\nN=%d L=%d\nI=0 W=!NULL\n" % (V + 2, len(arcs)))
        for w in range(V):
            f.write("I=%d W=%s\n" % (1 + w, words[w]))
        f.write("I=%d W=!NULL\n" % (V + 1))
        for j, (a_, b_, l) in enumerate(arcs):
            f.write("J=%d S=%d E=%d l=%.4f\n" % (j, a_, b_, l))
    mmf = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    net = capi.Net(os.path.join(d, "net.slf"), os.path.join(d, "dict"), mmf)
    p = dict(genBeam=float(rng.choice([1.0e10, rng.uniform(20, 200)])), wordBeam=float(rng.choice([1.0e10, rng.uniform(10, 100)])),
             lmScale=float(rng.choice([1.0, rng.uniform(0.5, 8)])), wordPen=float(rng.choice([0.0, rng.uniform(-20, 10)])), prScale=float(rng.choice([1.0, rng.uniform(0.5, 3)])))
    if rng.random() < 0.3:
        p["maxActive"] = int(rng.integers(2, 30))                     # HVite -u
    model = capi.Model(mmf.packed()); om = pyoracle.Model(mmf.packed())
    res = capi.Decoder(model, net, lmScale=p["lmScale"]).run(s.feats, **p)
    ok = True
    keep = os.environ.get("HTKAMD_FUZZ_KEEP_DECODE")                 # e.g. "1840": copy that case's files next to the logs
    if keep and it in [int(k) for k in keep.split(",")]:
        import json, shutil
        dst = os.path.join(ROOT, "gpurun_out", "fuzz_decode_%d" % it)
        shutil.copytree(d, dst, dirs_exist_ok=True)
        np.savez(os.path.join(dst, "feats.npz"), **{"u%d" % u: x for u, x in enumerate(s.feats)})
        json.dump(dict(params=p, gpu=[[list(map(float, w)) for w in (r[0] or [])] for r in res], total=[float(r[1]) for r in res]),
                  open(os.path.join(dst, "case.json"), "w"))
    for u, (words_g, total) in enumerate(res):
        ow, ot = pyoracle.decode(om, s.feats[u], net.arrays(), **p)
        if words_g != ow or (ow is not None and total != ot):
            ok = False
            print("DECODE it %d u%d params %s\n  gpu    %s\n  oracle %s" % (it, u, p, words_g, ow))
    if rng.random() < 0.4:                   # token sets (HVite -n k): the kernel's lattice == the oracle's
        k = int(rng.integers(2, 9))
        al = int(rng.integers(1, 4)) if rng.random() < 0.35 else 0            # -m / -f with -n: alignment records inside the arcs
        lats = capi.Decoder(model, net, lmScale=p["lmScale"]).run_lattice(s.feats, k, align=al, **p)
        am = net.arrays()["model"]
        for u, got in enumerate(lats):
            ref = pyoracle.decode_nbest(om, s.feats[u], net.arrays(), k, align=al, **p)
            same = (got is None) == (ref is None)
            if same and got is not None:
                arcs = lambda l: sorted(zip(l["arcStart"].tolist(), l["arcEnd"].tolist(), l["arcAc"].tolist(), l["arcLm"].tolist(), l["arcPr"].tolist(), l["arcScore"].tolist()))
                same = got["total"] == ref["total"] and all(np.array_equal(got[f], ref[f]) for f in ("nodeFrame", "nodeNet", "nodeLike")) and arcs(got) == arcs(ref)
                if same and al:
                    def recs(l, mo):
                        return sorted((int(l["arcStart"][j]), int(l["arcEnd"][j]), float(l["arcScore"][j]),
                                       tuple((int(l["alState"][q]), int(mo(l, q)), int(l["alDur"][q]), float(l["alLike"][q])) for q in range(int(l["arcAlignOff"][j]), int(l["arcAlignOff"][j + 1]))))
                                      for j in range(len(l["arcStart"])))
                    same = recs(got, lambda l, q: l["alModel"][q]) == recs(ref, lambda l, q: am[l["alNode"][q]])
            if not same:
                ok = False
                print("NBEST it %d u%d k=%d params %s: kernel %s, oracle %s" % (it, u, k, p, None if got is None else (len(got["nodeFrame"]), len(got["arcStart"])),
                                                                             None if ref is None else (len(ref["nodeFrame"]), len(ref["arcStart"]))))
    return ok


def fuzz_mfcc(rng, it):
    """Random front-end configurations (the sweep of fuzz_oracle_vs_ref.py pins the oracle to HCopy on the same family):
    device MFCC of a ragged batch vs the oracle -- bit-equal but for the odd value where the device's double log() rounds the
    other way (tolerance class of SURVEY App. A: 1e-4 relative, 1e-3 absolute floor)."""
    from fuzz_oracle_vs_ref import mfcc_case
    kind, kw, _, rate = mfcc_case(rng)
    waves = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(rate // 20, rate))
        t = np.arange(n) / rate
        x = float(rng.uniform(500, 6000)) * np.sin(2 * np.pi * float(rng.uniform(100, 1500)) * t) * np.sin(2 * np.pi * 3 * t) + rng.normal(0, float(rng.uniform(50, 1500)), n)
        waves.append(x.clip(-32768, 32767).astype("<i2"))
    got, frameOff = capi.Mfcc(capi.mfcc_config(kind, **kw)).compute_host(waves)
    ocfg = pyoracle.mfcc_cfg(kind, **kw)
    refs = [pyoracle.mfcc(w, ocfg) for w in waves]
    ref = np.concatenate(refs) if refs else np.zeros((0, got.shape[1]), np.float32)
    ok = list(frameOff) == list(np.concatenate([[0], np.cumsum([r.shape[0] for r in refs])])) and got.shape == ref.shape
    ok = ok and bool(np.allclose(got, ref, rtol=1e-4, atol=1e-3)) and (ref.size == 0 or (got == ref).mean() > 0.995)
    if not ok:
        print("MFCC it %d %s %s: shapes %s %s, equal fraction %.5f, max abs %.3g" % (it, kind, kw, got.shape, ref.shape, (got == ref).mean() if got.shape == ref.shape and ref.size else -1,
                                                                                   np.abs(got - ref).max() if got.shape == ref.shape and ref.size else -1))
    return ok


def fuzz_quals(rng, it):
    """Random qualifier steps (the family fuzz_oracle_vs_ref.py pins against HCopy, plus _N) on ragged batches with very short
    tables: htkamd_parm_qualify == oracle, bit for bit."""
    from fuzz_oracle_vs_ref import quals_case
    kind, kw, _ = quals_case(rng)
    null = 12 if rng.random() < 0.3 else -1
    utts = [rng.normal(0, 3, size=(int(rng.choice([1, 2, 3, 4, 5, 7, 9, int(rng.integers(10, 200))])), 13)).astype(np.float32) for _ in range(int(rng.integers(1, 6)))]
    q = capi.ParmQuals(13, kw["nZeroMean"], 1, int(kw["hasA"]), int(kw["hasT"]), kw["delWin"], kw["accWin"], kw["thirdWin"], null, int(kw["v1Compat"]), int(kw["simpleDiffs"]))
    d, frameOff, cols = capi.parm_qualify(utts, q)
    got = d.to_host(np.float32, (int(frameOff[-1]), cols))
    ok = True
    for u, x in enumerate(utts):
        ref = pyoracle.parm_qualify(x, nullECol=null, **kw)
        if not np.array_equal(got[frameOff[u]:frameOff[u + 1]], ref):
            ok = False
            print("QUALS it %d %s %s null %d: utterance %d (T=%d) differs" % (it, kind, kw, null, u, x.shape[0]))
    return ok


def fuzz_herest_cli(rng, it, tmp):
    """tools/bin/herest against the reference's HERest binary (oracle/_ref, present on the GPU box with the tree) on random sets: one pass,
    ML (-u tmvw, random -v / -w / -m) or MAP (-u pmvw / pmv / pm with a random HMAP: MAPTAU, MINVAR, MIXWEIGHTFLOOR), every number of the
    two MMFs to 2e-4 relative (the files carry 7 digits; means relative to max(|mean|, sigma))."""
    import subprocess
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "HERest"); exe = os.path.join(ROOT, "tools", "bin", "herest")
    if not (os.path.exists(ref_exe) and os.path.exists(exe)):
        return True
    d = os.path.join(tmp, "h%d" % it); os.makedirs(os.path.join(d, "ours")); os.makedirs(os.path.join(d, "ref"))
    s = synth.generate(int(rng.integers(10, 30)), int(rng.integers(1, 4)), int(rng.integers(5, 12)), int(rng.integers(8, 16)),
                       int(rng.integers(60, 140)), int(rng.integers(1, 10**6)), D=13)
    pk = s.packed()
    names = ["p%d" % i for i in range(pk["numPhys"])]
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    scp = []
    for u, (X, q) in enumerate(zip(s.feats, s.seqs)):
        fn = os.path.join(d, "u%d.mfc" % u)
        synth.write_htk_param(fn, X, kind=9); scp.append(fn)
        open(os.path.join(d, "u%d.lab" % u), "w").write("\n".join(names[int(h)] for h in q) + "\n")
    conf = ""
    if rng.random() < 0.5:
        flags = str(rng.choice(["pmvw", "pmv", "pm", "pmw"]))
        conf = "HMAP: MAPTAU = %g\nHMAP: MINVAR = %g\n" % (float(rng.choice([0.5, 4.0, 20.0, 60.0])), float(rng.choice([0.0, 0.02, 0.3])))
        if rng.random() < 0.5:
            conf += "HMAP: MIXWEIGHTFLOOR = %g\n" % float(rng.choice([1.0, 3.0]))
        opts = ["-u", flags]
    else:
        opts = ["-u", str(rng.choice(["tmvw", "mv", "tw", "tmv"])), "-v", "%g" % float(rng.choice([0.0, 0.05, 0.5])), "-w", "%g" % float(rng.choice([0.0, 2.0])),
                "-m", "%d" % int(rng.choice([1, 3]))]
    if rng.random() < 0.4:
        opts += ["-t", "%g" % float(rng.choice([150.0, 400.0]))]
    open(os.path.join(d, "config"), "w").write(conf)
    out = []
    for e, sub in ((ref_exe, "ref"), (exe, "ours")):
        r = subprocess.run([e, "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-M", os.path.join(d, sub), "-L", d] + opts + [os.path.join(d, "hmmlist")] + scp,
                           capture_output=True, text=True)
        out.append(r)
    if out[0].returncode != 0 or out[1].returncode != 0:
        if out[0].returncode != 0 and out[1].returncode != 0:
            return True                                              # both refuse (e.g. a model without enough examples made the pass fail)
        print("HEREST it %d opts %s conf %r: rc ref %d ours %d\n  %s\n  %s" % (it, opts, conf, out[0].returncode, out[1].returncode, out[0].stdout[-200:], (out[1].stdout + out[1].stderr)[-200:]))
        return False
    a = capi.Mmf(files=[os.path.join(d, "ours", "MMF")], hmm_list=os.path.join(d, "hmmlist")).packed()
    b = capi.Mmf(files=[os.path.join(d, "ref", "MMF")], hmm_list=os.path.join(d, "hmmlist")).packed()
    ok = a["numComp"] == b["numComp"] and a["numGauss"] == b["numGauss"]
    if ok:
        sig = np.sqrt(np.maximum(b["var"], 1e-12))
        em = np.max(np.abs(a["mean"] - b["mean"]) / np.maximum(np.abs(b["mean"]), sig))
        # (a variance that comes out near zero is the difference of two large float sums in the reference: compared on the scale of
        #  the variances the data were drawn with, 0.5 .. 2)
        ev = np.max(np.abs(a["var"] - b["var"]) / np.maximum(np.abs(b["var"]), 5e-2))
        ew = np.max(np.abs(a["compWeight"] - b["compWeight"]))
        lin = lambda v: np.where(np.asarray(v) > -0.5e10, np.exp(np.asarray(v, np.float64)), 0.0)
        et = np.max(np.abs(lin(a["transP"]) - lin(b["transP"])))
        ok = em <= 2e-4 and ev <= 2e-4 and ew <= 2e-5 and et <= 2e-5
        if not ok:
            print("HEREST it %d opts %s conf %r: mean %.3g var %.3g weight %.3g trans %.3g" % (it, opts, conf, em, ev, ew, et))
    else:
        print("HEREST it %d opts %s: different structure" % (it, opts))
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return ok


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    tmp = tempfile.mkdtemp()
    res = dict(fb=[0, 0], streams=[0, 0], align=[0, 0], decode=[0, 0], mfcc=[0, 0], quals=[0, 0], herest=[0, 0])
    only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None          # e.g. "mfcc,quals": these families only
    for it in range(n):
        for name, fn in (("fb", lambda: fuzz_fb(rng, it)), ("streams", lambda: fuzz_streams(rng, it)), ("align", lambda: fuzz_align(rng, it)), ("decode", lambda: fuzz_decode(rng, it, tmp)),
                         ("mfcc", lambda: fuzz_mfcc(rng, it)), ("quals", lambda: fuzz_quals(rng, it)), ("herest", lambda: fuzz_herest_cli(rng, it, tmp))):
            if only is not None and name not in only:
                continue
            try:
                ok = fn()
            except Exception as e:  # noqa: BLE001
                ok = False
                print("%s it %d raised %r" % (name, it, e))
            res[name][0] += 1; res[name][1] += int(ok)
    print("passed/total:", {k: "%d/%d" % (v[1], v[0]) for k, v in res.items()})
    sys.exit(0 if all(v[0] == v[1] for v in res.values()) else 1)


if __name__ == "__main__":
    main()
