"""Randomised check of the ORACLE against the reference's own binaries (CPU only; needs oracle/_ref, i.e. this container):
   decode : random model sets / dictionaries (pronunciation variants, probabilities, tee model) / lattices / switches ->
            the reference's HVite label files vs oracle.decode
   fb     : random sets, utterances and pruning -> the reference HERest's per-utterance "Utterance prob per frame" trace and the
            utterances it skips vs oracle.fb_utt
   align  : HVite -a -f -m label files (state and model level, tee models, beams) vs oracle.viterbi_align
   update : the models a HERest pass writes (random -v / -w / -m) vs oracle F-B + oracle MLUpdateModels
   acc    : the HER1.acc a `HERest -p 1` pass dumps (random pruning / update flags) vs the oracle's accumulators written by OUR
            writer: the two files byte for byte
   mmf    : our MMF writer (text, binary) through the reference's HHEd and back; HHEd's binary through our reader
   quals  : the qualifier step on parameter files (_D _A _T _Z, windows, V1COMPAT, SIMPLEDIFFS, very short files) vs oracle.parm_qualify
   mfcc   : the file HCopy codes from a WAV under a random front-end configuration vs oracle.mfcc, every float
The GPU sweep (tests/fuzz_parity.py) compares the HIP path with the oracle; this one keeps the oracle honest.
    python tests/fuzz_oracle_vs_ref.py [iterations] [seed]"""
import os
import re
import subprocess
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from htk_amd import capi, synth  # noqa: E402   (host-side readers only: no GPU is touched)
import pyoracle  # noqa: E402
from decode_util import format_words  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def _lat_same(a, b, xwrd=False):
    """Two SLF files: identical, byte for byte.  (Until round 4 this accepted "the same lattice up to the float noise of relative tokens":
    a relative token's likelihood is a float re-based at every TokSetMerge, HRec.c:361-364, so the order in which tokens arrive at a node
    is in its last bits -- and the oracle pulled in node order.  It walks HRec's instance list now, oracle/orc_ilist.h; cross-word
    networks, which kept the allowance a little longer, have the reference's nodes since net.c types null nodes and the word ends in
    front of a final null node as ProcessCrossWordLinks does.)"""
    return a == b


def _labels_same(got, want):
    return want is not None and got == want


def fuzz_decode(rng, it, tmp):
    d = os.path.join(tmp, "d%d" % it); os.makedirs(d, exist_ok=True)
    tee = None
    if rng.random() < 0.4:
        pk, names, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=13, NU=int(rng.integers(2, 4)))
        s = SimpleNamespace(seqs=seqs, feats=feats)
        synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
        NP = len(names); tee = names.index("sp")
    else:
        NP = int(rng.integers(6, 20))
        s = synth.generate(int(rng.integers(20, 50)), int(rng.integers(1, 4)), NP, int(rng.integers(2, 4)), int(rng.integers(40, 100)), int(rng.integers(1, 10**6)), D=13)
        synth.write_mmf(os.path.join(d, "MMF"), s, kind="USER")
        names = ["p%d" % i for i in range(NP)]
    wint = tee is None and rng.random() < 0.35                 # word-internal context expansion over a triphone-style logical list
    phones4, ctx = ["a", "b", "c", "d"], None
    xwrd = wint and rng.random() < 0.5                          # cross-word context expansion (FORCECXTEXP = T, ALLOWXWRDEXP = T)
    if xwrd:
        ctx = phones4 + ["sil"]                                  # every word must define some context: sil is one, sp is context free
        with open(os.path.join(d, "hmmlist"), "w") as f:
            f.write("\n".join(names) + "\n")
            for x in phones4:
                for l_ in [None] + ctx:
                    for r_ in [None] + ctx:
                        f.write("%s%s%s %s\n" % ("%s-" % l_ if l_ else "", x, "+%s" % r_ if r_ else "", names[int(rng.integers(0, NP))]))
            f.write("sil %s\nsp %s\n" % (names[int(rng.integers(0, NP))], names[int(rng.integers(0, NP))]))
    elif wint:
        ctx = phones4 + (["sp"] if rng.random() < 0.5 else [])   # sp may or may not be somebody's context
        with open(os.path.join(d, "hmmlist"), "w") as f:
            f.write("\n".join(names) + "\n")
            for x in phones4:
                for l_ in [None] + ctx:
                    for r_ in [None] + ctx:
                        if l_ is None and r_ is None and rng.random() < 0.5:
                            continue                               # the bare model is there only sometimes
                        f.write("%s%s%s %s\n" % ("%s-" % l_ if l_ else "", x, "+%s" % r_ if r_ else "", names[int(rng.integers(0, NP))]))
            f.write("sil %s\nsp %s\n" % (names[int(rng.integers(0, NP))], names[int(rng.integers(0, NP))]))
    else:
        open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    V = int(rng.integers(3, 10))
    words = ["W%d" % w for w in range(V)]
    with open(os.path.join(d, "dict"), "w") as f:
        for w in range(V):
            for v in range(int(rng.integers(1, 3))):
                if xwrd:
                    ph = [phones4[int(k)] for k in rng.integers(0, 4, size=int(rng.integers(1, 5)))]        # one-phone words: the (lc, rc) cross-bar
                    if len(ph) > 2 and rng.random() < 0.15:
                        ph.insert(int(rng.integers(1, len(ph))), "sp")                                      # a context-free phone inside the word
                    ph = (["sp"] if rng.random() < 0.1 else []) + ph + (["sp"] if rng.random() < 0.6 else [])
                    if w == 0:
                        ph = ["sil"]
                elif wint:
                    ph = [phones4[int(k)] for k in rng.integers(0, 4, size=int(rng.integers(2, 5)))] + (["sp"] if rng.random() < 0.6 else [])
                    if w == 0:
                        ph = ["sil"]
                elif tee is None:
                    ph = [names[int(k)] for k in rng.integers(0, NP, size=int(rng.integers(1, 4)))]
                else:
                    real = [k for k in range(NP) if k != tee]
                    ph = [names[int(rng.choice(real))] for _ in range(int(rng.integers(1, 4)))] + (["sp"] if rng.random() < 0.6 else [])
                out = "" if rng.random() < 0.7 else ("[] " if rng.random() < 0.3 else "[o%d] " % w)
                pp = "" if rng.random() < 0.5 else "%.2f " % rng.uniform(0.1, 1.0)
                f.write("W%d %s%s%s\n" % (w, out, pp, " ".join(ph)))
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
    p = dict(genBeam=float(rng.choice([1.0e10, round(rng.uniform(20, 200), 2)])), wordBeam=float(rng.choice([1.0e10, round(rng.uniform(10, 100), 2)])),
             lmScale=float(rng.choice([1.0, round(rng.uniform(0.5, 8), 2)])), wordPen=float(rng.choice([0.0, round(rng.uniform(-20, 10), 2)])),
             prScale=float(rng.choice([1.0, round(rng.uniform(0.5, 3), 2)])))
    scp = []
    for u, X in enumerate(s.feats):
        fn = os.path.join(d, "u%d.mfc" % u)
        synth.write_htk_param(fn, X, kind=9)
        scp.append(fn)
    open(os.path.join(d, "scp"), "w").write("\n".join(scp) + "\n")
    open(os.path.join(d, "config"), "w").write("FORCECXTEXP = T\nALLOWXWRDEXP = T\n" if xwrd else "")
    opts = []
    if p["genBeam"] < 1e9: opts += ["-t", "%.2f" % p["genBeam"]]
    if p["wordBeam"] < 1e9: opts += ["-v", "%.2f" % p["wordBeam"]]
    opts += ["-s", "%.2f" % p["lmScale"], "-p", "%.2f" % p["wordPen"], "-r", "%.2f" % p["prScale"]]
    if rng.random() < 0.35:                                      # maximum-model pruning (HRec.c:1966-1985); counts instances
        p["maxActive"] = int(rng.integers(2, 25))
        opts += ["-u", str(p["maxActive"])]
    mlf = os.path.join(d, "out.mlf")
    r = subprocess.run([os.path.join(REF, "HVite"), "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-S", os.path.join(d, "scp"), "-i", mlf,
                        "-w", os.path.join(d, "net.slf")] + opts + [os.path.join(d, "dict"), os.path.join(d, "hmmlist")], capture_output=True, text=True)
    ref = {}
    if os.path.exists(mlf):
        cur = None
        for line in open(mlf).read().splitlines()[1:]:
            if line.startswith('"'):
                cur = os.path.basename(line.strip('"')).replace(".rec", ""); ref[cur] = []
            elif line == ".":
                cur = None
            elif cur is not None:
                ref[cur].append(line)
    mmf = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    net = capi.Net(os.path.join(d, "net.slf"), os.path.join(d, "dict"), mmf, flags=(capi.NET_ALLOWXWRDEXP | capi.NET_FORCECXTEXP) if xwrd else 0)
    assert net.xwrd == xwrd
    om = pyoracle.Model(mmf.packed())
    ok = True
    for u, X in enumerate(s.feats):
        ow, ot = pyoracle.decode(om, X, net.arrays(), **p)
        got = None if ow is None else format_words(ow, net.out_syms)
        want = ref.get("u%d" % u)
        if got != want and not (got is None and want is None):
            # "No tokens survived" utterances: the reference writes no entry
            ok = False
            print("DECODE it %d u%d params %s rc %d\n  oracle %s\n  HVite  %s" % (it, u, p, r.returncode, got, want))
    if ok and rng.random() < 0.4:
        # ---- N-best: HVite -n k 1 -z lat (lattice files) and -n k m (alternative transcriptions) against the oracle's token sets put
        # through the product's host code (htkamd_lattice_write / htkamd_lattice_nbest); run inside d with relative names so that the
        # header lines of the lattices agree
        k, mtr = int(rng.integers(2, 7)), int(rng.integers(2, 5))
        rel = [os.path.basename(f) for f in scp]
        open(os.path.join(d, "scp_rel"), "w").write("\n".join(rel) + "\n")
        # a third of the cases with -m / -f: alignment records inside the arcs (LatFromPaths' lAlign, the reference built with -DPHNALG)
        al_mode = int(rng.integers(1, 4)) if rng.random() < 0.33 else 0
        al_opts = (["-m"] if al_mode & 1 else []) + (["-f"] if al_mode & 2 else [])
        base = [os.path.join(REF, "HVite"), "-C", "config", "-H", "MMF", "-S", "scp_rel", "-w", "net.slf"] + opts + al_opts
        subprocess.run(base + ["-l", ".", "-n", str(k), "1", "-z", "lat", "dict", "hmmlist"], cwd=d, capture_output=True, text=True)
        subprocess.run(base + ["-i", "nb.mlf", "-n", str(k), str(mtr), "dict", "hmmlist"], cwd=d, capture_output=True, text=True)
        nb, cur = {}, None
        if os.path.exists(os.path.join(d, "nb.mlf")):
            for line in open(os.path.join(d, "nb.mlf")).read().splitlines()[1:]:
                if line.startswith('"'):
                    cur = os.path.basename(line.strip('"')).replace(".rec", ""); nb[cur] = [[]]
                elif line == ".":
                    cur = None
                elif line == "///":
                    nb[cur].append([])
                elif cur is not None:
                    nb[cur][-1].append(line)
        arr = net.arrays()
        for u, X in enumerate(s.feats):
            lat = pyoracle.decode_nbest(om, X, arr, k, align=al_mode, **p)
            ref_lat = os.path.join(d, "u%d.lat" % u)
            if lat is None:
                if os.path.exists(ref_lat):
                    ok = False; print("NBEST it %d u%d: oracle has no lattice, HVite wrote one" % (it, u))
                continue
            lat["nodePron"] = np.array([arr["model"][n] if n >= 0 else -1 for n in lat["nodeNet"]], np.int32)
            lat.update(lmScale=p["lmScale"], wordPen=p["wordPen"], prScale=p["prScale"])
            mine = os.path.join(d, "u%d.mylat" % u)
            if al_mode:
                lat["alModel"] = np.array([arr["model"][n] for n in lat["alNode"]], np.int32); lat["alignModels"] = bool(al_mode & 1)
            capi.lattice_write(lat, net, mine, utterance=rel[u], lm_name="net.slf", vocab_name="dict", mmf=mmf if al_mode else None)
            if not os.path.exists(ref_lat) or not _lat_same(open(mine).read(), open(ref_lat).read(), xwrd):
                ok = False; print("NBEST it %d u%d k=%d params %s: lattice files differ (%s)" % (it, u, k, p, d))
            if al_mode:
                got = capi.lattice_nbest_align(lat, net, mmf, mtr, states=bool(al_mode & 2), models=bool(al_mode & 1))
            else:
                alts = capi.lattice_nbest(lat, net, mtr)
                got = [["%d %d %s %f" % (st_ * 100000, en_ * 100000, net.out_syms[w], np.float32(sc)) for w, st_, en_, sc in a if w >= 0 and net.out_syms[w] != ""] for a in alts]
            if open(mine).read() == open(ref_lat).read() and not _labels_same(got, nb.get("u%d" % u)):
                ok = False; print("NBEST it %d u%d k=%d m=%d params %s:\n  oracle %s\n  HVite  %s" % (it, u, k, mtr, p, got, nb.get("u%d" % u)))
    if not ok and os.environ.get("FUZZ_KEEP"):
        import json, shutil
        json.dump(p, open(os.path.join(d, "params.json"), "w"))
        shutil.copytree(d, os.path.join(os.environ["FUZZ_KEEP"], "decode_%d" % it), dirs_exist_ok=True)
    return ok


def fuzz_fb(rng, it, tmp):
    d = os.path.join(tmp, "f%d" % it); os.makedirs(os.path.join(d, "out"), exist_ok=True)
    if rng.random() < 0.5:
        pk, names, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=13, NU=int(rng.integers(2, 5)))
    else:
        s = synth.generate(int(rng.integers(20, 50)), int(rng.integers(1, 5)), int(rng.integers(8, 25)), int(rng.integers(2, 5)),
                           int(rng.integers(30, 120)), int(rng.integers(1, 10**6)), D=13)
        pk, seqs, feats = s.packed(), s.seqs, s.feats
        names = ["p%d" % i for i in range(pk["numPhys"])]
    feats = [f[: max(3, len(f) - int(rng.integers(0, 10)))] for f in feats]
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    scp = []
    for u, (X, q) in enumerate(zip(feats, seqs)):
        fn = os.path.join(d, "u%d.mfc" % u)
        synth.write_htk_param(fn, X, kind=9); scp.append(fn)
        open(os.path.join(d, "u%d.lab" % u), "w").write("\n".join(names[int(h)] for h in q) + "\n")
    open(os.path.join(d, "config"), "w").write("")
    prune, opts = {}, []
    r_ = rng.random()
    if r_ < 0.35:
        t = round(float(rng.uniform(20, 200)), 2)
        prune = dict(pruneInit=t, pruneInc=0.0, pruneLim=t); opts = ["-t", "%.2f" % t]
    elif r_ < 0.6:
        a, b, c = round(float(rng.uniform(1, 30)), 2), round(float(rng.uniform(5, 40)), 2), round(float(rng.uniform(60, 300)), 2)
        prune = dict(pruneInit=a, pruneInc=b, pruneLim=c); opts = ["-t", "%.2f" % a, "%.2f" % b, "%.2f" % c]
    r = subprocess.run([os.path.join(REF, "HERest"), "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-M", os.path.join(d, "out"), "-L", d, "-T", "1", "-m", "0"] +
                       opts + [os.path.join(d, "hmmlist")] + scp, capture_output=True, text=True)
    # per file: the LAST "Utterance prob per frame" after its "Processing Data" line, or a skip
    per, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.search(r"Processing Data: (\S+?)\.mfc", line)
        if m:
            cur = os.path.basename(m.group(1)); per[cur] = None
        m = re.search(r"Utterance prob per frame = (\S+)", line)
        if m and cur:
            per[cur] = m.group(1)
    crashed = r.returncode != 0
    om = pyoracle.Model(pk); oacc = pyoracle.Accs(om); cfg = pyoracle.fb_cfg(**prune)
    ok = True
    for u, (X, q) in enumerate(zip(feats, seqs)):
        rc, opr, _ = pyoracle.fb_utt(om, cfg, X, np.asarray(q, np.int32), oacc)
        want = per.get("u%d" % u)
        if rc in (1, -7390):                                   # -7390: the beta pass succeeded (prob printed), the alpha pass aborts the reference
            got = "%e" % (opr / X.shape[0])
            if want != got and not (crashed and want is None):
                ok = False
                print("FB it %d u%d prune %s: oracle rc %d %s, HERest %s (exit %d)" % (it, u, prune, rc, got, want, r.returncode))
        elif want is not None and not crashed:
            ok = False
            print("FB it %d u%d prune %s: oracle rc %d, HERest printed %s" % (it, u, prune, rc, want))
        if rc == -7390 and not crashed:
            ok = False
            print("FB it %d u%d: oracle says alpha failure, HERest finished normally" % (it, u))
    return ok


def fuzz_align(rng, it, tmp):
    """HVite -a -f -m (phone-level transcriptions, DoAlignment): state/model label lines vs oracle.viterbi_align."""
    d = os.path.join(tmp, "a%d" % it); os.makedirs(d, exist_ok=True)
    if rng.random() < 0.5:
        pk, names, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=13, NU=int(rng.integers(2, 5)))
        # a tee model as a WORD of its own is fatal in the reference when it is skipped ("LatFromPaths: Align have dur<=0"):
        # the transcriptions keep the other five topologies (single-state, skip, tied state)
        tee = names.index("sp")
        seqs = [[h for h in q if h != tee] for q in seqs]
    else:
        s = synth.generate(int(rng.integers(20, 50)), int(rng.integers(1, 4)), int(rng.integers(8, 25)), int(rng.integers(2, 5)),
                           int(rng.integers(40, 120)), int(rng.integers(1, 10**6)), D=13)
        pk, seqs, feats = s.packed(), s.seqs, s.feats
        names = ["p%d" % i for i in range(pk["numPhys"])]
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    open(os.path.join(d, "dict"), "w").write("".join("%s %s\n" % (n, n) for n in sorted(names)))
    scp = []
    for u, (X, q) in enumerate(zip(feats, seqs)):
        fn = os.path.join(d, "u%d.mfc" % u)
        synth.write_htk_param(fn, X, kind=9); scp.append(fn)
        open(os.path.join(d, "u%d.lab" % u), "w").write("\n".join(names[int(h)] for h in q) + "\n")
    open(os.path.join(d, "config"), "w").write("")
    beam = float(rng.choice([1.0e10, round(float(rng.uniform(5, 80)), 2)]))
    opts = ["-t", "%.2f" % beam] if beam < 1e9 else []
    mlf = os.path.join(d, "out.mlf")
    subprocess.run([os.path.join(REF, "HVite"), "-a", "-f", "-m", "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-L", d, "-i", mlf] + opts +
                   [os.path.join(d, "dict"), os.path.join(d, "hmmlist")] + scp, capture_output=True, text=True)
    ref, cur = {}, None
    if os.path.exists(mlf):
        for line in open(mlf).read().splitlines()[1:]:
            if line.startswith('"'):
                cur = os.path.basename(line.strip('"')).replace(".rec", ""); ref[cur] = []
            elif line == ".":
                cur = None
            elif cur is not None:
                ref[cur].append(line)
    om = pyoracle.Model(pk)
    ok = True
    for u, (X, q) in enumerate(zip(feats, seqs)):
        r = pyoracle.viterbi_align(om, X, np.asarray(q, np.int32), genBeam=beam)
        got = None if r is None else pyoracle.format_rec(r, np.asarray(q, np.int32), names)
        want = ref.get("u%d" % u)
        if got != want:
            ok = False
            print("ALIGN it %d u%d beam %g\n  oracle %s\n  HVite  %s" % (it, u, beam, None if got is None else got[:4], None if want is None else want[:4]))
    return ok


def fuzz_update(rng, it, tmp):
    """One HERest pass end to end: the models the reference writes vs oracle F-B + oracle MLUpdateModels (to the 7 printed digits)."""
    d = os.path.join(tmp, "p%d" % it); os.makedirs(os.path.join(d, "out"), exist_ok=True)
    s = synth.generate(int(rng.integers(10, 30)), int(rng.integers(1, 4)), int(rng.integers(5, 12)), int(rng.integers(6, 14)),
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
    open(os.path.join(d, "config"), "w").write("")
    minVar = float(rng.choice([0.0, 0.05, 0.5])); wf = float(rng.choice([0.0, 2.0, 5.0])); minEgs = int(rng.choice([1, 3]))
    r = subprocess.run([os.path.join(REF, "HERest"), "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-M", os.path.join(d, "out"), "-L", d,
                        "-v", "%g" % minVar, "-w", "%g" % wf, "-m", "%d" % minEgs, os.path.join(d, "hmmlist")] + scp, capture_output=True, text=True)
    if r.returncode != 0:
        print("UPDATE it %d: HERest failed: %s" % (it, r.stdout[-300:])); return False
    pkr = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist")).packed()    # what the reference loaded (7 digits)
    om = pyoracle.Model(pkr); oacc = pyoracle.Accs(om)
    for X, q in zip(s.feats, s.seqs):
        pyoracle.fb_utt(om, pyoracle.fb_cfg(), X, np.asarray(q, np.int32), oacc)
    pyoracle.update(om, oacc, minEgs=minEgs, minVar=minVar, mixWeightFloor=wf * 1.0e-5, singleProcess=True)
    ref = capi.Mmf(files=[os.path.join(d, "out", "MMF")], hmm_list=os.path.join(d, "hmmlist")).packed()
    ok = True
    # component by component (a defunct mixture is left out of the file, so Gaussian numbers differ between the two loads)
    aw, bw = np.asarray(om.compWeight, np.float64), np.asarray(ref["compWeight"], np.float64)
    if np.max(np.abs(aw - bw)) > 2e-6:
        ok = False; print("UPDATE it %d: weights differ by %.3g" % (it, np.max(np.abs(aw - bw))))
    for c in range(pkr["numComp"]):
        if bw[c] <= 1.0e-5:
            continue
        g, rg = int(pkr["compGauss"][c]), int(ref["compGauss"][c])
        for k in ("mean", "var"):
            a, b = np.asarray(getattr(om, k), np.float64)[g], np.asarray(ref[k], np.float64)[rg]
            e = np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))
            if e > 2e-6:
                ok = False; print("UPDATE it %d: %s of component %d differs by %.3g (minVar %g w %g m %d)" % (it, k, c, e, minVar, wf, minEgs))
    lin = lambda v: np.where(np.asarray(v) > -0.5e10, np.exp(np.asarray(v, np.float64)), 0.0)
    e = np.max(np.abs(lin(om.transP) - lin(ref["transP"])))
    if e > 2e-6:
        ok = False; print("UPDATE it %d: transP differs by %.3g" % (it, e))
    if not ok and os.environ.get("FUZZ_KEEP"):
        import json, shutil
        json.dump(dict(minVar=minVar, wf=wf, minEgs=minEgs), open(os.path.join(d, "params.json"), "w"))
        shutil.copytree(d, os.path.join(os.environ["FUZZ_KEEP"], "update_%d" % it), dirs_exist_ok=True)
    return ok


def fuzz_acc(rng, it, tmp):
    """HERest -p 1 (accumulate and dump, HERest.c:600-640 + HTrain.c DumpAccs) vs oracle accumulators through htkamd_accs_dump_file."""
    d = os.path.join(tmp, "a%d" % it); os.makedirs(os.path.join(d, "out"), exist_ok=True)
    s = synth.generate(int(rng.integers(6, 20)), int(rng.integers(1, 4)), int(rng.integers(4, 10)), int(rng.integers(5, 12)),
                       int(rng.integers(50, 120)), int(rng.integers(1, 10**6)), D=int(rng.choice([5, 13])))
    pk0 = s.packed()
    names = ["p%d" % i for i in range(pk0["numPhys"])]
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk0, names)
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    scp = []
    for u, (X, q) in enumerate(zip(s.feats, s.seqs)):
        fn = os.path.join(d, "u%d.mfc" % u)
        synth.write_htk_param(fn, X, kind=9); scp.append(fn)
        open(os.path.join(d, "u%d.lab" % u), "w").write("\n".join(names[int(h)] for h in q) + "\n")
    open(os.path.join(d, "config"), "w").write("")
    flagsel = str(rng.choice(["tmvw", "mv", "tw", "m", "tmv"]))
    uFlags = sum({"m": capi.UPMEANS, "v": capi.UPVARS, "t": capi.UPTRANS, "w": capi.UPMIXES}[ch] for ch in flagsel)
    prune = float(rng.choice([0.0, 60.0, 150.0]))
    args = ["-u", flagsel] + (["-t", "%g" % prune] if prune > 0 else [])
    r = subprocess.run([os.path.join(REF, "HERest"), "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-M", os.path.join(d, "out"), "-L", d,
                        "-p", "1"] + args + [os.path.join(d, "hmmlist")] + scp, capture_output=True, text=True)
    ref_acc = os.path.join(d, "out", "HER1.acc")
    if r.returncode != 0 or not os.path.exists(ref_acc):
        print("ACC it %d: HERest failed: %s" % (it, r.stdout[-300:])); return False
    pk = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist")).packed()
    om = pyoracle.Model(pk); oacc = pyoracle.Accs(om)
    cfg = pyoracle.fb_cfg(uFlags=uFlags) if prune == 0 else pyoracle.fb_cfg(pruneInit=prune, pruneLim=prune, uFlags=uFlags)
    totalPr, totalT = 0.0, 0
    for X, q in zip(s.feats, s.seqs):
        rc, pr, _ = pyoracle.fb_utt(om, cfg, X, np.asarray(q, np.int32), oacc)
        if rc == 1:                                            # TRUE: the utterance was accumulated
            totalPr += pr; totalT += X.shape[0]
    lay = capi.accs_layout(pk)
    v = np.zeros(lay.total, np.float64)
    v[lay.totalPr], v[lay.totalT] = totalPr, totalT
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs"):
        x = np.asarray(getattr(oacc, k), np.float64).reshape(-1)
        off = getattr(lay, k); v[off:off + x.size] = x
    ours = os.path.join(d, "ours.acc")
    capi.accs_dump_file(pk, v, names, ours, uFlags)
    a, b = open(ours, "rb").read(), open(ref_acc, "rb").read()
    ok = a == b
    if not ok:
        w = np.zeros_like(v)
        try:
            capi.accs_load_file(pk, w, names, ref_acc, uFlags)
            bad = [(k, float(np.max(np.abs(w[getattr(lay, k):getattr(lay, k) + np.asarray(getattr(oacc, k)).size] - np.asarray(getattr(oacc, k), np.float64).reshape(-1)))))
                   for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs")]
        except Exception as e:  # noqa: BLE001
            bad = str(e)
        print("ACC it %d: files differ (%d vs %d bytes, -u %s -t %g): %s" % (it, len(a), len(b), flagsel, prune, bad))
        if os.environ.get("FUZZ_KEEP"):
            import shutil
            shutil.copytree(d, os.path.join(os.environ["FUZZ_KEEP"], "acc_%d" % it), dirs_exist_ok=True)
    return ok


def mfcc_case(rng):
    """A random front-end configuration: (HCopy config text, keyword arguments of mfcc_cfg, TARGETKIND)."""
    base = str(rng.choice(["0", "E"]))
    quals = [base] + (["D"] + (["A"] if rng.random() < 0.6 else []) if rng.random() < 0.7 else []) + (["Z"] if rng.random() < 0.3 else [])
    kind = "MFCC_" + "_".join(quals)
    rate = int(rng.choice([8000, 16000]))
    kw = dict(sampPeriod=1.0e7 / rate, winDur=float(rng.choice([200000.0, 250000.0, 320000.0])), frPeriod=float(rng.choice([100000.0, 80000.0, 125000.0])),
              numChans=int(rng.choice([20, 24, 26, 40])), numCeps=int(rng.choice([8, 12, 13])), cepLifter=int(rng.choice([0, 22, 30])),
              preEmph=float(rng.choice([0.0, 0.95, 0.97])), useHam=bool(rng.random() < 0.8), usePower=bool(rng.random() < 0.3),
              zMeanSource=bool(rng.random() < 0.3), rawEnergy=bool(rng.random() < 0.6), eNormalise=bool(rng.random() < 0.6),
              delWin=int(rng.choice([1, 2, 3])), accWin=int(rng.choice([1, 2])))
    if rng.random() < 0.3:
        kw["loFreq"], kw["hiFreq"] = float(rng.choice([64.0, 125.0, 300.0])), float(rng.choice([3400.0, 3800.0]))
    if rng.random() < 0.2:
        kw["silFloor"], kw["eScale"] = 40.0, 0.2
    T = lambda b: "T" if b else "F"
    cfg = ("SOURCEFORMAT = WAV\nTARGETKIND = %s\nWINDOWSIZE = %.1f\nTARGETRATE = %.1f\nNUMCHANS = %d\nNUMCEPS = %d\nCEPLIFTER = %d\nPREEMCOEF = %g\n"
           "USEHAMMING = %s\nUSEPOWER = %s\nZMEANSOURCE = %s\nRAWENERGY = %s\nENORMALISE = %s\nDELTAWINDOW = %d\nACCWINDOW = %d\n"
           % (kind, kw["winDur"], kw["frPeriod"], kw["numChans"], kw["numCeps"], kw["cepLifter"], kw["preEmph"], T(kw["useHam"]), T(kw["usePower"]),
              T(kw["zMeanSource"]), T(kw["rawEnergy"]), T(kw["eNormalise"]), kw["delWin"], kw["accWin"]))
    if "loFreq" in kw:
        cfg += "LOFREQ = %g\nHIFREQ = %g\n" % (kw["loFreq"], kw["hiFreq"])
    if "silFloor" in kw:
        cfg += "SILFLOOR = %g\nESCALE = %g\n" % (kw["silFloor"], kw["eScale"])
    return kind, kw, cfg, rate


def fuzz_mfcc(rng, it, tmp):
    """Waveform -> MFCC: the file the reference's HCopy codes from a WAV vs oracle.mfcc, every float."""
    import wave
    d = os.path.join(tmp, "m%d" % it); os.makedirs(d, exist_ok=True)
    kind, kw, cfg, rate = mfcc_case(rng)
    n = int(rng.integers(rate // 10, rate))
    t = np.arange(n) / rate
    x = (float(rng.uniform(500, 6000)) * np.sin(2 * np.pi * float(rng.uniform(100, 1500)) * t) * np.sin(2 * np.pi * 3 * t) + rng.normal(0, float(rng.uniform(50, 1500)), n))
    x = x.clip(-32768, 32767).astype("<i2")
    with wave.open(os.path.join(d, "x.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(x.tobytes())
    open(os.path.join(d, "cfg"), "w").write(cfg)
    r = subprocess.run([os.path.join(REF, "HCopy"), "-C", os.path.join(d, "cfg"), os.path.join(d, "x.wav"), os.path.join(d, "x.mfc")], capture_output=True, text=True)
    if r.returncode != 0:
        print("MFCC it %d: HCopy failed on %s: %s" % (it, cfg.replace("\n", " "), (r.stdout + r.stderr)[-300:])); return False
    ref, _, _ = capi.parm_read(os.path.join(d, "x.mfc"))
    got = pyoracle.mfcc(x, pyoracle.mfcc_cfg(kind, **kw))
    if got.shape != ref.shape or not np.array_equal(got, ref):
        bad = "shape %s vs %s" % (got.shape, ref.shape) if got.shape != ref.shape else "%d of %d values differ, max %.3g" % ((got != ref).sum(), got.size, np.abs(got - ref).max())
        print("MFCC it %d: %s: %s" % (it, cfg.replace("\n", " "), bad))
        return False
    return True


def quals_case(rng):
    """Random qualifier step on a 13-column MFCC_E table: (TARGETKIND, oracle keyword arguments, HCopy config text)."""
    q = ["D"] + (["A"] + (["T"] if rng.random() < 0.5 else []) if rng.random() < 0.7 else [])
    z = rng.random() < 0.4
    kind = "MFCC_E_" + "_".join(q) + ("_Z" if z else "")
    kw = dict(hasD=True, hasA="A" in q, hasT="T" in q, delWin=int(rng.integers(1, 5)), accWin=int(rng.integers(1, 4)), thirdWin=int(rng.integers(1, 4)),
              nZeroMean=12 if z else 0, v1Compat=bool(rng.random() < 0.25), simpleDiffs=bool(rng.random() < 0.25))
    cfg = "TARGETKIND = %s\nDELTAWINDOW = %d\nACCWINDOW = %d\nTHIRDWINDOW = %d\nV1COMPAT = %s\nSIMPLEDIFFS = %s\n" % (
        kind, kw["delWin"], kw["accWin"], kw["thirdWin"], "T" if kw["v1Compat"] else "F", "T" if kw["simpleDiffs"] else "F")
    return kind, kw, cfg


def fuzz_quals(rng, it, tmp):
    """The qualifier step at load time (_D _A _T _Z, windows, V1COMPAT, SIMPLEDIFFS): the file HCopy writes from an MFCC_E file vs
    oracle.parm_qualify, every float; short tables (fewer rows than the windows need) included."""
    d = os.path.join(tmp, "q%d" % it); os.makedirs(d, exist_ok=True)
    kind, kw, cfg = quals_case(rng)
    T = int(rng.choice([1, 2, 3, 4, 5, 7, 9, int(rng.integers(10, 120))]))
    X = rng.normal(0, 3, size=(T, 13)).astype(np.float32)
    synth.write_htk_param(os.path.join(d, "x.mfc"), X, kind=6 | 0o100)          # MFCC_E
    open(os.path.join(d, "cfg"), "w").write(cfg)
    r = subprocess.run([os.path.join(REF, "HCopy"), "-C", os.path.join(d, "cfg"), os.path.join(d, "x.mfc"), os.path.join(d, "y.mfc")], capture_output=True, text=True)
    if r.returncode != 0:
        print("QUALS it %d: HCopy failed (%s, T=%d): %s" % (it, cfg.replace("\n", " "), T, (r.stdout + r.stderr)[-200:])); return False
    ref, _, _ = capi.parm_read(os.path.join(d, "y.mfc"))
    got = pyoracle.parm_qualify(X, **kw)
    if got.shape != ref.shape or not np.array_equal(got, ref):
        print("QUALS it %d: %s T=%d: %s" % (it, cfg.replace("\n", " "), T, "shape %s vs %s" % (got.shape, ref.shape) if got.shape != ref.shape else
                                              "%d of %d values differ" % ((got != ref).sum(), got.size)))
        return False
    return True


def fuzz_mmf(rng, it, tmp):
    """Model files: what htkamd_mmf_write (text and binary) produces is read by the reference's HHEd and saved again -- the text
    must come back byte for byte, the binary must load to the same numbers -- and what HHEd saves in binary is read back by
    htkamd_mmf_read to the same arrays.  Random sets: tied states, 1-6 mixtures, mixed topologies with a tee model, defunct
    mixture components (weight 0: left out of the file, HModel.c:3094)."""
    import ctypes as C
    d = os.path.join(tmp, "h%d" % it); os.makedirs(d, exist_ok=True)
    if rng.random() < 0.4:
        pk, names, _, _ = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=int(rng.choice([5, 13])), NU=1)
    else:
        NPm = int(rng.integers(3, 15))                               # every state macro in use: HHEd purges unused macros when it saves
        s = synth.generate(int(rng.integers(3, 3 * NPm + 1)), int(rng.integers(1, 7)), NPm, 1, 10, int(rng.integers(1, 10**6)), D=int(rng.choice([3, 13, 26])))
        pk = s.packed(); names = ["p%d" % i for i in range(pk["numPhys"])]
        if len(set(int(x) for x in pk["hmmState"])) < pk["numStates"]:
            return True                                               # (the generator left a state unused: not a case for this check)
    synth.write_mmf_packed(os.path.join(d, "src.mmf"), pk, names)
    lst = os.path.join(d, "hmmlist"); open(lst, "w").write("\n".join(names) + "\n")
    m = capi.Mmf(files=[os.path.join(d, "src.mmf")], hmm_list=lst)
    q = m.packed()
    w = q["compWeight"].copy()
    if rng.random() < 0.4:                                          # a defunct component in a state with several
        cand = [i for i in range(q["numStates"]) if q["stateCompOff"][i + 1] - q["stateCompOff"][i] >= 2]
        if cand:
            st = int(rng.choice(cand)); c0, c1 = int(q["stateCompOff"][st]), int(q["stateCompOff"][st + 1])
            w[c0 + int(rng.integers(0, c1 - c0))] = 0.0
            w[c0:c1] /= w[c0:c1].sum()
    def gconsts(var):                                                # HHEd fixes every gConst before it saves (FixAllGConsts)
        g_ = np.empty(len(var), np.float32)
        for i in range(len(var)):
            x = np.zeros(1, np.float32)
            capi.lib().htkamd_host_fix_diag_gconst(C.c_int(var.shape[1]), np.ascontiguousarray(var[i]).ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
            g_[i] = x[0]
        return g_
    g = gconsts(q["var"])
    par = dict(mean=q["mean"], var=q["var"], gconst=g, compWeight=w, transP=q["transP"])
    m.write(par, one_file=os.path.join(d, "ours.mmf"))
    # the binary file from what the TEXT file holds (7 digits), which is all HHEd gets to see
    m2 = capi.Mmf(files=[os.path.join(d, "ours.mmf")], hmm_list=lst)
    q2 = m2.packed()
    m2.write(dict(mean=q2["mean"], var=q2["var"], gconst=gconsts(q2["var"]), compWeight=q2["compWeight"], transP=q2["transP"]), one_file=os.path.join(d, "ours.bin"), binary=True)
    open(os.path.join(d, "e.hed"), "w").close()
    ok = True
    for src, out, extra in (("ours.mmf", "ref_from_text.mmf", []), ("ours.bin", "ref_from_bin.mmf", []), ("ours.mmf", "ref.bin", ["-B"])):
        r = subprocess.run([os.path.join(REF, "HHEd")] + extra + ["-H", os.path.join(d, src), "-w", os.path.join(d, out), os.path.join(d, "e.hed"), lst], capture_output=True, text=True)
        if r.returncode != 0:
            print("MMF it %d: HHEd failed on %s: %s" % (it, src, (r.stdout + r.stderr)[-300:])); return False
    ours = open(os.path.join(d, "ours.mmf"), "rb").read()
    for out in ("ref_from_text.mmf", "ref_from_bin.mmf"):
        if open(os.path.join(d, out), "rb").read() != ours:
            ok = False; print("MMF it %d: %s differs from the file we wrote" % (it, out))
    if open(os.path.join(d, "ref.bin"), "rb").read() != open(os.path.join(d, "ours.bin"), "rb").read():
        ok = False; print("MMF it %d: binary file differs from HHEd -B" % it)
    a, b = capi.Mmf(files=[os.path.join(d, "ref.bin")], hmm_list=lst).packed(), capi.Mmf(files=[os.path.join(d, "ours.mmf")], hmm_list=lst).packed()
    for k in ("mean", "var", "compWeight", "transP", "compGauss", "stateCompOff", "hmmState", "hmmTrans"):
        if k in ("mean", "var", "compWeight", "transP"):
            same = np.allclose(a[k], b[k], rtol=2e-7, atol=1e-30) if k != "transP" else np.allclose(np.exp(np.maximum(a[k], -80)), np.exp(np.maximum(b[k], -80)), rtol=1e-6, atol=1e-30)
        else:
            same = np.array_equal(a[k], b[k])
        if not same:
            ok = False; print("MMF it %d: %s differs between HHEd's binary and our text" % (it, k))
    if not ok and os.environ.get("FUZZ_KEEP"):
        import shutil
        shutil.copytree(d, os.path.join(os.environ["FUZZ_KEEP"], "mmf_%d" % it), dirs_exist_ok=True)
    return ok


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
    if not os.path.exists(os.path.join(REF, "HVite")):
        sys.exit("needs oracle/_ref (make -C oracle)")
    kinds = (("decode", fuzz_decode), ("fb", fuzz_fb), ("align", fuzz_align), ("update", fuzz_update), ("acc", fuzz_acc), ("mfcc", fuzz_mfcc),
             ("quals", fuzz_quals), ("mmf", fuzz_mmf))
    if os.environ.get("FUZZ_KINDS"):                      # e.g. FUZZ_KINDS=acc,mmf
        kinds = tuple(k for k in kinds if k[0] in os.environ["FUZZ_KINDS"].split(","))
    res = {k[0]: [0, 0] for k in kinds}
    with tempfile.TemporaryDirectory() as tmp:
        for it in range(n):
            for name, fn in kinds:
                ok = fn(rng, it, tmp)
                res[name][0] += 1; res[name][1] += int(ok)
    print("passed/total:", {k: "%d/%d" % (v[1], v[0]) for k, v in res.items()})
    sys.exit(0 if all(v[0] == v[1] for v in res.values()) else 1)


if __name__ == "__main__":
    main()
