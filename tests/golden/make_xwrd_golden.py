"""Known answers for CROSS-WORD context expansion (ExpandWordNet HNet.c:3438 with xc > 0: CreateX1Model :2773, CreateXEModels :3141,
ProcessCrossWordLinks :2559, SetNullContexts :2514): the reference's HVite with  FORCECXTEXP = T, ALLOWXWRDEXP = T  on a synthetic
triphone system.
    python tests/golden/make_xwrd_golden.py          (needs oracle/_ref)   -> tests/golden/decode/xwrd/
Model set: 12 physical models; logical names = every triphone l-p+r, left / right biphone and monophone of p in {a,b,c,d} with contexts
from {a,b,c,d,sil}, tied round-robin to 10 of them; `sil` (a context, context independent) and `sp` (a tee-less short pause that is
nobody's context: context free).  Dictionary: multi-phone words ending in sp, a one-phone word (the (lc, rc) cross-bar), a word without
trailing sp, SIL.  Two networks: a word loop through one null node (null words pass the contexts on) under `loop.slf`, and a
hand-written lattice with direct word-to-word arcs and LM scores under `net.slf`."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from htk_amd import synth  # noqa: E402
from make_decode_golden import run_hvite, sample, REF, OUT  # noqa: E402

CFG = "FORCECXTEXP = T\nALLOWXWRDEXP = T\n"


def tri(l, p, r):
    return ("%s-" % l if l else "") + p + ("+%s" % r if r else "")


def main():
    rng = np.random.default_rng(77)
    d = os.path.join(OUT, "xwrd"); os.makedirs(d, exist_ok=True)
    s = synth.generate(30, 3, 12, 3, 80, 11, D=13)
    pk = s.packed()
    synth.write_mmf(os.path.join(d, "MMF"), s, kind="USER")
    names = ["p%d" % i for i in range(12)]
    phones, ctx = ["a", "b", "c", "d"], ["a", "b", "c", "d", "sil"]
    logical, k = [], 0
    for p_ in phones:
        for l_ in [None] + ctx:
            for r_ in [None] + ctx:
                logical.append((tri(l_, p_, r_), "p%d" % (k % 10))); k += 1
    logical += [("sil", "p10"), ("sp", "p11")]
    with open(os.path.join(d, "hmmlist"), "w") as f:
        for n in names:
            f.write(n + "\n")
        for lg, ph in logical:
            f.write("%s %s\n" % (lg, ph))
    words = {"SIL": ["sil"], "W1": ["a", "b", "sp"], "W2": ["c", "d", "a", "sp"], "W3": ["b", "sp"], "W4": ["d", "c", "b", "a"], "W5": ["c"]}
    open(os.path.join(d, "dict"), "w").write("".join("%s %s\n" % (w, " ".join(p)) for w, p in words.items()))
    # word loop through ONE null node (the reference's own expansion fails on HBuild's null-to-null links: "FindWordNode: Node !NULL not
    # created", its comment at HNet.c:3502 calls that part flawed): SIL -> words -> !NULL -> words ..., words -> SIL
    wl0 = ["W1", "W2", "W3", "W4", "W5"]
    with open(os.path.join(d, "loop.slf"), "w") as f:
        la = [(0, 1 + j, -1.61) for j in range(5)] + [(1 + j, 6, 0.0) for j in range(5)] + [(6, 1 + j, -1.61) for j in range(5)] + [(1 + j, 7, -2.0) for j in range(5)]
        f.write("VERSION=1.0\nN=8 L=%d\n" % len(la))
        for i, w in enumerate(["SIL"] + wl0 + ["!NULL", "SIL"]):
            f.write("I=%d W=%s\n" % (i, w))
        for j, (a, b, l) in enumerate(la):
            f.write("J=%d S=%d E=%d l=%.2f\n" % (j, a, b, l))
    # a lattice with direct arcs: SIL -> {W1..W5} -> {W1..W5} -> {W1..W5} -> SIL, LM scores on the arcs
    wl = ["W1", "W2", "W3", "W4", "W5"]
    nodes = ["SIL"] + wl * 3 + ["SIL"]
    arcs = []
    for j in range(5):
        arcs.append((0, 1 + j, -1.0 - 0.3 * j))
    for layer in range(2):
        for a in range(5):
            for b in range(5):
                arcs.append((1 + 5 * layer + a, 6 + 5 * layer + b, -0.5 - 0.25 * ((a + 2 * b + layer) % 5)))
    for a in range(5):
        arcs.append((11 + a, 16, -0.7))
    with open(os.path.join(d, "net.slf"), "w") as f:
        f.write("VERSION=1.0\nN=%d L=%d\n" % (len(nodes), len(arcs)))
        for i, w in enumerate(nodes):
            f.write("I=%d W=%s\n" % (i, w))
        for j, (a, b, l) in enumerate(arcs):
            f.write("J=%d S=%d E=%d l=%.2f\n" % (j, a, b, l))
    lmap = dict(logical)
    cxt_of = lambda ph: ph if ph in ctx else None

    def models_of(seq):
        """cross-word model names of a word sequence"""
        flat = [(w, p) for w in seq for p in words[w]]
        out = []
        for i, (w, p) in enumerate(flat):
            if p in ("sil", "sp"):
                out.append(p); continue
            l = next((q for _, q in reversed(flat[:i]) if cxt_of(q)), None)
            r = next((q for _, q in flat[i + 1:] if cxt_of(q)), None)
            out.append(tri(l, p, r))
        return out

    feats3, featsL = [], []
    for u in range(4):
        seq = ["SIL"] + [wl[j] for j in rng.integers(0, 5, size=3)] + ["SIL"]
        feats3.append(sample(pk, [int(lmap[m][1:]) for m in models_of(seq)], rng, frames_per_state=3))
    for u in range(3):
        seq = ["SIL"] + [wl[j] for j in rng.integers(0, 5, size=int(rng.integers(2, 6)))] + ["SIL"]
        featsL.append(sample(pk, [int(lmap[m][1:]) for m in models_of(seq)], rng, frames_per_state=3))
    opts3 = ["-t 250.0", "-t 250.0 -p -10.0 -s 4.0", "-t 60.0 -v 40.0", "-m -t 250.0"]
    e3 = run_hvite(d, feats3, opts3, 9, CFG)
    os.rename(os.path.join(d, "feats.npz"), os.path.join(d, "feats_net.npz")); os.rename(os.path.join(d, "expected.json"), os.path.join(d, "expected_net.json"))
    os.rename(os.path.join(d, "net.slf"), os.path.join(d, "net3.slf")); os.rename(os.path.join(d, "loop.slf"), os.path.join(d, "net.slf"))
    eL = run_hvite(d, featsL, ["-t 250.0", "-t 250.0 -p -15.0", "-m -t 250.0"], 9, CFG)
    os.rename(os.path.join(d, "feats.npz"), os.path.join(d, "feats_loop.npz")); os.rename(os.path.join(d, "expected.json"), os.path.join(d, "expected_loop.json"))
    os.rename(os.path.join(d, "net.slf"), os.path.join(d, "loop.slf")); os.rename(os.path.join(d, "net3.slf"), os.path.join(d, "net.slf"))
    open(os.path.join(d, "config"), "w").write(CFG)
    for nm, e in (("net", e3), ("loop", eL)):
        print(nm, {k: sum(len(v) for v in per.values()) for k, per in e.items()})
        print(json.dumps(e[list(e)[0]], indent=0)[:600])


if __name__ == "__main__":
    main()
