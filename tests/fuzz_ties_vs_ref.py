"""Tie stress for the network decoder's oracle (CPU, this container): dictionaries full of homophones (the same model sequence under
several words, with and without a closing tee model, different pronunciation probabilities) and lattices whose arcs repeat a few
log-probability values, so that different word sequences reach a node with EXACTLY the same likelihood -- sums of the same float
terms in another order are exact in double.  What survives such a tie in the reference is decided by the order of its instance
list (SetEntryState HRec.c:1303 keeps the first arrival); the oracle must make the same choice.
   python tests/fuzz_ties_vs_ref.py [iterations] [seed] [--device]
With --device (a GPU box; oracle/_ref travels there) the HIP decoder is held against HVite on the same cases as well, in its default mode
(batch kernel; utterances in which it met an exact tie once more in the list's order, decode_ord.hip) and with every utterance through the
list kernel."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from htk_amd import capi, synth  # noqa: E402
import pyoracle  # noqa: E402
from decode_util import format_words  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def tie_case(rng, d):
    """Writes MMF / hmmlist / dict / net.slf into d; returns (features, decoder parameters, HVite options)."""
    pkt, names, seqs, feats = synth.make_topo_set(seed=int(rng.integers(1, 10**6)), D=13, NU=int(rng.integers(2, 4)))
    synth.write_mmf_packed(os.path.join(d, "MMF"), pkt, names)
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    tee = names.index("sp")
    real = [k for k in range(len(names)) if k != tee]
    nBase = int(rng.integers(1, 4))                                  # distinct model sequences shared by all the words
    base = [[names[int(rng.choice(real))] for _ in range(int(rng.integers(1, 3)))] for _ in range(nBase)]
    V = int(rng.integers(3, 7))
    probs = ["", "0.50 ", "0.50 ", "0.25 ", "%.2f " % rng.uniform(0.1, 1.0)]
    with open(os.path.join(d, "dict"), "w") as f:
        for w in range(V):
            for _ in range(int(rng.integers(1, 3))):
                ph = list(base[int(rng.integers(0, nBase))]) + (["sp"] if rng.random() < 0.4 else [])
                f.write("W%d %s%s\n" % (w, probs[int(rng.integers(0, len(probs)))], " ".join(ph)))
    vals = [float(np.log(x)) for x in (0.1, 0.2, 0.2, 0.4)]       # few distinct arc values
    arcs = []
    for w in range(V):
        arcs.append((0, 1 + w, float(np.log(1.0 / V))))
        for k in rng.choice(V, size=min(V, int(rng.integers(2, 5))), replace=False):
            arcs.append((1 + w, 1 + int(k), vals[int(rng.integers(0, len(vals)))]))
        arcs.append((1 + w, V + 1, float(np.log(0.1))))
    with open(os.path.join(d, "net.slf"), "w") as f:
        f.write("VERSION=1.0\nN=%d L=%d\nI=0 W=!NULL\n" % (V + 2, len(arcs)))
        for w in range(V):
            f.write("I=%d W=W%d\n" % (1 + w, w))
        f.write("I=%d W=!NULL\n" % (V + 1))
        for j, (a_, b_, l) in enumerate(arcs):
            f.write("J=%d S=%d E=%d l=%.4f\n" % (j, a_, b_, l))
    p = dict(genBeam=float(rng.choice([1.0e10, round(rng.uniform(30, 200), 2)])), wordBeam=float(rng.choice([1.0e10, round(rng.uniform(10, 100), 2)])),
             lmScale=float(rng.choice([1.0, 2.0])), wordPen=float(rng.choice([0.0, -4.0])), prScale=float(rng.choice([1.0, 2.0, round(rng.uniform(0.5, 3), 2)])))
    opts = []
    if p["genBeam"] < 1e9: opts += ["-t", "%.2f" % p["genBeam"]]
    if p["wordBeam"] < 1e9: opts += ["-v", "%.2f" % p["wordBeam"]]
    opts += ["-s", "%.2f" % p["lmScale"], "-p", "%.2f" % p["wordPen"], "-r", "%.2f" % p["prScale"]]
    return feats, p, opts


def run_hvite(d, feats, opts):
    scp = []
    for u, X in enumerate(feats):
        fn = os.path.join(d, "u%d.mfc" % u)
        synth.write_htk_param(fn, X, kind=9)
        scp.append(fn)
    open(os.path.join(d, "scp"), "w").write("\n".join(scp) + "\n")
    open(os.path.join(d, "config"), "w").write("")
    mlf = os.path.join(d, "out.mlf")
    if os.path.exists(mlf):
        os.remove(mlf)
    subprocess.run([os.path.join(REF, "HVite"), "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-S", os.path.join(d, "scp"), "-i", mlf,
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
    return ref


def main():
    device = "--device" in sys.argv
    argv = [a for a in sys.argv if a != "--device"]
    n = int(argv[1]) if len(argv) > 1 else 50
    rng = np.random.default_rng(int(argv[2]) if len(argv) > 2 else 1)
    tmp = tempfile.mkdtemp()
    bad = 0; tot = 0
    dbad = 0; dtied = 0
    for it in range(n):
        d = os.path.join(tmp, "t%d" % it); os.makedirs(d)
        feats, p, opts = tie_case(rng, d)
        ref = run_hvite(d, feats, opts)
        mmf = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
        net = capi.Net(os.path.join(d, "net.slf"), os.path.join(d, "dict"), mmf)
        om = pyoracle.Model(mmf.packed())
        if device:
            dec = capi.Decoder(capi.Model(mmf.packed()), net, lmScale=p["lmScale"])
            for mode in (capi.ORDER_AUTO, capi.ORDER_EXACT):
                dec.set_order(mode)
                res = dec.run(feats, **p)
                if mode == capi.ORDER_AUTO:
                    dtied += dec.last_tied()
                for u, (w, _) in enumerate(res):
                    got = None if w is None else format_words(w, net.out_syms)
                    want = ref.get("u%d" % u)
                    if got != want and not (got is None and want is None):
                        dbad += 1
                        print("TIE-DEVICE it %d u%d mode %d %s\n  device %s\n  HVite  %s" % (it, u, mode, opts, got, want))
        for u, X in enumerate(feats):
            ow, ot = pyoracle.decode(om, X, net.arrays(), **p)
            got = None if ow is None else format_words(ow, net.out_syms)
            want = ref.get("u%d" % u)
            tot += 1
            if got != want and not (got is None and want is None):
                bad += 1
                print("TIE it %d u%d %s\n  oracle %s\n  HVite  %s" % (it, u, opts, got, want))
                if os.environ.get("FUZZ_KEEP"):
                    import json, shutil
                    json.dump(dict(params=p, opts=opts), open(os.path.join(d, "params.json"), "w"))
                    np.savez(os.path.join(d, "feats.npz"), **{"u%d" % k: x for k, x in enumerate(feats)})
                    shutil.copytree(d, os.path.join(os.environ["FUZZ_KEEP"], "tie_%d" % it), dirs_exist_ok=True)
    print("utterances equal to HVite: %d/%d" % (tot - bad, tot))
    if device:
        print("device: %d differences from HVite over %d utterances x 2 modes; %d utterances took the list kernel in the default mode" % (dbad, tot, dtied))
    sys.exit(1 if (bad or dbad) else 0)


if __name__ == "__main__":
    main()
