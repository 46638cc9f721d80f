"""Known answers for network decoding (HVite -w): the reference's HVite run on small synthetic systems.
    python tests/golden/make_decode_golden.py         (needs oracle/_ref, i.e. `make -C oracle`)

Cases (all files land in tests/golden/decode/<case>/, a few hundred kB in total):
  loop   12 single-phone words, word loop built by the reference's HBuild, 3 utterances               HVite -t 250
  bigram multi-phone words, two pronunciations with probabilities, an output symbol, a word without output symbol,
         a hand-written back-off bigram lattice (per-word successor arcs + back-off null node)        3 option sets
  tee    mixed-topology models incl. a tee model `sp` at the end of some pronunciations, word loop    HVite -t 250 / -t 40
  wint   a monophone dictionary over a model set of word-internal triphones/biphones (logical names tied to 12 physical
         models), so that ExpandWordNet's context expansion is exercised                              2 option sets
  ties   a case the randomised GPU sweep recorded (python tests/fuzz_parity.py 1841 20261002 with HTKAMD_FUZZ_KEEP_DECODE=1840: the
         generator's files are kept as they were written): homophones W0 = "a sp" (tee model skipped) and W2 = "a" with different
         pronunciation probabilities make "... W2 W0 W2" and "... W0 W2 W2" EXACTLY equally likely (the same float terms summed in
         another order); the reference keeps the token that arrives first in its instance-list order      -t 83.53 -v 33.40 -r 1.7962
Each case: MMF (text), hmmlist, dict, net.slf, feats.npz (the utterances' feature matrices), expected.json =
{option string: {utterance: [label lines of the .rec file]}} exactly as HVite wrote them (no entry = "No tokens survived").
NOTE: HBuild numbers the nodes of a word loop in an order that varies from run to run; after re-running this script keep the
committed net.slf of `loop` and `tee` (git checkout) -- the expectations do not depend on the numbering."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from htk_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "decode")


def run_hvite(case_dir, feats, opts_list, kind_code, cfg_text):
    os.makedirs(os.path.join(case_dir, "tmp"), exist_ok=True)
    scp = []
    for u, X in enumerate(feats):
        fn = os.path.join(case_dir, "tmp", "u%d.mfc" % u)
        synth.write_htk_param(fn, X, kind=kind_code)
        scp.append(fn)
    with open(os.path.join(case_dir, "tmp", "scp"), "w") as f:
        f.write("\n".join(scp) + "\n")
    with open(os.path.join(case_dir, "tmp", "config"), "w") as f:
        f.write(cfg_text)
    expected = {}
    for opts in opts_list:
        mlf = os.path.join(case_dir, "tmp", "rec.mlf")
        cmd = [os.path.join(REF, "HVite"), "-C", os.path.join(case_dir, "tmp", "config"), "-H", os.path.join(case_dir, "MMF"),
               "-S", os.path.join(case_dir, "tmp", "scp"), "-i", mlf, "-w", os.path.join(case_dir, "net.slf")] + opts.split() + \
              [os.path.join(case_dir, "dict"), os.path.join(case_dir, "hmmlist")]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        per = {}
        cur = None
        for line in open(mlf).read().splitlines()[1:]:
            if line.startswith('"'):
                cur = os.path.basename(line.strip('"')).replace(".rec", ""); per[cur] = []
            elif line == ".":
                cur = None
            elif cur is not None:
                per[cur].append(line)
        expected[opts] = per
    np.savez_compressed(os.path.join(case_dir, "feats.npz"), **{"u%d" % u: X for u, X in enumerate(feats)})
    with open(os.path.join(case_dir, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    for fn in os.listdir(os.path.join(case_dir, "tmp")):
        os.remove(os.path.join(case_dir, "tmp", fn))
    os.rmdir(os.path.join(case_dir, "tmp"))
    return expected


def sample(pk, phones, rng, frames_per_state=4):
    fr = []
    for ph in phones:
        for j in range(pk["hmmStateOff"][ph], pk["hmmStateOff"][ph + 1]):
            st = pk["hmmState"][j]
            c = rng.integers(pk["stateCompOff"][st], pk["stateCompOff"][st + 1])
            g = pk["compGauss"][c]
            fr.append(pk["mean"][g] + np.sqrt(pk["var"][g]) * rng.normal(size=(frames_per_state, pk["vecSize"])))
    return np.concatenate(fr).astype(np.float32)


def main():
    if not os.path.exists(os.path.join(REF, "HVite")):
        sys.exit("needs oracle/_ref (make -C oracle)")
    rng = np.random.default_rng(2024)
    # ---------------------------------------------------------------- loop
    d = os.path.join(OUT, "loop"); os.makedirs(d, exist_ok=True)
    s = synth.generate(30, 3, 12, 3, 80, 11, D=13)
    synth.write_mmf(os.path.join(d, "MMF"), s, kind="USER")
    names = ["p%d" % i for i in range(12)]
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    open(os.path.join(d, "dict"), "w").write("".join("%s %s\n" % (n, n) for n in sorted(names)))
    open(os.path.join(d, "wlist"), "w").write("\n".join(sorted(names)) + "\n")
    subprocess.check_call([os.path.join(REF, "HBuild"), "wlist", "net.slf"], cwd=d)
    os.remove(os.path.join(d, "wlist"))
    run_hvite(d, s.feats, ["-t 250.0", "-t 25.0", "-m -t 250.0"], 9, "")
    # ---------------------------------------------------------------- bigram
    d = os.path.join(OUT, "bigram"); os.makedirs(d, exist_ok=True)
    synth.write_mmf(os.path.join(d, "MMF"), s, kind="USER")
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    open(os.path.join(d, "dict"), "w").write("AB 0.7 p0 p1\nAB 0.3 p0 p2\nCD p3 p4 p5\nE [eee] p6\nF [] p7\nG p8 p9\nH p10\nI p11\n")
    words = ["AB", "CD", "E", "F", "G", "H", "I"]
    V = len(words)
    # nodes: 0 start !NULL, 1..V words, V+1 back-off !NULL, V+2 end !NULL
    arcs = []
    for w in range(V):
        arcs.append((0, 1 + w, float(np.log(1.0 / V))))                         # unigram start
        succ = rng.choice(V, size=3, replace=False)
        pr = rng.dirichlet(np.ones(3)) * 0.8
        for k, p in zip(succ, pr):
            arcs.append((1 + w, 1 + int(k), float(np.log(p))))                 # bigram arcs
        arcs.append((1 + w, V + 1, float(np.log(0.15))))                        # to back-off node
        arcs.append((1 + w, V + 2, float(np.log(0.05))))                        # sentence end
    for w in range(V):
        arcs.append((V + 1, 1 + w, float(np.log(1.0 / V))))                     # back-off unigrams
    with open(os.path.join(d, "net.slf"), "w") as f:
        f.write("VERSION=1.0\nN=%d L=%d\n" % (V + 3, len(arcs)))
        f.write("I=0 W=!NULL\n")
        for w in range(V):
            f.write("I=%d W=%s\n" % (1 + w, words[w]))
        f.write("I=%d W=!NULL\nI=%d W=!NULL\n" % (V + 1, V + 2))
        for j, (a, b, l) in enumerate(arcs):
            f.write("J=%d S=%d E=%d l=%.4f\n" % (j, a, b, l))
    prons = {"AB": [[0, 1], [0, 2]], "CD": [[3, 4, 5]], "E": [[6]], "F": [[7]], "G": [[8, 9]], "H": [[10]], "I": [[11]]}
    pk = s.packed()
    feats = []
    for u in range(4):
        seq = [words[k] for k in rng.integers(0, V, size=5)]
        ph = []
        for w in seq:
            alt = prons[w]
            ph += alt[int(rng.integers(0, len(alt)))]
        feats.append(sample(pk, ph, rng))
    run_hvite(d, feats, ["-t 250.0", "-t 250.0 -s 5.0 -p -10.0", "-t 60.0 -v 30.0 -s 2.0 -p 3.0 -r 2.0", "-m -t 250.0 -s 5.0 -p -10.0", "-t 250.0 -r 2.25", "-v 25.0 -r 1.37", "-u 6 -t 250.0", "-u 3 -s 2.0"], 9, "")
    # ---------------------------------------------------------------- tee
    d = os.path.join(OUT, "tee"); os.makedirs(d, exist_ok=True)
    pk2, tnames, _, _ = synth.make_topo_set(seed=33, D=13, NU=1)
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk2, tnames)
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(tnames) + "\n")
    # models: a b sp(tee) c d e ; words end with the optional pause model
    open(os.path.join(d, "dict"), "w").write("W1 a sp\nW2 b c sp\nW3 d\nW4 e a sp\nW5 c\n")
    open(os.path.join(d, "wlist"), "w").write("W1\nW2\nW3\nW4\nW5\n")
    subprocess.check_call([os.path.join(REF, "HBuild"), "wlist", "net.slf"], cwd=d)
    os.remove(os.path.join(d, "wlist"))
    idx = {n: i for i, n in enumerate(tnames)}
    wp = {"W1": ["a", "sp"], "W2": ["b", "c", "sp"], "W3": ["d"], "W4": ["e", "a", "sp"], "W5": ["c"]}
    feats = []
    for u in range(4):
        seq = [list(wp)[k] for k in rng.integers(0, 5, size=5)]
        ph = []
        for w in seq:
            for p in wp[w]:
                if p == "sp" and rng.random() < 0.5:
                    continue                                                    # pause skipped
                ph.append(idx[p])
        feats.append(sample(pk2, ph, rng, frames_per_state=3))
    run_hvite(d, feats, ["-t 250.0", "-t 40.0", "-v 20.0 -p -19.0", "-t 250.0 -v 40.0 -r 2.0", "-v 20.0 -p -21.0", "-u 5", "-u 9 -t 250.0 -v 40.0"], 9, "")
    # ---------------------------------------------------------------- wint: word-internal triphones
    d = os.path.join(OUT, "wint"); os.makedirs(d, exist_ok=True)
    synth.write_mmf(os.path.join(d, "MMF"), s, kind="USER")
    phones = ["a", "b", "c", "d"]
    logical = []
    k = 0
    for p_ in phones:
        for l_ in [None] + phones:
            for r_ in [None] + phones:
                if l_ is None and r_ is None:
                    continue                                                    # bare monophones: only "b" below
                name = ("%s-" % l_ if l_ else "") + p_ + ("+%s" % r_ if r_ else "")
                logical.append((name, "p%d" % (k % 10))); k += 1
    logical += [("b", "p3"), ("sil", "p10"), ("sp", "p11")]
    with open(os.path.join(d, "hmmlist"), "w") as f:
        for n in names:
            f.write(n + "\n")                                                   # the physical models themselves
        for lg, ph in logical:
            f.write("%s %s\n" % (lg, ph))
    open(os.path.join(d, "dict"), "w").write("SIL sil\nW1 a b sp\nW2 c d a sp\nW3 b sp\nW4 d c b a sp\n")
    open(os.path.join(d, "wlist"), "w").write("SIL\nW1\nW2\nW3\nW4\n")
    subprocess.check_call([os.path.join(REF, "HBuild"), "wlist", "net.slf"], cwd=d)
    os.remove(os.path.join(d, "wlist"))
    lmap = dict(logical)
    wseq = {"SIL": ["sil"], "W1": ["a+b", "a-b", "sp"], "W2": ["c+d", "c-d+a", "d-a", "sp"], "W3": ["b", "sp"], "W4": ["d+c", "d-c+b", "c-b+a", "b-a", "sp"]}
    feats = []
    for u in range(4):
        seq = ["SIL"] + [list(wseq)[1 + k] for k in rng.integers(0, 4, size=4)] + ["SIL"]
        ph = [int(lmap[m][1:]) for w in seq for m in wseq[w]]
        feats.append(sample(pk, ph, rng, frames_per_state=3))
    run_hvite(d, feats, ["-t 250.0", "-t 250.0 -p -20.0 -s 3.0", "-m -t 250.0"], 9, "")
    # ---------------------------------------------------------------- ties: inputs are committed (recorded from the sweep), expectations from HVite
    d = os.path.join(OUT, "ties")
    z = np.load(os.path.join(d, "feats.npz"))
    feats = [z["u%d" % u] for u in range(len(z.files))]
    run_hvite(d, feats, ["-t 83.5292734595038 -v 33.403143058686354 -r 1.7961669175753967", "-r 1.7961669175753967", "-t 83.5292734595038 -r 2.0"], 9, "")
    for c in ("loop", "bigram", "tee", "wint", "ties"):
        e = json.load(open(os.path.join(OUT, c, "expected.json")))
        print(c, {k: sum(len(v) for v in per.values()) for k, per in e.items()})


if __name__ == "__main__":
    main()
