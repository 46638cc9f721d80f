"""Known answers for N-best recognition and lattice output (HVite -n N [M] -z lat; HRec.c TokSetMerge :279, CreateLattice :1679,
WriteLattice HNet.c:631, TranscriptionFromLattice HRec.c:2176): the reference's HVite on the committed decode cases.
    python tests/golden/make_nbest_golden.py         (needs oracle/_ref)  -> tests/golden/decode/nbest/<case>/{u*.lat, nbest.json}
HVite runs inside the case directory with relative file names, so the header lines of the lattices (UTTERANCE=, lmname=, vocab=) do not
depend on where the repository lives."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from htk_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
DEC = os.path.join(HERE, "decode")
# case, lattice, features, tokens, transcriptions, other switches
CASES = [("bigram", "net", "feats", 4, 3, "-t 250.0"), ("bigram", "net", "feats", 2, 2, "-t 250.0 -s 2.0 -p -5.0 -r 1.5"), ("loop", "net", "feats", 3, 3, "-t 250.0"),
         ("tee", "net", "feats", 3, 2, "-t 250.0"), ("xwrd", "loop", "feats_loop", 4, 3, "-t 250.0"),
         ("bigram", "net", "feats", 3, 2, "-t 250.0 -u 6"), ("loop", "net", "feats", 3, 3, "-t 250.0 -u 5"), ("tee", "net", "feats", 4, 2, "-t 250.0 -u 4"),   # -u: maximum-model pruning with token sets
         # -m / -f together with -n (the reference is built with -DPHNALG): alignment records inside the lattice arcs (d=), model / state level alternatives
         ("loop", "net", "feats", 3, 3, "-t 250.0 -m"), ("bigram", "net", "feats", 4, 2, "-t 250.0 -f"), ("tee", "net", "feats", 3, 2, "-t 250.0 -m -f"), ("wint", "net", "feats", 3, 2, "-t 250.0 -m")]


def main():
    out_root = os.path.join(DEC, "nbest")
    os.makedirs(out_root, exist_ok=True)
    index = []
    for case, slf, feats, ntok, ntrans, opts in CASES:
        d = os.path.join(DEC, case)
        tag = "%s_n%d_%s" % (case, ntok, "".join(c for c in opts if c.isalnum()))
        out = os.path.join(out_root, tag); os.makedirs(out, exist_ok=True)
        tmp = os.path.join(d, "nbtmp"); os.makedirs(tmp, exist_ok=True)
        z = np.load(os.path.join(d, feats + ".npz"))
        scp = []
        for u in range(len(z.files)):
            synth.write_htk_param(os.path.join(tmp, "u%d.mfc" % u), z["u%d" % u], kind=9)
            scp.append("nbtmp/u%d.mfc" % u)
        open(os.path.join(tmp, "scp"), "w").write("\n".join(scp) + "\n")
        cfg = "config" if os.path.exists(os.path.join(d, "config")) else "nbtmp/config"
        if cfg != "config":
            open(os.path.join(tmp, "config"), "w").write("")
        base = [os.path.join(REF, "HVite"), "-C", cfg, "-H", "MMF", "-S", "nbtmp/scp", "-w", slf + ".slf"] + opts.split()
        subprocess.run(base + ["-l", "nbtmp", "-n", str(ntok), "1", "-z", "lat", "dict", "hmmlist"], cwd=d, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        subprocess.run(base + ["-i", "nbtmp/nb.mlf", "-n", str(ntok), str(ntrans), "dict", "hmmlist"], cwd=d, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        nb, cur = {}, None
        for line in open(os.path.join(tmp, "nb.mlf")).read().splitlines()[1:]:
            if line.startswith('"'):
                cur = os.path.basename(line.strip('"')).replace(".rec", ""); nb[cur] = [[]]
            elif line == ".":
                cur = None
            elif line == "///":
                nb[cur].append([])
            elif cur is not None:
                nb[cur][-1].append(line)
        for u in range(len(z.files)):
            if os.path.exists(os.path.join(tmp, "u%d.lat" % u)):
                shutil.copy(os.path.join(tmp, "u%d.lat" % u), os.path.join(out, "u%d.lat" % u))
        json.dump(dict(case=case, slf=slf, feats=feats, nToks=ntok, nTrans=ntrans, opts=opts, nbest=nb), open(os.path.join(out, "nbest.json"), "w"), indent=1)
        shutil.rmtree(tmp)
        index.append(tag)
        print(tag, {u: [len(a) for a in alts] for u, alts in nb.items()})
    json.dump(index, open(os.path.join(out_root, "index.json"), "w"))


if __name__ == "__main__":
    main()
