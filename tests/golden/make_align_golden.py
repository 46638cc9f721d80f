"""Known answers for HVite -a from WORD-level transcriptions with a multi-pronunciation dictionary (DoAlignment HVite.c:830):
the reference's HVite on the `bigram` case's model set and dictionary (tests/golden/decode/bigram: AB has two pronunciations
with probabilities, E has an output symbol, F none).
    python tests/golden/make_align_golden.py        (needs oracle/_ref, i.e. `make -C oracle`)
Writes tests/golden/decode/align/{feats.npz, words.json, expected.json}."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from htk_amd import synth  # noqa: E402
from make_decode_golden import sample  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
SRC = os.path.join(HERE, "decode", "bigram")
OUT = os.path.join(HERE, "decode", "align")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(77)
    s = synth.generate(30, 3, 12, 3, 80, 11, D=13)               # the model set of the loop/bigram cases
    pk = s.packed()
    prons = {"AB": [[0, 1], [0, 2]], "CD": [[3, 4, 5]], "E": [[6]], "F": [[7]], "G": [[8, 9]], "H": [[10]], "I": [[11]]}
    words = list(prons)
    feats, trans = [], []
    for u in range(4):
        seq = [words[k] for k in rng.integers(0, len(words), size=6)]
        if u == 0:
            seq = ["AB", "E", "AB", "F", "CD", "AB"]
        ph = []
        for w in seq:
            alt = prons[w]
            ph += alt[int(rng.integers(0, len(alt)))]
        feats.append(sample(pk, ph, rng, frames_per_state=3))
        trans.append(seq)
    expected = {}
    with tempfile.TemporaryDirectory() as d:
        scp = []
        with open(os.path.join(d, "words.mlf"), "w") as f:
            f.write("#!MLF!#\n")
            for u, (X, seq) in enumerate(zip(feats, trans)):
                fn = os.path.join(d, "u%d.mfc" % u)
                synth.write_htk_param(fn, X, kind=9)
                scp.append(fn)
                f.write('"*/u%d.lab"\n%s\n.\n' % (u, "\n".join(seq)))
        open(os.path.join(d, "scp"), "w").write("\n".join(scp) + "\n")
        open(os.path.join(d, "config"), "w").write("")
        for opts in ["-t 250.0", "-m -t 250.0", "-b H -m -t 250.0", "-m -r 3.0 -t 60.0"]:
            mlf = os.path.join(d, "out.mlf")
            cmd = [os.path.join(REF, "HVite"), "-a", "-C", os.path.join(d, "config"), "-H", os.path.join(SRC, "MMF"), "-S", os.path.join(d, "scp"),
                   "-I", os.path.join(d, "words.mlf"), "-i", mlf] + opts.split() + [os.path.join(SRC, "dict"), os.path.join(SRC, "hmmlist")]
            subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            per, cur = {}, None
            for line in open(mlf).read().splitlines()[1:]:
                if line.startswith('"'):
                    cur = os.path.basename(line.strip('"')).replace(".rec", ""); per[cur] = []
                elif line == ".":
                    cur = None
                elif cur is not None:
                    per[cur].append(line)
            expected[opts] = per
    np.savez_compressed(os.path.join(OUT, "feats.npz"), **{"u%d" % u: X for u, X in enumerate(feats)})
    json.dump(trans, open(os.path.join(OUT, "words.json"), "w"))
    json.dump(expected, open(os.path.join(OUT, "expected.json"), "w"), indent=1)
    for k, per in expected.items():
        print(k, per["u0"][:4])


if __name__ == "__main__":
    main()
