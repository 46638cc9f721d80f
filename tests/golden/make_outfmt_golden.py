"""Known answers for HVite's output formatting (-o flags, -m, -i MLF): the reference's HVite on the committed decode cases
`loop` (word loop) and `wint` (word-internal triphones), one MLF per option set kept verbatim.
    python tests/golden/make_outfmt_golden.py      -> tests/golden/decode/outfmt/{case}__{opts}.mlf"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from htk_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "decode", "outfmt")
CASES = {"loop": ["-o N", "-o S", "-o ST", "-o C", "-m -o W", "-m -o N", "-m", "-f -m", "-f", "-f -m -o N", "-f -m -o M", "-f -m -o WS"],
         "wint": ["-m -o X", "-m -o SWX", "-o TS", "-f -m -o X"]}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for case, optsets in CASES.items():
        d = os.path.join(HERE, "decode", case)
        z = np.load(os.path.join(d, "feats.npz"))
        with tempfile.TemporaryDirectory() as tmp:
            scp = []
            for u in range(len(z.files)):
                fn = os.path.join(tmp, "u%d.mfc" % u)
                synth.write_htk_param(fn, z["u%d" % u], kind=9)
                scp.append(fn)
            open(os.path.join(tmp, "scp"), "w").write("\n".join(scp) + "\n")
            open(os.path.join(tmp, "config"), "w").write("")
            for opts in optsets:
                mlf = os.path.join(tmp, "out.mlf")
                subprocess.run([os.path.join(REF, "HVite"), "-C", os.path.join(tmp, "config"), "-H", os.path.join(d, "MMF"), "-S", os.path.join(tmp, "scp"),
                                "-i", mlf, "-w", os.path.join(d, "net.slf"), "-t", "250.0"] + opts.split() + [os.path.join(d, "dict"), os.path.join(d, "hmmlist")],
                               check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                txt = open(mlf).read().replace(tmp + "/", "*/")
                open(os.path.join(OUT, "%s__%s.mlf" % (case, opts.replace(" ", "_"))), "w").write(txt)
    print(sorted(os.listdir(OUT)))
