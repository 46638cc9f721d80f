#!/usr/bin/env python3
"""What the reference writes into the alignment field of lattice arcs (LatFromPaths' lAlign, HRec.c:1590-1640; WriteLattice `-q d`,
HNet.c:503-516) when HVite -- built with -DPHNALG, as HTKTools/Makefile.in:45 builds it -- is given -m together with -n: per arc
`d=:model,duration,likelihood:`.  This script counts, on the committed decode case `bigram` and on a demo test file, the arcs whose
alignment likelihoods do not add up to the arc's own acoustic likelihood `a=` (to 0.02): about two arcs in three, positive "log
likelihoods" among them.  The records of a relative token are differences of likelihoods taken from align records of OTHER tokens of the
set once TokSetMerge has re-based them; only the best token's arcs are consistent.  That is why lattice alignment records are not
restated: there is no function of the lattice to be equal to.
    python tests/golden/check_ref_lalign.py        (needs oracle/_ref)"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from htk_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "HVite")
DEC = os.path.join(ROOT, "tests", "golden", "decode", "bigram")
DEMO = os.path.join(ROOT, "tests", "golden", "demo")


def count(path):
    n = bad = pos = 0
    for line in open(path):
        if not line.startswith("J="):
            continue
        a, d = re.search(r"a=(\S+)", line), re.search(r"d=:(.*):", line)
        if not d:
            continue
        recs = [r.split(",") for r in d.group(1).split(":")]
        s = sum(float(r[2]) for r in recs)
        n += 1; bad += abs(s - float(a.group(1))) > 0.02; pos += any(float(r[2]) > 0 for r in recs)
    return n, bad, pos


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        z = np.load(os.path.join(DEC, "feats.npz"))
        files = []
        for u in range(len(z.files)):
            synth.write_htk_param(os.path.join(d, "u%d.mfc" % u), z["u%d" % u], kind=9); files.append(os.path.join(d, "u%d.mfc" % u))
        subprocess.run([REF, "-H", "MMF", "-w", "net.slf", "-t", "250.0", "-l", d, "-n", "4", "3", "-m", "-z", "lat", "-q", "tvaldmn", "dict", "hmmlist"] + files,
                       cwd=DEC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cfg = os.path.join(d, "cfg"); open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        subprocess.run([REF, "-C", cfg, "-d", os.path.join(DEMO, "hmm_final"), "-w", os.path.join(DEMO, "monLattice"), "-l", d, "-t", "300.0", "-p", "5.0", "-s", "0.0",
                        "-n", "4", "2", "-m", "-z", "lat", "-q", "tvaldmn", os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist"), os.path.join(DEMO, "test", "te1.mfc")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in ["u%d.lat" % u for u in range(len(z.files))] + ["te1.lat"]:
            n, bad, pos = count(os.path.join(d, f))
            print("%-8s arcs with alignment %4d   likelihoods not adding up to a= %4d   with a positive log likelihood %3d" % (f, n, bad, pos))
