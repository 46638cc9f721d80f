#!/usr/bin/env python3
"""HVite on tied-mixture sets (hsKind TIEDHS): per frame PrecomputeTMix with the -c threshold (tmBeam, HRec.c:1987), per state SOutP's sum
over the kept pool entries (HRec.c:493-503), per state the stream-weighted sum (cPOutP).  The demo's re-estimated <TMIX> sets
(tests/golden/demo/hmm_tmix/tiedhs_after_herest, tiedhs3_after_herest) through the reference's HVite:
    rec    recognition of the test and training files with the loop lattice (-t 300.0 -p 5.0 -s 0.0 -m -f), default -c and -c 3.0
    align  forced alignment of the training files (-a -m -f)
    tests/golden/demo/hmm_tmix/hvite_expected.json
    python tests/golden/make_tmix_hvite_golden.py"""
import glob
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
REF = os.path.join(ROOT, "oracle", "_ref")

if __name__ == "__main__":
    tm = os.path.join(DEMO, "hmm_tmix")
    test = sorted(glob.glob(os.path.join(DEMO, "test", "*.mfc"))); train = sorted(glob.glob(os.path.join(DEMO, "train", "*.mfc")))
    out = {}
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "cfg"); open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        for name in ("tiedhs_after_herest", "tiedhs3_after_herest"):
            per = {}
            for what, files, opts in (("rec", test + train, ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0", "-m", "-f"]),
                                      ("rec_c3", test + train, ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0", "-c", "3.0"]),
                                      ("align", train, ["-a", "-m", "-f", "-L", os.path.join(DEMO, "labels"), "-t", "300.0"])):
                od = os.path.join(d, name + "_" + what); os.makedirs(od)
                subprocess.run([os.path.join(REF, "HVite"), "-C", cfg, "-H", os.path.join(tm, name), "-l", od] + opts + [os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + files,
                               check=True, stdout=subprocess.DEVNULL)
                per[what] = {os.path.basename(f)[:-4]: open(os.path.join(od, os.path.basename(f)[:-4] + ".rec")).read().splitlines() for f in files}
            out[name] = per
            a, b = per["rec_c3"], {k: [" ".join(l.split()[:4]) for l in v] for k, v in per["rec"].items()}
    json.dump(out, open(os.path.join(tm, "hvite_expected.json"), "w"), indent=0)
    print({k: {w: len(v) for w, v in per.items()} for k, per in out.items()})
