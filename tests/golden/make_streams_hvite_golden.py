#!/usr/bin/env python3
"""HVite on a multi-stream set: the state output probability is the stream-WEIGHTED sum of the streams' mixture log likelihoods
(cPOutP HRec.c:510-548 / POutP HModel.c:5570-5583: outp += w[s] * SOutP_s, float).  The demo's 3-stream set after one pass of HERest
(tests/golden/demo/hmm_streams3/after_herest), once as it is (weights 1) and once with <SWEIGHTS> 1.0 0.5 2.0 in every state, through
the reference's HVite:
    rec    recognition of the test and training files with the demo's loop lattice (-t 300.0 -p 5.0 -s 0.0), with -m -f; recw: without
    align  forced alignment of the training files (-a -m -f) from their label files
    tests/golden/demo/hmm_streams3/hvite_expected.json, sw_after_herest (the re-weighted set)
    python tests/golden/make_streams_hvite_golden.py"""
import glob
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
REF = os.path.join(ROOT, "oracle", "_ref")

if __name__ == "__main__":
    d3 = os.path.join(DEMO, "hmm_streams3")
    src = open(os.path.join(d3, "after_herest")).read()
    sw = src.replace("<SWEIGHTS> 3\n 1.000000e+00 1.000000e+00 1.000000e+00", "<SWEIGHTS> 3\n 1.000000e+00 5.000000e-01 2.000000e+00")
    assert sw != src
    open(os.path.join(d3, "sw_after_herest"), "w").write(sw)
    test = sorted(glob.glob(os.path.join(DEMO, "test", "*.mfc"))); train = sorted(glob.glob(os.path.join(DEMO, "train", "*.mfc")))
    out = {}
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "cfg"); open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        for name in ("after_herest", "sw_after_herest"):
            per = {}
            for what, files, opts in (("rec", test + train, ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0", "-m", "-f"]),
                                      ("recw", test + train, ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0"]),
                                      ("align", train, ["-a", "-m", "-f", "-L", os.path.join(DEMO, "labels"), "-t", "300.0"])):
                od = os.path.join(d, name + "_" + what); os.makedirs(od)
                subprocess.run([os.path.join(REF, "HVite"), "-C", cfg, "-H", os.path.join(d3, name), "-l", od] + opts + [os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + files,
                               check=True, stdout=subprocess.DEVNULL)
                per[what] = {os.path.basename(f)[:-4]: open(os.path.join(od, os.path.basename(f)[:-4] + ".rec")).read().splitlines() for f in files}
            out[name] = per
    json.dump(out, open(os.path.join(d3, "hvite_expected.json"), "w"), indent=0)
    print({k: {w: len(v) for w, v in per.items()} for k, per in out.items()})
    a, b = out["after_herest"]["rec"], out["sw_after_herest"]["rec"]
    print("files whose recognition differs between the two weightings:", sum(a[k] != b[k] for k in a), "of", len(a))
