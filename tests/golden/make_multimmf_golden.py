#!/usr/bin/env python3
"""Known answer for a set loaded from SEVERAL master files (the usual `-H hmm0/macros -H hmm0/hmmdefs -M hmm1` flow): the tied set of
tests/golden/demo/hmm_tied cut into `macros` (options + the ~u / ~v vectors) and `hmmdefs` (the models), put through the reference's
HHEd with an empty script and through the reference's HERest: SaveHMMSet writes every macro back to the file it was loaded from
(HModel.c:4388-4470), so both produce out/macros AND out/hmmdefs.
    python tests/golden/make_multimmf_golden.py   -> tests/golden/demo/hmm_multi/{macros, hmmdefs, hhed_macros, hhed_hmmdefs, herest_macros, herest_hmmdefs}"""
import glob
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
REF = os.path.join(ROOT, "oracle", "_ref")

if __name__ == "__main__":
    out = os.path.join(DEMO, "hmm_multi")
    os.makedirs(out, exist_ok=True)
    lines = open(os.path.join(DEMO, "hmm_tied", "newMacros")).read().splitlines(keepends=True)
    cut = next(i for i, l in enumerate(lines) if l.startswith("~h"))
    open(os.path.join(out, "macros"), "w").writelines(lines[:cut])
    open(os.path.join(out, "hmmdefs"), "w").writelines(lines[:3] + lines[cut:])          # the options again, then the models
    with tempfile.TemporaryDirectory() as d:
        hed = os.path.join(d, "empty.hed"); open(hed, "w").write("")
        o1 = os.path.join(d, "o1"); os.makedirs(o1)
        subprocess.check_call([os.path.join(REF, "HHEd"), "-H", os.path.join(out, "macros"), "-H", os.path.join(out, "hmmdefs"), "-M", o1, hed, os.path.join(DEMO, "bcplist")])
        shutil.copy(os.path.join(o1, "macros"), os.path.join(out, "hhed_macros")); shutil.copy(os.path.join(o1, "hmmdefs"), os.path.join(out, "hhed_hmmdefs"))
        cfg = os.path.join(d, "cfg"); open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        o2 = os.path.join(d, "o2"); os.makedirs(o2)
        subprocess.check_call([os.path.join(REF, "HERest"), "-C", cfg, "-w", "3", "-v", "0.05", "-u", "tmvw", "-H", os.path.join(out, "macros"), "-H", os.path.join(out, "hmmdefs"),
                               "-M", o2, "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc"))),
                              stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(o2, "macros"), os.path.join(out, "herest_macros")); shutil.copy(os.path.join(o2, "hmmdefs"), os.path.join(out, "herest_hmmdefs"))
    print(sorted(os.listdir(out)))
