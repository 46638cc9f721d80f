#!/usr/bin/env python3
"""Known answer for shared mean / variance vectors (~u / ~v macros, HModel.c:1737-1790 GetMean / GetVariance): the reference's HHEd
(oracle/_ref) ties vectors of HTKDemo's 3-component models (tests/golden/demo/hmm_mixup/newMacros) with
    TI vCL {(C,L).state[3].mix[1-2].cov}      two variances across two models
    TI uSV {(S,V).state[2].mix[1].mean}       one mean across two models
    TI vN  {N.state[2-4].mix[1].cov}          one variance inside a model
and the reference's HERest makes one embedded pass from the tied set.
    python tests/golden/make_tied_golden.py   -> tests/golden/demo/hmm_tied/{newMacros, after_herest, herest.log}"""
import glob
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
HED = "TI vCL {(C,L).state[3].mix[1-2].cov}\nTI uSV {(S,V).state[2].mix[1].mean}\nTI vN {N.state[2-4].mix[1].cov}\n"

if __name__ == "__main__":
    out = os.path.join(DEMO, "hmm_tied")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        hed = os.path.join(d, "ti.hed")
        open(hed, "w").write(HED)
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "HHEd"), "-H", os.path.join(DEMO, "hmm_mixup", "newMacros"), "-M", out, hed, os.path.join(DEMO, "bcplist")])
        cfg = os.path.join(d, "cfg")
        open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        os.makedirs(os.path.join(d, "next"))
        log = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "HERest"), "-C", cfg, "-w", "3", "-v", "0.05", "-u", "tmvw", "-H", os.path.join(out, "newMacros"),
                              "-M", os.path.join(d, "next"), "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", "-T", "1", os.path.join(DEMO, "bcplist")] +
                             sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc"))), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
        shutil.copy(os.path.join(d, "next", "newMacros"), os.path.join(out, "after_herest"))
        keep = [l for l in log.splitlines() if "average log prob" in l or "floored variance" in l]
        open(os.path.join(out, "herest.log"), "w").write("\n".join(keep) + "\n")
        print("\n".join(keep))
    print(sorted(os.listdir(out)))
