#!/usr/bin/env python3
"""Known answer for mixture splitting: the reference's HHEd (oracle/_ref) run on HTKDemo's final single-Gaussian models
(tests/golden/demo/hmm_final) with the script
    MU 3 {*.state[2-4].mix}
    MU +2 {S.state[2].mix}
    python tests/golden/make_mixup_golden.py        -> tests/golden/demo/hmm_mixup/newMacros (HHEd saves the edited set as one file)"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")

if __name__ == "__main__":
    out = os.path.join(DEMO, "hmm_mixup")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        hed = os.path.join(d, "mu.hed")
        open(hed, "w").write("MU 3 {*.state[2-4].mix}\nMU +2 {S.state[2].mix}\n")
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "HHEd"), "-d", os.path.join(DEMO, "hmm_final"), "-M", out, hed, os.path.join(DEMO, "bcplist")])
        # one embedded pass of the reference's HERest from the split set (the demo's switches)
        cfg = os.path.join(d, "cfg")
        open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        os.makedirs(os.path.join(d, "next"))
        import glob
        log = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "HERest"), "-C", cfg, "-w", "3", "-v", "0.05", "-u", "tmvw", "-H", os.path.join(out, "newMacros"),
                              "-M", os.path.join(d, "next"), "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", "-T", "1", os.path.join(DEMO, "bcplist")] +
                             sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc"))), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
        import shutil
        shutil.copy(os.path.join(d, "next", "newMacros"), os.path.join(out, "after_herest"))
        keep = [l for l in log.splitlines() if "average log prob" in l or "floored variance" in l]
        open(os.path.join(out, "herest.log"), "w").write("\n".join(keep) + "\n")
        print("\n".join(keep))
    print(sorted(os.listdir(out)))
