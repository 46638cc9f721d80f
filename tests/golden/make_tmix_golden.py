#!/usr/bin/env python3
"""Tied-mixture systems the way the HTK book builds them (HHEd: JO 8 2.0 / TI MIX_ {*.state[2-4].mix} / HK TIEDHS): every state's mixture is
the same pool of Gaussians (~m "TM_1_1" ..) with its own weights, written as <TMIX>; hsKind TIEDHS: PrecomputeTMix's top-M arithmetic
(HModel.c:5308), SOutP's linear sum (:5555), UpMixParms' TIEDHS branches (HFB.c:1503,1559,1597), the pool re-estimated once per set
(HERest.c:1272).
    tests/golden/demo/hmm_tmix/  tiedhs_newMacros, tiedhs_after_herest, tiedhs_HER1.acc (`-p 1`), tiedhs.log   one stream, pool of 8
                                 tiedhs3_*: the same on the 3-stream set (pools of 4 | 4 | 2)
                                 newMacros, after_herest, herest.log: the pool WITHOUT `HK TIEDHS` (hsKind SHAREDHS, ~m macros in ordinary
                                 mixtures) through the reference -- for the record: its ConvLogWt (HUtil.c:474-485, GoNextMix with
                                 noSkip = FALSE) converts the weights of the FIRST state that uses a shared pdf only, every other state's
                                 linear weights are then read as log weights: -59.47 per frame where the arithmetic gives -61.00
    python tests/golden/make_tmix_golden.py"""
import glob
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
REF = os.path.join(ROOT, "oracle", "_ref")

if __name__ == "__main__":
    files = sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc")))
    out = os.path.join(DEMO, "hmm_tmix")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "cfg"); open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        S3 = ("JO 4 2.0\nTI MIXA_ {*.state[2-4].stream[1].mix}\nJO 4 2.0\nTI MIXB_ {*.state[2-4].stream[2].mix}\nJO 2 2.0\n"
              "TI MIXC_ {*.state[2-4].stream[3].mix}\nHK TIEDHS\n")
        for kind, src, hed in (("shared", "hmm_mixup", "JO 8 2.0\nTI MIX_ {*.state[2-4].mix}\n"), ("tiedhs", "hmm_mixup", "JO 8 2.0\nTI MIX_ {*.state[2-4].mix}\nHK TIEDHS\n"),
                               ("tiedhs3", "hmm_streams3", S3)):
            sd = os.path.join(d, kind); os.makedirs(sd); os.makedirs(os.path.join(sd, "next")); os.makedirs(os.path.join(sd, "acc"))
            open(os.path.join(sd, "e.hed"), "w").write(hed)
            subprocess.check_call([os.path.join(REF, "HHEd"), "-H", os.path.join(DEMO, src, "newMacros"), "-M", sd, os.path.join(sd, "e.hed"), os.path.join(DEMO, "bcplist")])
            base = [os.path.join(REF, "HERest"), "-C", cfg, "-w", "3", "-v", "0.05", "-u", "tmvw", "-H", os.path.join(sd, "newMacros"),
                    "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", "-T", "1"]
            log = subprocess.run(base + ["-M", os.path.join(sd, "next"), os.path.join(DEMO, "bcplist")] + files,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
            if kind != "shared":
                subprocess.run(base + ["-M", os.path.join(sd, "acc"), "-p", "1", os.path.join(DEMO, "bcplist")] + files, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True)
                shutil.copy(os.path.join(sd, "acc", "HER1.acc"), os.path.join(out, kind + "_HER1.acc"))
            keep = [l for l in log.splitlines() if "average log prob" in l or "floored variance" in l]
            print(kind, "\n".join(keep))
            if kind == "shared":
                shutil.copy(os.path.join(sd, "newMacros"), os.path.join(out, "newMacros"))
                shutil.copy(os.path.join(sd, "next", "newMacros"), os.path.join(out, "after_herest"))
                open(os.path.join(out, "herest.log"), "w").write("\n".join(keep) + "\n")
            else:
                shutil.copy(os.path.join(sd, "newMacros"), os.path.join(out, kind + "_newMacros"))
                shutil.copy(os.path.join(sd, "next", "newMacros"), os.path.join(out, kind + "_after_herest"))
                open(os.path.join(out, kind + ".log"), "w").write("\n".join(keep) + "\n")
    print(sorted(os.listdir(out)))
