#!/usr/bin/env python3
"""Multi-stream sets (<STREAMINFO> S > 1) through the reference: its HHEd splits the demo's 3-mixture monophones into streams
(SS 3 -> 12 | 12 | 2 with the energy terms as the last stream; SS 2 -> 13 | 13), its HERest makes one embedded pass.
    tests/golden/demo/hmm_streams3/  newMacros (the set), after_herest (re-estimated), HER1.acc (`-p 1` accumulators), herest.log, stats (`-s`)
    tests/golden/demo/hmm_streams2/  newMacros, HER1.acc, herest.log, stats, after_herest
S = 3 is the one stream count for which HFB.c's Setotprob is consistent: on meeting a tied state for the second time at a frame it
takes `sum/2` of the streams' REPLACED values (HFB.c:1044,1059-1064) = (S-1)/2 times the state's log probability.  The S = 2 fixture
pins oracle/htk_oracle.c's restatement of exactly that (average log probability -33.6 where the intended arithmetic gives -59.1).
    python tests/golden/make_streams_golden.py"""
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
    for S in (3, 2):
        out = os.path.join(DEMO, "hmm_streams%d" % S)
        os.makedirs(out, exist_ok=True)
        with tempfile.TemporaryDirectory() as d:
            hed = os.path.join(d, "ss.hed"); open(hed, "w").write("SS %d\n" % S)
            subprocess.check_call([os.path.join(REF, "HHEd"), "-H", os.path.join(DEMO, "hmm_mixup", "newMacros"), "-M", out, hed, os.path.join(DEMO, "bcplist")])
            cfg = os.path.join(d, "cfg"); open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
            base = [os.path.join(REF, "HERest"), "-C", cfg, "-w", "3", "-v", "0.05", "-u", "tmvw", "-H", os.path.join(out, "newMacros"),
                    "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", "-T", "1"]
            os.makedirs(os.path.join(d, "next")); os.makedirs(os.path.join(d, "acc"))
            log = subprocess.run(base + ["-s", os.path.join(d, "stats"), "-M", os.path.join(d, "next"), os.path.join(DEMO, "bcplist")] + files, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
            shutil.copy(os.path.join(d, "stats"), os.path.join(out, "stats"))
            subprocess.run(base + ["-M", os.path.join(d, "acc"), "-p", "1", os.path.join(DEMO, "bcplist")] + files, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True)
            shutil.copy(os.path.join(d, "next", "newMacros"), os.path.join(out, "after_herest"))
            shutil.copy(os.path.join(d, "acc", "HER1.acc"), os.path.join(out, "HER1.acc"))
            keep = [l for l in log.splitlines() if "average log prob" in l or "floored variance" in l]
            open(os.path.join(out, "herest.log"), "w").write("\n".join(keep) + "\n")
            print(S, "\n".join(keep))
        print(sorted(os.listdir(out)))
