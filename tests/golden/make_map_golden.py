#!/usr/bin/env python3
"""Known answer for MAP re-estimation (HERest -u p..., MAPUpdateModels HMap.c:413): the reference's HERest (oracle/_ref) makes one
embedded pass over HTKDemo's training files from the 3-component set tests/golden/demo/hmm_mixup/newMacros with
    -u pmvw   and   HMAP: MAPTAU = 6.0, HMAP: MINVAR = 0.02, HMAP: MIXWEIGHTFLOOR = 2.0, HMAP: TRACE = 1
(a second run: -u pm with the default MAPTAU = 20).
    python tests/golden/make_map_golden.py   -> tests/golden/demo/hmm_map/{after_pmvw, after_pm, herest_pmvw.log, herest_pm.log}"""
import glob
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
RUNS = {"pmvw": "TARGETKIND = MFCC_E_D\nHMAP: MAPTAU = 6.0\nHMAP: MINVAR = 0.02\nHMAP: MIXWEIGHTFLOOR = 2.0\nHMAP: TRACE = 1\n",
        "pm": "TARGETKIND = MFCC_E_D\nHMAP: TRACE = 1\n",
        "tied_pmv": "TARGETKIND = MFCC_E_D\nHMAP: MAPTAU = 3.0\nHMAP: TRACE = 1\n"}          # the set with ~u / ~v vectors (hmm_tied/newMacros)

if __name__ == "__main__":
    out = os.path.join(DEMO, "hmm_map")
    os.makedirs(out, exist_ok=True)
    for flags, conf in RUNS.items():
        with tempfile.TemporaryDirectory() as d:
            cfg = os.path.join(d, "cfg")
            open(cfg, "w").write(conf)
            os.makedirs(os.path.join(d, "next"))
            src = os.path.join(DEMO, "hmm_tied" if flags.startswith("tied_") else "hmm_mixup", "newMacros")
            log = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "HERest"), "-C", cfg, "-u", flags.replace("tied_", ""), "-H", src,
                                  "-M", os.path.join(d, "next"), "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", "-T", "1", os.path.join(DEMO, "bcplist")] +
                                 sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc"))), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
            shutil.copy(os.path.join(d, "next", "newMacros"), os.path.join(out, "after_" + flags))
            keep = [l for l in log.splitlines() if "average log prob" in l or "floored variance" in l or "Observed components" in l or "MAP Updating" in l]
            open(os.path.join(out, "herest_%s.log" % flags), "w").write("\n".join(keep) + "\n")
            print(flags, "\n".join(keep))
    print(sorted(os.listdir(out)))
