#!/usr/bin/env python3
"""Golden output of the reference's HCompV (oracle/_ref/HCompV, built from /root/reference by oracle/Makefile) on HTKDemo's
seven training files with TARGETKIND = MFCC_E_D: the flat-start model (global mean and variance in every state, -m) and the
variance floor macro file (-f 0.01).

    python tests/golden/make_compv_golden.py
"""
import glob
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
OUT = os.path.join(ROOT, "tests", "golden", "compv")

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "cfg")
        open(cfg, "w").write("TARGETKIND = MFCC_E_D\n")
        os.makedirs(os.path.join(d, "out"))
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "HCompV"), "-C", cfg, "-f", "0.01", "-m", "-M", os.path.join(d, "out"),
                               os.path.join(DEMO, "hmm1", "S")] + sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc"))))
        for f in ("S", "vFloors"):
            shutil.copy(os.path.join(d, "out", f), os.path.join(OUT, f))
        # flat start: the five monophones are copies of that model, the variance floor macro rides in the same file;
        # one embedded re-estimation pass of the reference's HERest from there
        body = open(os.path.join(d, "out", "S")).read()
        opts, proto = body[:body.index('~h "S"')], body[body.index('~h "S"'):]
        names = open(os.path.join(DEMO, "bcplist")).read().split()
        with open(os.path.join(OUT, "flat_hmm0.mmf"), "w") as f:
            f.write(opts + open(os.path.join(d, "out", "vFloors")).read())
            for n in names:
                f.write(proto.replace('~h "S"', '~h "%s"' % n))
        os.makedirs(os.path.join(d, "hmm1"))
        log = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "HERest"), "-C", cfg, "-H", os.path.join(OUT, "flat_hmm0.mmf"), "-M", os.path.join(d, "hmm1"),
                              "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", "-T", "1", "-s", os.path.join(OUT, "flat_stats"), os.path.join(DEMO, "bcplist")] +
                             sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc"))), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
        shutil.copy(os.path.join(d, "hmm1", "flat_hmm0.mmf"), os.path.join(OUT, "flat_hmm1_expected.mmf"))
        keep = [l for l in log.splitlines() if "average log prob" in l or "floored variance" in l]
        open(os.path.join(OUT, "flat_herest.log"), "w").write("\n".join(keep) + "\n")
        print("\n".join(keep))
    print(sorted(os.listdir(OUT)))
