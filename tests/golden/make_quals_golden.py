#!/usr/bin/env python3
"""Golden vectors for the qualifier step of HParm (third differentials, _Z on a table, _N) -- generated with the reference's
own HCopy / HList (oracle/_ref, built from /root/reference by oracle/Makefile) from HTKDemo's MFCC_E file tr1.mfc
(also under tests/golden/demo/train).  HCopy cannot code _N (it is applied when an observation is extracted), so that
one is pinned by HList's print-out of the first 40 observations (3 decimals).

    python tests/golden/make_quals_golden.py
"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref")
SRC = os.path.join(ROOT, "tests", "golden", "demo", "train", "tr1.mfc")
OUT = os.path.join(ROOT, "tests", "golden", "quals")

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        for kind in ("MFCC_E_D_A_T", "MFCC_E_D_A_T_Z", "MFCC_E_D_Z"):
            cfg = os.path.join(d, "cfg")
            open(cfg, "w").write("TARGETKIND = %s\nTHIRDWINDOW = 3\n" % kind)
            subprocess.check_call([os.path.join(REF, "HCopy"), "-C", cfg, SRC, os.path.join(OUT, "tr1_%s.mfc" % kind)])
        for tag, extra in (("V1", "V1COMPAT = T\n"), ("SD", "SIMPLEDIFFS = T\n")):      # the two variants of the difference computation
            cfg = os.path.join(d, "cfg" + tag)
            open(cfg, "w").write("TARGETKIND = MFCC_E_D_A\nDELTAWINDOW = 3\n" + extra)
            subprocess.check_call([os.path.join(REF, "HCopy"), "-C", cfg, SRC, os.path.join(OUT, "tr1_MFCC_E_D_A_%s.mfc" % tag)])
        cfg = os.path.join(d, "cfgN")
        open(cfg, "w").write("TARGETKIND = MFCC_E_D_A_N\n")
        txt = subprocess.check_output([os.path.join(REF, "HList"), "-C", cfg, "-o", "-h", "-e", "39", SRC]).decode()
        txt = txt.replace(SRC, "tr1.mfc")
        open(os.path.join(OUT, "tr1_MFCC_E_D_A_N.hlist"), "w").write(txt)
    print(sorted(os.listdir(OUT)))
