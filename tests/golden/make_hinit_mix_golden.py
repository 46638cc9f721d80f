#!/usr/bin/env python3
"""Known answer for HInit with mixtures (UniformSegment -> FlatCluster HTrain.c:763, FindBestMixes HInit.c:738, UpdateCounts :878):
the demo's prototypes rewritten with 2 / 3 / 2 mixture components in states 2 / 3 / 4, then the reference's HInit (oracle/_ref) as
HTKDemo calls it:   HInit -i 10 -L labels -l X -o X -C hinit.conf -D -M out -T 1 proto_mix/X train/*.mfc
    python tests/golden/make_hinit_mix_golden.py  -> tests/golden/demo/hinit_mix/{proto/X, hmm0/X, hinit.log}"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEMO = os.path.join(ROOT, "tests", "golden", "demo")
MIXES = {2: 2, 3: 3, 4: 2}


def mix_proto(text):
    out, lines, i = [], text.splitlines(), 0
    while i < len(lines):
        m = re.match(r"\s*<State> (\d+) <NumMixes> 1", lines[i])
        if not m:
            out.append(lines[i]); i += 1
            continue
        st = int(m.group(1)); M = MIXES[st]
        out.append("  <State> %d <NumMixes> %d " % (st, M))
        out.append(lines[i + 1])                          # <Stream> 1
        body = lines[i + 3:i + 7]                         # <Mean> n / values / <Variance> n / values
        for k in range(M):
            out.append("  <Mixture> %d %.4f" % (k + 1, 1.0 / M))
            out += body
        i += 7
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    out = os.path.join(DEMO, "hinit_mix")
    os.makedirs(os.path.join(out, "proto"), exist_ok=True); os.makedirs(os.path.join(out, "hmm0"), exist_ok=True)
    keep = []
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "hinit.conf")
        open(cfg, "w").write("TARGETKIND = MFCC_E_D\nSAVEGLOBOPTS = TRUE\nKEEPDISTINCT=F\n")
        for name in "SCVNL":
            open(os.path.join(out, "proto", name), "w").write(mix_proto(open(os.path.join(DEMO, "proto", name)).read()))
            log = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "HInit"), "-i", "10", "-L", os.path.join(DEMO, "labels"), "-l", name, "-o", name, "-C", cfg,
                                  "-D", "-M", os.path.join(out, "hmm0"), "-T", "1", os.path.join(out, "proto", name)] + sorted(glob.glob(os.path.join(DEMO, "train", "tr*.mfc"))),
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
            for l in log.splitlines():
                if re.search(r"Estimation (converged|aborted)|Iteration \d+: Average LogP|ERROR|WARNING", l):
                    keep.append("HInit %s: %s" % (name, l.strip()))
    open(os.path.join(out, "hinit.log"), "w").write("\n".join(keep) + "\n")
    print("\n".join(keep))
