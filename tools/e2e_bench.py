#!/usr/bin/env python3
"""End to end, files -> models: one HERest iteration by this repository's `tools/bin/herest` from files on disk (text MMF, HTK parameter
files listed in a script file, label files), wall clock from process start to the new MMF on disk -- everything bench.py leaves out
(reading and parsing the model set, reading the parameter files, host->device copies, host preparation, MMF writing) -- next to the
reference's HERest on ONE core over a sample of the same files (its whole run, model loading included and reported separately).

    python tools/e2e_bench.py [--states 5000 --mix 16 --utts 1250 --frames 500 --ref-utts 40]
Prints one JSON line."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from htk_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--states", type=int, default=5000); ap.add_argument("--mix", type=int, default=16); ap.add_argument("--phones", type=int, default=6000)
    ap.add_argument("--utts", type=int, default=1250); ap.add_argument("--frames", type=int, default=500); ap.add_argument("--ref-utts", type=int, default=40)
    ap.add_argument("--score", default="fastest")
    ap.add_argument("--iterations", type=int, default=4, help="also time K iterations in one process (tools/bin/herest --iterations K) against K chained processes")
    a = ap.parse_args()
    s = synth.generate_fast(a.states, a.mix, a.phones, a.utts, a.frames, seed=1000, model_seed=3)
    pk = s.packed()
    H = int(pk["numPhys"])
    names = ["p%d" % i for i in range(H)]
    d = tempfile.mkdtemp(prefix="e2e_")
    out = {}
    try:
        t0 = time.perf_counter()
        synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
        open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
        with open(os.path.join(d, "labels.mlf"), "w") as f:
            f.write("#!MLF!#\n")
            for u in range(a.utts):
                f.write('"*/u%05d.lab"\n' % u + "\n".join(names[int(h)] for h in s.seqs[u]) + "\n.\n")
        with open(os.path.join(d, "scp"), "w") as f:
            for u in range(a.utts):
                fn = os.path.join(d, "u%05d.mfc" % u)
                synth.write_htk_param(fn, s.feats[u], kind=9)
                f.write(fn + "\n")
        open(os.path.join(d, "config"), "w").close()
        os.makedirs(os.path.join(d, "next")); os.makedirs(os.path.join(d, "refout"))
        out["corpus"] = {"mmf_MB": round(os.path.getsize(os.path.join(d, "MMF")) / 1e6, 1), "parm_MB": round(a.utts * a.frames * 39 * 4 / 1e6, 1),
                         "write_s": round(time.perf_counter() - t0, 2)}
        exe = os.path.join(ROOT, "tools", "bin", "herest")
        cmd = [exe, "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-S", os.path.join(d, "scp"), "-I", os.path.join(d, "labels.mlf"),
               "-M", os.path.join(d, "next"), "-v", "0.01", "-T", "16384", "--score", a.score, os.path.join(d, "hmmlist")]
        runs = []
        for k in range(3):                                            # first run also pays the page-cache fill and the GPU context; all are reported
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                print(r.stdout[-800:], r.stderr[-800:]); sys.exit(1)
            runs.append(round(dt, 3))
        out["herest_amd_wall_s"] = runs
        out["herest_amd_utts_per_s"] = round(a.utts / min(runs), 1)
        m = [l for l in r.stdout.splitlines() if "average log prob" in l]
        out["log"] = m[-1].strip() if m else ""
        out["phases_s"] = {l.split()[1] if False else " ".join(l.split()[1:-2]): float(l.split()[-2]) for l in r.stdout.splitlines() if l.startswith("Timing:")}
        # the same with HTK's binary model files (HERest -B): what remains is the data path
        os.makedirs(os.path.join(d, "bin0")); os.makedirs(os.path.join(d, "bin1"))
        cb = [c for c in cmd]
        cb[cb.index("-M") + 1] = os.path.join(d, "bin0")
        r = subprocess.run(cb[:-1] + ["-B", cb[-1]], capture_output=True, text=True)
        if r.returncode == 0:
            cb2 = [c for c in cmd]
            cb2[cb2.index("-H") + 1] = os.path.join(d, "bin0", "MMF"); cb2[cb2.index("-M") + 1] = os.path.join(d, "bin1")
            runs = []
            for k in range(3):
                t0 = time.perf_counter()
                r = subprocess.run(cb2[:-1] + ["-B", cb2[-1]], capture_output=True, text=True)
                runs.append(round(time.perf_counter() - t0, 3))
            if r.returncode == 0:
                out["binary_mmf"] = {"wall_s": runs, "mmf_MB": round(os.path.getsize(os.path.join(d, "bin0", "MMF")) / 1e6, 1),
                                     "phases_s": {" ".join(l.split()[1:-2]): float(l.split()[-2]) for l in r.stdout.splitlines() if l.startswith("Timing:")}}
        # K Baum-Welch iterations: one process with --iterations K (everything stays on the device, one set written) against K processes
        # chained through binary model files (HTK's recipe: one HERest run per iteration)
        K = a.iterations
        if K > 1:
            os.makedirs(os.path.join(d, "itK"))
            ck = [c for c in cmd]
            ck[ck.index("-M") + 1] = os.path.join(d, "itK")
            runs = []
            for k in range(2):
                t0 = time.perf_counter()
                r = subprocess.run(ck[:-1] + ["-B", "--iterations", str(K), ck[-1]], capture_output=True, text=True)
                runs.append(round(time.perf_counter() - t0, 3))
            if r.returncode == 0:
                t0 = time.perf_counter()
                srcm = os.path.join(d, "MMF")
                for k in range(K):
                    os.makedirs(os.path.join(d, "ch%d" % k))
                    cc = [c for c in cmd]
                    cc[cc.index("-H") + 1] = srcm; cc[cc.index("-M") + 1] = os.path.join(d, "ch%d" % k)
                    rc = subprocess.run(cc[:-1] + ["-B", cc[-1]], capture_output=True, text=True)
                    srcm = os.path.join(d, "ch%d" % k, "MMF")
                chain = round(time.perf_counter() - t0, 3)
                out["iterations"] = {"K": K, "one_process_wall_s": runs, "chain_of_K_processes_wall_s": chain,
                                     "phases_s": {" ".join(l.split()[1:-2]): float(l.split()[-2]) for l in r.stdout.splitlines() if l.startswith("Timing:")},
                                     "log": [l.strip() for l in r.stdout.splitlines() if "average log prob" in l]}
        ref = os.path.join(ROOT, "oracle", "_ref", "HERest")
        if os.path.exists(ref) and a.ref_utts > 0:
            tt = []
            for n in (a.ref_utts, 2 * a.ref_utts):
                open(os.path.join(d, "scp_ref"), "w").write("".join(os.path.join(d, "u%05d.mfc\n" % u) for u in range(n)))
                t0 = time.perf_counter()
                r = subprocess.run([ref, "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-S", os.path.join(d, "scp_ref"), "-I", os.path.join(d, "labels.mlf"),
                                    "-M", os.path.join(d, "refout"), "-v", "0.01", os.path.join(d, "hmmlist")], capture_output=True, text=True)
                tt.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    print(r.stdout[-500:], r.stderr[-500:]); break
            if len(tt) == 2:
                per = (tt[1] - tt[0]) / a.ref_utts
                out["reference_one_core"] = {"run_s": [round(x, 2) for x in tt], "s_per_utt": round(per, 4), "fixed_s(load+update+save)": round(tt[0] - per * a.ref_utts, 2),
                                             "projected_s_for_all": round(tt[0] - per * a.ref_utts + per * a.utts, 1)}
        print(json.dumps(out))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
