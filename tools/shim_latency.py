#!/usr/bin/env python3
"""Per-file cost of the link-compatible boundary: the reference's unchanged HERest.o over shim/htklib_hfb_shim.c + libhtk_amd.so
(oracle/_ref/HERest_amd) serves FBFile one utterance per call, a batch of one -- prepare, four small launches, results, per file.
Measured by differencing a run over n and a run over 2n files of the headline set (model loading by the reference's own LoadHMMSet
cancels), next to the batched API's rate from bench.py.   python tools/shim_latency.py [n]   (GPU box, needs oracle/_ref)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import c3_herest as c3  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    exe = os.path.join(c3.REF, "HERest_amd")
    if not os.path.exists(exe):
        sys.exit("needs oracle/_ref/HERest_amd (make -C oracle _ref/HERest_amd)")
    s, pk = c3.workload(2 * n)
    with tempfile.TemporaryDirectory(prefix="shimlat_") as d:
        c3.write_files(d, s, pk)
        os.makedirs(os.path.join(d, "out"))
        res = {}
        for mode, env in (("fast", {}), ("exact", {"HTKAMD_SHIM_EXACT": "1"})):
          tt = []
          for k in (n, 2 * n):
            scp = os.path.join(d, "scp%d" % k)
            open(scp, "w").write("\n".join(os.path.join(d, "u%05d.mfc" % u) for u in range(k)) + "\n")
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-S", scp, "-L", d, "-M", os.path.join(d, "out"), "-m", "3", "-v", "0.01",
                                os.path.join(d, "hmmlist")], capture_output=True, text=True, env=dict(os.environ, **env))
            tt.append(time.perf_counter() - t0)
            if r.returncode != 0:
                sys.exit(r.stdout[-500:] + r.stderr[-500:])
          per = (tt[1] - tt[0]) / n
          res[mode] = {"wall_s": [round(x, 3) for x in tt], "ms_per_file": round(per * 1e3, 3), "files_per_s": round(1.0 / per, 1), "fixed_s": round(tt[0] - per * n, 2)}
        # what the device does for ONE such utterance, kernel by kernel (the same library, through the C ABI): the part of a call no host code can remove
        from htk_amd import capi
        import numpy as np
        m = capi.Model(pk)
        X = np.ascontiguousarray(s.feats[0]); dX = capi.DevArray(X)
        fo = np.array([0, X.shape[0]], np.int32); lo = np.array([0, len(s.seqs[0])], np.int32)
        dev = {}
        for mode, sm in (("fast", capi.SCORE_EXACT | capi.SCORE_FASTLADD), ("exact", capi.SCORE_EXACT)):
            fb, acc = capi.ForwardBackward(m), capi.Accs(m)
            ks = []
            for rep in range(5):
                fb.prepare(dX.ptr.value, fo, lo, s.seqs[0].astype(np.int32)); fb.execute(capi.fb_config(scoreMode=sm), acc); fb.results()
                ks.append(fb.kernel_times5())
            k = np.median(np.array(ks), axis=0) * 1e3
            dev[mode] = {"score": round(float(k[0]), 4), "beta": round(float(k[1]), 4), "alpha": round(float(k[2]), 4), "stats": round(float(k[3]), 4), "mix_stats": round(float(k[4]), 4),
                         "sum": round(float(k.sum()), 4)}
        print(json.dumps({"files": [n, 2 * n], "ms_per_file": res["fast"]["ms_per_file"], "files_per_s": res["fast"]["files_per_s"],
                          "default": res["fast"], "HTKAMD_SHIM_EXACT=1": res["exact"], "device_ms_per_file": dev,
                          "note": "FBFile through the HFB shim: one utterance (500 frames, 41 models) per call.  Scores exact in both modes; default = fp32-transcendental log-add in the "
                                  "recursions (tolerance class), HTKAMD_SHIM_EXACT=1 = the table-driven log-add (alpha / beta bit-compatible).  device_ms_per_file: the kernels of one such "
                                  "utterance alone (events around them, 2 x 500 dependent recursion steps) -- the floor of a call"}))
