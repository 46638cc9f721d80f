#!/usr/bin/env python3
"""DESIGN.md §2's headline-parity table from the JSON reports the GPU suite wrote (tests/test_gpu_headline_parity.py ->
gpurun_out/headline_parity_<mode>.json, copied to profiles/r06_headline_parity_<mode>.json; tools/r06_parvar.sh for the build variants):
    python tools/parity_table.py [directory = profiles] [prefix = r06_headline_parity_]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
pre = sys.argv[2] if len(sys.argv) > 2 else "r06_headline_parity_"
ROWS = [("exact", "exact"), ("bf16x3fast", "bf16 × 3 + fast log-add (the bench's mode)"), ("fastest", "fp16 × 2 + fast log-add (`--score fastest`)"),
        ("float64_scorer", "a scorer without rounding error (diagnostic build `-DEX_TRUTH`: float64 arithmetic on the fp32 parameters)"),
        ("bf16x3fast_round5_layout", "bf16 × 3, round 5's operand layout (`-DB16_LAYOUT=0`)"), ("bf16x3fast_six_kstep_layout", "bf16 × 3, six-k-step layout (`HTKAMD_BF16_CHUNKED=1`)")]
print("| mode | means > 1e-4 | worst mean | variances > 1e-4 of their own value: two frames or more (worst) + under two frames (worst) | beyond 1e-4 of the second moment | weights worst |")
print("|---|---|---|---|---|---|")
for key, label in ROWS:
    p = os.path.join(d, pre + key + ".json")
    if not os.path.exists(p):
        continue
    r = json.load(open(p))
    r = r.get("model", r)
    print("| %s | %d | %.1e | %d (%.2e) + %d (%.2e) | %d | %.1e |" % (
        label, r["mean"]["n_above_1e4"] + r["mean_low_occ"]["n_above_1e4"], max(r["mean"]["worst_rel"], r["mean_low_occ"]["worst_rel"]),
        r["var"]["n_above_1e4"], r["var"]["worst_rel"], r["var_low_occ"]["n_above_1e4"], r["var_low_occ"]["worst_rel"],
        r["var"]["n_fail"] + r["var_low_occ"]["n_fail"], r["weight"]["worst_rel"]))
