#!/usr/bin/env python3
"""GPU busy time and the largest idle gaps of a rocprofv3 --kernel-trace CSV (all streams merged): python tools/trace_gaps.py trace.csv [from_frac]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * frac):]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
last_name = iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, last_name, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: last_name = n
busy += cur_e - cur_s
print("span %.2f ms  busy %.2f ms (%.1f %%)  kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(iv)))
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print("  gap %8.1f us  after %-40s before %s" % (g / 1e3, a, b))
