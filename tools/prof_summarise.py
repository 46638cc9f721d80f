"""Condense the rocprofv3 output of tools/prof_r05.sh into the small files profiles/ keeps:
   <tag>_kernel_stats.csv (copied), <tag>_traffic.json (FETCH_SIZE / WRITE_SIZE KB per launch of every kernel of the timed pass),
   <tag>_sq.json (SQ counters per launch).     python tools/prof_summarise.py gpurun_out/<tag> <tag>"""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join("profiles", "%s_kernel_stats.csv" % tag))


def per_kernel(counter_dir, names):
    acc = {}
    for f in glob.glob(os.path.join(counter_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if not k.startswith("k_"):
                continue
            if r["Counter_Name"] in names:
                d = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [])
                d.append(float(r["Counter_Value"]))
    return acc


traffic = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in per_kernel(os.path.join(out, "pmc_" + c), {c}).items():
        vals = v[c]
        # the bench issues the initial-model pass, the timed iteration and the stand-alone latency passes: report the per-launch mean
        traffic.setdefault(k, {})[c + "_KB"] = sum(vals) / len(vals)
        traffic[k]["launches"] = len(vals)
# the other_paths legs' kernels, profiled on their own (tools/prof_r06.sh step 7, tools/r06_pmc_cmd.sh): k_decode joins the traffic table, both keep their counters
legs = {}
for leg in ("dec", "mfcc"):
    p = os.path.join(out, leg, "summary.json")
    if os.path.exists(p):
        legs[leg] = json.load(open(p))
        d = legs[leg]
        if leg == "dec" and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            traffic["k_decode"] = {"FETCH_SIZE_KB": d["FETCH_SIZE"]["mean_per_launch"], "WRITE_SIZE_KB": d["WRITE_SIZE"]["mean_per_launch"], "launches": d["FETCH_SIZE"]["n"],
                                   "workload": "tools/dec_diag.py 256: 256 x 500 frames on the 6 000-word back-off bigram network (bench.py other_paths.hvite_decoding)"}
if legs:
    json.dump({"_comment": "counters of the other_paths legs' dominant kernels, each leg run on its own (tools/r06_pmc_cmd.sh): per-launch means; k_mfcc_frames: 2 000 x 3 s = 596 000 frames "
                           "(tools/mfcc_bench.py), k_decode: 256 x 500 frames on the bigram network (tools/dec_diag.py)", "legs": legs},
              open(os.path.join("profiles", "%s_legs.json" % tag), "w"), indent=1)
try:
    cfg = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])["config"]
except Exception:  # noqa: BLE001
    cfg = {}
json.dump({"_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0`, counters' KB per launch, "
                       "as counted -- bench.py reports them as counted and, for the scoring kernels, an upper bound with FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 tallies "
                       "16-byte-per-lane streaming reads at half; narrower loads uncalibrated)",
           "workload": {k: cfg.get(k) for k in ("states", "mix", "utts_per_gpu", "frames", "chunks")}, "kernels": traffic},
          open(os.path.join("profiles", "%s_traffic.json" % tag), "w"), indent=1)
sq = {}
names = {"SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"}
for k, v in per_kernel(os.path.join(out, "pmc_SQ"), names).items():
    sq[k] = {c: sum(x) / len(x) for c, x in v.items()}
json.dump({"_comment": "SQ counters per launch (mean over the launches of `bench.py --steps 1 --warmup 0`)", "kernels": sq},
          open(os.path.join("profiles", "%s_sq.json" % tag), "w"), indent=1)
mf = {}
mnames = {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_MFMA"}
for k, v in per_kernel(os.path.join(out, "pmc_MFMA"), mnames).items():
    mf[k] = {c: sum(x) / len(x) for c, x in v.items()}
    if mf[k].get("SQ_BUSY_CU_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD (4 per CU) against SQ_BUSY_CU_CYCLES per CU (MI355X_MICROARCH.md)
        mf[k]["mfma_busy_frac"] = mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * mf[k]["SQ_BUSY_CU_CYCLES"])
if mf:
    json.dump({"_comment": "matrix-pipe counters per launch (mean over the launches of `bench.py --steps 1 --warmup 0`); mfma_busy_frac = "
                           "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)", "kernels": mf},
              open(os.path.join("profiles", "%s_mfma.json" % tag), "w"), indent=1)
tcc = {}
for k, v in per_kernel(os.path.join(out, "pmc_TCC"), {"TCC_HIT_sum", "TCC_MISS_sum"}).items():
    tcc[k] = {c: sum(x) / len(x) for c, x in v.items()}
    h, m_ = tcc[k].get("TCC_HIT_sum", 0.0), tcc[k].get("TCC_MISS_sum", 0.0)
    if h + m_ > 0:
        tcc[k]["l2_hit_rate"] = h / (h + m_)
if tcc:
    json.dump({"_comment": "L2 (TCC) hits and misses per launch, summed over the XCDs (mean over the launches of `bench.py --steps 1 --warmup 0`)", "kernels": tcc},
              open(os.path.join("profiles", "%s_tcc.json" % tag), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:1500])
