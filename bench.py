#!/usr/bin/env python3
"""bench.py -- HERest hot path (GMM scoring -> forward-backward -> statistics -> accumulator all-reduce)
on the configuration BASELINE.json's metric is quoted on: 5k tied states x 16 mixtures, 39-dim features,
500-frame synthetic utterances; config[2] sharded 8-way = 1250 utterances per GPU (weak scaling: every rank
processes its own 1250-utterance shard, the accumulator vector is summed with RCCL once per pass).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one EM ITERATION of embedded Baum-Welch over the rank's shard with the features already resident in HBM:
ZeroAccs, CreateInsts/beam taper on the host, K1 scoring, K2 beta, K3 alpha + occupation/transition statistics,
K4 mixture statistics, all-reduce(sum) of the fp64 accumulator vector, UpdateModels + table refresh on the device,
per-utterance results read back.  Step k+1 runs on the model step k wrote: nothing overlaps between steps.
PyTorch is used for device memory, the stream, the barrier and torch.distributed (backend nccl = RCCL);
everything numeric is the HIP library behind include/htk_amd.h.

One JSON line is printed by rank 0 (see README / DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FRAME_STATE = lambda M, D: M * (4 * D + 8)
FP32_PEAK_TFLOPS = 157.3                                 # MI355X dense FP32 (vector = matrix), MI355X_MICROARCH.md
F16_PEAK_TFLOPS = 2500.0                                 # dense f16 / bf16 matrix peak (no sparsity), MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                                    # HBM3E, MI355X_MICROARCH.md
PROFILE_TRAFFIC = "r06_traffic.json"                     # profiles/: PMC passes of this round's kernels (tools/prof_r06.sh)
PROFILE_LEGS = "r06_legs.json"                           # ... and of the other_paths legs' kernels, each leg profiled on its own


def cpu_baseline(s, pk, budget_s: float):
    """Oracle (CPU restatement of the reference, oracle/htk_oracle.c) timed on one host core over a bounded
    sample of the same workload.  Reported, never the thing measured as `value`."""
    from oracle import pyoracle as po
    om = po.Model(pk)
    acc = po.Accs(om)
    cfg = po.fb_cfg()
    n, t0, prs = 0, time.perf_counter(), []
    n_timed, dt = 0, 0.0
    while n < len(s.feats):                                  # the rate is taken over the first budget_s seconds; the CHECK wants the whole shard
        rc, pr, _ = po.fb_utt(om, cfg, s.feats[n], s.seqs[n], acc)   # (a sample whose size follows the host's speed made the check's worst entry a lottery)
        prs.append(pr)
        n += 1
        if n_timed == 0 and time.perf_counter() - t0 > budget_s:
            n_timed, dt = n, time.perf_counter() - t0
    if n_timed == 0:
        n_timed, dt = n, time.perf_counter() - t0
    return n, dt, np.array(prs), acc, n_timed


def cpu_baseline_reference(s, pk, n_utt: int, workers: int = 1):
    """The reference's own HERest (oracle/_ref/HERest, built from /root/reference by oracle/Makefile and shipped with the tree) as
    `workers` parallel processes `HERest -p k` (its parallel mode: every process accumulates its own shard and dumps HERk.acc,
    HERest.c:543-550), each over n_utt utterances of the same shard.  Model loading (a 25 MB text MMF per process) is taken out by
    differencing a round over n and a round over 2n utterances per process.  Returns (utterances, seconds, workers) or None."""
    import subprocess
    import tempfile
    from htk_amd import synth
    exe = os.path.join(ROOT, "oracle", "_ref", "HERest")
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="herest_ref_")
    try:
        H = int(pk["numPhys"])
        names = ["p%d" % i for i in range(H)]
        synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
        with open(os.path.join(d, "hmmlist"), "w") as f:
            f.write("\n".join(names) + "\n")
        os.makedirs(os.path.join(d, "out"))
        pool = min(len(s.feats), max(2 * n_utt, 240))              # files on disk; the workers read overlapping windows of them
        for u in range(pool):
            synth.write_htk_param(os.path.join(d, "u%05d.mfc" % u), s.feats[u], kind=9)
            with open(os.path.join(d, "u%05d.lab" % u), "w") as f:
                f.write("\n".join(names[int(h)] for h in s.seqs[u]) + "\n")
        open(os.path.join(d, "config"), "w").close()
        times = []
        # (a short untimed round first: the first round of processes on a box -- cold binaries, clocks, the freshly written files -- has been seen to take long
        #  enough to halve the difference below, i.e. to double the reported rate)
        for n in (max(1, n_utt // 8), n_utt, 2 * n_utt):
            procs = []
            for k in range(workers):
                scp = os.path.join(d, "scp%d" % k)
                with open(scp, "w") as f:
                    f.write("\n".join(os.path.join(d, "u%05d.mfc" % ((k * 7 + i) % pool)) for i in range(n)) + "\n")
            t0 = time.perf_counter()
            for k in range(workers):
                procs.append(subprocess.Popen([exe, "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-S", os.path.join(d, "scp%d" % k),
                                               "-L", d, "-M", os.path.join(d, "out"), "-p", str(k + 1), os.path.join(d, "hmmlist")],
                                              stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT))
            if any(p.wait() != 0 for p in procs):
                return None
            times.append(time.perf_counter() - t0)
        dt = times[2] - times[1]
        return (workers * n_utt, dt, workers) if dt > 0 else None
    except OSError:
        return None
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def other_paths(s, pk, dX, frame_off_all, n_align=256, n_decode=256, cpu_utts=2):       # the decoder runs a workgroup per utterance: a machine of 256 CUs wants that many
    """The path's other two consumers, at the same set, outside the timed region (reported, never `value`):
    HVite -a forced alignment (K5) of the shard's first utterances -- checked against the oracle's token likelihood on one of them --
    and HVite -w decoding (K7) over a word loop of the set's 6 000 one-model words with -t 250 (BASELINE config[3])."""
    import tempfile
    from htk_amd import capi, synth
    from oracle import pyoracle as po
    out = {}
    model = capi.Model(pk)
    n_align = min(n_align, len(s.feats))
    fo = frame_off_all[:n_align + 1].astype(np.int32)
    lo = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs[:n_align]])]).astype(np.int32)
    lb = np.concatenate(s.seqs[:n_align]).astype(np.int32)
    vit = capi.Viterbi(model)
    got = vit.align(dX.data_ptr(), fo, lo, lb)
    dt = 1e30
    for _ in range(5):                                      # the fastest of five (see the decoding leg)
        t0 = time.perf_counter()
        got = vit.align(dX.data_ptr(), fo, lo, lb)
        dt = min(dt, time.perf_counter() - t0)
    ref = po.viterbi_align(po.Model(pk), s.feats[0], s.seqs[0])
    assert ref is not None and got[0]["status"] == 1 and got[0]["total"] == ref["total"], "bench: alignment differs from the oracle"
    out["hvite_alignment"] = {"utterances": n_align, "ms": dt * 1e3, "utterances_per_sec": n_align / dt, "frames_per_sec": float(fo[-1]) / dt,
                              "oracle_check": "token likelihood of utterance 0 bit-identical", "arithmetic": "exact (K1 + K5)"}
    # decoding: the network builder wants names and topologies only -- a stand-in model file with the set's 6 000 models over one state
    V = int(pk["numPhys"])
    d = tempfile.mkdtemp(prefix="bench_dec_")
    try:
        names = ["p%d" % i for i in range(V)]
        Dv = int(pk["vecSize"])
        with open(os.path.join(d, "MMF"), "w") as f:
            f.write("~o\n<STREAMINFO> 1 %d\n<VECSIZE> %d<NULLD><USER><DIAGC>\n" % (Dv, Dv))
            f.write('~t "T0"\n<TRANSP> 5\n 0 1 0 0 0\n 0 0.6 0.4 0 0\n 0 0 0.6 0.4 0\n 0 0 0 0.7 0.3\n 0 0 0 0 0\n')
            f.write('~s "S0"\n<MEAN> %d\n%s\n<VARIANCE> %d\n%s\n' % (Dv, " 0" * Dv, Dv, " 1" * Dv))
            for n_ in names:
                f.write('~h "%s"\n<BEGINHMM>\n<NUMSTATES> 5\n<STATE> 2\n~s "S0"\n<STATE> 3\n~s "S0"\n<STATE> 4\n~s "S0"\n~t "T0"\n<ENDHMM>\n' % n_)
        open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
        open(os.path.join(d, "dict"), "w").write("".join("%s %s\n" % (n_, n_) for n_ in names))
        # BASELINE config[3] as it is worded: a back-off BIGRAM network over the 6 000 words (ProcessBoBiGram's shape, HBuild.c:368-461: five
        # explicit successors per word + the back-off node that reaches every word), HVite -t 250 -s 5 -p -10
        n_arcs = synth.write_bigram_slf(os.path.join(d, "net.slf"), names)
        mmf = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
        net = capi.Net(os.path.join(d, "net.slf"), os.path.join(d, "dict"), mmf)
        dec = capi.Decoder(model, net, lmScale=5.0)
        feats = s.feats[:n_decode]
        kw = dict(genBeam=250.0, lmScale=5.0, wordPen=-10.0)
        res = dec.run(feats, **kw)
        dt = 1e30
        for _ in range(3):                                  # the fastest of three: the leg runs after the CPU baseline, on a device whose clocks have gone idle
            t0 = time.perf_counter()
            res = dec.run(feats, **kw)
            dt = min(dt, time.perf_counter() - t0)
        # ... and the same decoding with the bf16 x 3 matrix-core scores (tolerance class; HVite's output is held to the exact scores above): how many
        # utterances come out with the same words and boundaries, what the leg then takes
        tol_side = None
        try:
            res_t = dec.run(feats, scoreMode=capi.SCORE_BF16, **kw)
            dt_t = 1e30
            for _ in range(2):
                t0 = time.perf_counter()
                res_t = dec.run(feats, scoreMode=capi.SCORE_BF16, **kw)
                dt_t = min(dt_t, time.perf_counter() - t0)
            sc_t, tok_t = dec.last_times()
            same = sum(1 for (wa, _), (wb, _) in zip(res, res_t) if (wa is None) == (wb is None) and (wa is None or [x[:3] for x in wa] == [x[:3] for x in wb]))
            tol_side = {"arithmetic": "bf16 x 3 matrix-core scores (k_score_bf16w) + K7", "ms": dt_t * 1e3, "utterances_per_sec": len(feats) / dt_t, "score_ms": sc_t, "token_ms": tok_t,
                        "utterances_with_the_exact_runs_words_and_boundaries": "%d/%d" % (same, len(feats))}
            res = dec.run(feats, **kw)                      # (the exact run again: what follows reads its times and counts)
        except capi.HtkAmdError as e:
            tol_side = {"error": str(e)[:200]}
        hit = tot = 0
        for (w, _), q in zip(res, s.seqs[:n_decode]):
            rec = [] if w is None else [net.out_syms[p_] for p_, _, _, _ in w]
            ref_ = ["p%d" % k for k in q]
            tot += len(ref_); hit += sum(1 for x, y in zip(rec, ref_) if x == y) if len(rec) == len(ref_) else 0
        frames_ = sum(f.shape[0] for f in feats)
        sc_ms, tok_ms = dec.last_times()                                         # device events of the last run
        S_ = int(pk["numStates"]); M_ = int(np.max(np.diff(pk["stateCompOff"])))
        # the token kernel against HBM: per (utterance, frame) the score column (4 B per tied state) and a Path record (16 B) per word end --
        # what is left in memory with the models' tokens in registers and the word ends' exit tokens in LDS.  The kernel
        # is nowhere near that bound: it is a chain of dependent steps, bound by the instructions it issues (tools/dec_diag.py, -DDEC_CLK)
        dec_bytes = frames_ * (4.0 * S_ + 16.0 * V)
        ach = dec_bytes / (tok_ms * 1e-3) / 1e9 if tok_ms > 0 else 0.0
        # the dense scores against the fp32 vector peak the exact kernel is bounded by (no FMA: half of it at best)
        sc_flop = float(frames_) * S_ * FLOP_PER_FRAME_STATE(M_, Dv)
        sc_ach = sc_flop / (sc_ms * 1e-3) / 1e12 if sc_ms > 0 else 0.0
        out["hvite_decoding"] = {"utterances": len(feats), "ms": dt * 1e3, "utterances_per_sec": len(feats) / dt, "frames_per_sec": frames_ / dt,
                                 "network": "back-off bigram over %d one-model words (%d arcs, fan-in %d at the back-off node), -t 250 -s 5 -p -10 (BASELINE config[3])" % (V, n_arcs, V),
                                 "words_correct": "%d/%d" % (hit, tot),
                                 "arithmetic": "exact (K1 dense + K7)", "exact_order_utterances": dec.last_tied(), "tolerance_class_scores": tol_side,
                                 "model_instance_steps": dict(zip(("live", "dead"), dec.last_live())),
                                 "roofline": {"kernel": "k_decode (token passing)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                              "ms": tok_ms, "bytes_per_unit": 4.0 * S_ + 16.0 * V, "unit_is": "(utterance, frame)", "units_per_launch": frames_,
                                              "traffic": None, "note": "traffic: counter bytes per launch from profiles/ (k_decode FETCH_SIZE + WRITE_SIZE), filled in when the committed PMC passes are of this workload"},
                                 "score_roofline": {"kernel": "k_score_exact (every tied state, every frame) + k_score_transpose", "bound": "mfma", "achieved": sc_ach,
                                                    "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": sc_ach / FP32_PEAK_TFLOPS, "ms": sc_ms,
                                                    "note": "packed fp32 VALU, the reference's four roundings per dimension: half the fp32 peak at best"}}
        # the reference's HVite beside it (oracle/_ref/HVite, one core): the same network, dictionary and utterances from files; the 25 MB model
        # set's loading and the network expansion cancel in the difference of a run over n and a run over 2 n utterances
        exe = os.path.join(ROOT, "oracle", "_ref", "HVite")
        if cpu_utts > 0 and os.path.exists(exe):
            import subprocess
            synth.write_mmf_packed(os.path.join(d, "MMFfull"), pk, names)
            open(os.path.join(d, "config"), "w").close()
            for u in range(2 * cpu_utts):
                synth.write_htk_param(os.path.join(d, "u%05d.mfc" % u), s.feats[u], kind=9)
            tt = []
            for n_ in (cpu_utts, 2 * cpu_utts):
                scp = os.path.join(d, "scp%d" % n_)
                open(scp, "w").write("\n".join(os.path.join(d, "u%05d.mfc" % u) for u in range(n_)) + "\n")
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMFfull"), "-S", scp, "-i", os.path.join(d, "rec%d.mlf" % n_), "-w", os.path.join(d, "net.slf"),
                                    "-t", "250.0", "-s", "5.0", "-p", "-10.0", os.path.join(d, "dict"), os.path.join(d, "hmmlist")], stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
                tt.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    tt = None
                    break
            if tt and tt[1] > tt[0]:
                per = (tt[1] - tt[0]) / cpu_utts
                out["hvite_decoding"]["cpu_baseline"] = {"value": 1.0 / per, "unit": "utterances/s", "cores": 1, "kind": "reference",
                                                         "sample": "the reference's HVite (oracle/_ref) on %d and %d of the same utterances, same network; per-utterance time from the difference (%.2f s), fixed cost %.1f s" % (cpu_utts, 2 * cpu_utts, per, tt[0] - per * cpu_utts),
                                                         "frames_per_sec": 500.0 / per}
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    return out


def mfcc_leg(pk, n_utt=2000, cpu_files=40):
    """BASELINE config[4]: raw 16 kHz waveforms -> on-device HSigP / HParm MFCC_0_D_A (512-point FFT, 26 channels, 12 cepstra + C0, deltas and
    accelerations: 39 columns) -> GMM state scores of the frames, everything resident on the device; the reference's HCopy beside it on one
    host core (files on disk, the difference of a run over n and one over 2 n files).  Reported under other_paths, never `value`."""
    import ctypes as C
    import tempfile
    import torch
    from htk_amd import capi, synth
    rng = np.random.default_rng(7)
    n = 48000                                               # 3 s
    t = np.arange(n) / 16000.0
    base = (3000 * np.sin(2 * np.pi * 440 * t) + 2000 * np.sin(2 * np.pi * 1800 * t)).astype(np.float32)
    distinct = [(base + rng.normal(0, 500, n)).astype(np.int16) for _ in range(8)]
    waves = [distinct[i % 8] for i in range(n_utt)]
    cfg = capi.mfcc_config("MFCC_0_D_A")
    fe = capi.Mfcc(cfg)
    sampOff = np.concatenate([[0], np.cumsum([len(w) for w in waves])]).astype(np.int32)
    allw = np.concatenate(waves)
    per_utt = capi.lib().htkamd_mfcc_num_frames(C.byref(cfg), C.c_int(n))
    frames = per_utt * n_utt
    dW = capi.DevArray(allw)
    dO = capi.DevArray(nbytes=4 * frames * fe.cols)
    frameOff = np.zeros(n_utt + 1, np.int32)
    dt = 1e30
    for _ in range(4):
        t0 = time.perf_counter()
        capi.check(capi.lib().htkamd_mfcc_compute(fe.h, dW.ptr, sampOff.ctypes.data_as(C.c_void_p), C.c_int(n_utt), frameOff.ctypes.data_as(C.c_void_p), dO.ptr, None), "mfcc_compute")
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    out = {"utterances": n_utt, "seconds_of_audio": 3.0 * n_utt, "frames": int(frames), "ms": dt * 1e3, "frames_per_sec": frames / dt, "pcm_GB_per_sec": allw.nbytes / dt / 1e9,
           "x_real_time": 3.0 * n_utt / dt, "config": "16 kHz, 25 ms / 10 ms, 512-point FFT, 26 channels, 12 cepstra + C0, _D_A (39 columns)",
           # 800 bytes in / 52 out per static frame; here also the 39-column rows of the qualifier kernels (written once, read by the regressions)
           "hbm_GB_per_sec": (allw.nbytes + 4.0 * frames * fe.cols) / dt / 1e9}
    # What bounds the leg: not memory (800 bytes in, 156 out per frame: 0.01 of HBM) -- the frame kernel issues ~1 000 vector instructions and keeps the
    # LDS pipe busy for ~600 cycles per frame, bank conflicts included.  Both from the committed counters of this workload (profiles/r06_legs.json:
    # SQ_INSTS_VALU, SQ_ACTIVE_INST_LDS + SQ_LDS_BANK_CONFLICT per launch of k_mfcc_frames), priced against the call's time measured here:
    # vector issue = 1 024 SIMDs x one wave-instruction per 4 cycles at 2.4 GHz; LDS = one pipe per CU.
    try:
        lg = json.load(open(os.path.join(ROOT, "profiles", PROFILE_LEGS)))["legs"]["mfcc"]
        valu, lds_cyc = lg["SQ_INSTS_VALU"]["mean_per_launch"], lg["SQ_ACTIVE_INST_LDS"]["mean_per_launch"] + lg["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"]
        peak_issue = 1024 * 2.4e9 / 4.0
        scale = frames / 596000.0                           # the counters are of the 596 000-frame run
        out["roofline"] = {"kernel": "k_mfcc_frames (two frames per wavefront; + the regression kernels in the host-timed call)", "bound": "vector issue",
                           "achieved": valu * scale / dt / 1e9, "peak": peak_issue / 1e9, "unit": "G wave-instructions/s", "frac": valu * scale / dt / peak_issue,
                           "vector_instructions_per_frame": valu / 596000.0,
                           "lds": {"busy_cycles_per_frame": lds_cyc / 596000.0, "bank_conflict_share": lg["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / lds_cyc,
                                   "frac_of_lds_pipe": lds_cyc * scale / dt / (256 * 2.4e9)},
                           "kernel_us_in_the_profiled_run": lg.get("duration_us", {}).get("median"),
                           "note": "counters: profiles/%s (rocprofv3 --pmc, tools/r06_pmc_cmd.sh); time: this run's call.  HBM: %.3f of peak -- not the bound" % (
                               PROFILE_LEGS, (allw.nbytes + 4.0 * frames * fe.cols) / dt / 1e9 / HBM_PEAK_GBS)}
    except (OSError, KeyError, ValueError):
        out["roofline"] = {"kernel": "k_mfcc_frames", "bound": "vector issue", "achieved": None, "peak": 1024 * 2.4 / 4.0, "unit": "G wave-instructions/s", "frac": None,
                           "note": "no committed counters of this workload (profiles/%s)" % PROFILE_LEGS}
    # ... -> GMM scoring: the frames against the first 1 024 tied states of the headline set's shape (D = 39), bf16 x 3 matrix-core scores, on the device
    if int(pk["vecSize"]) == fe.cols:
        model = capi.Model(pk)
        ns = min(1024, int(pk["numStates"]))
        dS = capi.DevArray(np.arange(ns, dtype=np.int32))
        dY = capi.DevArray(nbytes=4 * frames * ns)
        ds = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            capi.check(capi.lib().htkamd_outp_block_mode(model.h, dO.ptr, C.c_int(frames), dS.ptr, C.c_int(ns), dY.ptr, C.c_int(frames), C.c_int(capi.SCORE_BF16), None), "outp_block")
            torch.cuda.synchronize()
            ds = min(ds, time.perf_counter() - t0)
        M_ = int(np.max(np.diff(pk["stateCompOff"])))
        out["scoring"] = {"states": ns, "ms": ds * 1e3, "frame_state_loglik_per_sec": frames * ns / ds,
                          "algorithmic_TFLOPs": frames * ns * FLOP_PER_FRAME_STATE(M_, fe.cols) / ds / 1e12,
                          "wav_to_scores_ms": (dt + ds) * 1e3}
    exe = os.path.join(ROOT, "oracle", "_ref", "HCopy")
    if cpu_files > 0 and os.path.exists(exe):
        import subprocess
        d = tempfile.mkdtemp(prefix="bench_mfcc_")
        try:
            open(os.path.join(d, "config"), "w").write("SOURCEFORMAT = WAV\nSOURCERATE = 625\nTARGETKIND = MFCC_0_D_A\nTARGETRATE = 100000\nWINDOWSIZE = 250000\n"
                                                       "NUMCHANS = 26\nNUMCEPS = 12\nCEPLIFTER = 22\nPREEMCOEF = 0.97\nUSEHAMMING = T\nENORMALISE = T\n")
            for u in range(2 * cpu_files):
                synth.write_wav(os.path.join(d, "w%04d.wav" % u), distinct[u % 8])
            tt = []
            for n_ in (cpu_files, 2 * cpu_files):
                scp = os.path.join(d, "scp%d" % n_)
                open(scp, "w").write("".join("%s %s\n" % (os.path.join(d, "w%04d.wav" % u), os.path.join(d, "w%04d.mfc" % u)) for u in range(n_)))
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-C", os.path.join(d, "config"), "-S", scp], stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
                tt.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    tt = None
                    break
            if tt and tt[1] > tt[0]:
                per = (tt[1] - tt[0]) / cpu_files
                out["cpu_baseline"] = {"value": per_utt / per, "unit": "frames/s", "cores": 1, "kind": "reference",
                                       "sample": "the reference's HCopy (oracle/_ref) on %d and %d of the same 3 s waveforms as WAV files; per-file time from the difference" % (cpu_files, 2 * cpu_files)}
        finally:
            import shutil
            shutil.rmtree(d, ignore_errors=True)
    return out


def subprocess_leg(argv, timeout_s, pick):
    """A leg that is a run of its own (another process, after this one's timed region): its last JSON line, reduced by `pick`."""
    import subprocess
    try:
        r = subprocess.run([sys.executable] + argv, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        if r.returncode != 0:
            return {"error": (r.stdout[-200:] + r.stderr[-300:])}
        return pick(json.loads(r.stdout.strip().splitlines()[-1]))
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def cgroup_cpu_quota():
    """CPUs the container's cgroup grants (cpu.max of cgroup v2, cfs quota of v1), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def host_cores() -> int:
    """Cores this process may use: the affinity mask, capped at the physical core count and at the cgroup's CPU quota (the reference is
    run as that many -p workers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        import psutil
        ph = psutil.cpu_count(logical=False)
        if ph:
            n = min(n, ph)
    except Exception:  # noqa: BLE001
        pass
    q = cgroup_cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q)))
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # (0.22 s timed at 2.2 ms per iteration: r04's 50 x 3.3 ms was called short)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--states", type=int, default=5000)
    ap.add_argument("--mix", type=int, default=16)
    ap.add_argument("--phones", type=int, default=6000)
    ap.add_argument("--utts", type=int, default=1250, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --utts utterances on every GPU (the contract's default); strong: --total-utts utterances split over the GPUs")
    ap.add_argument("--total-utts", type=int, default=10000, help="--scaling strong: utterances of the whole job (BASELINE config[2]: 10k)")
    ap.add_argument("--ragged", type=int, default=0, help="1: utterance lengths uniform in [frames/2, frames] (a look at mixed batches; the headline run uses 0)")
    ap.add_argument("--score", choices=["exact", "mfma", "fast", "bf16", "fastest"], default="bf16",
                    help="arithmetic: exact = bit-identical to the reference; mfma = fp32 matrix-core scores; fast = mfma + fp32-transcendental LAdd in the recursions; "
                         "bf16 (default: every asserted parity figure at half the 1e-4 bar or better) = bf16 x 3 matrix-core scores + that LAdd; "
                         "fastest = fp16 x 2 matrix-core scores + that LAdd (the parity figures reach 0.6 - 1.0 of the bar depending on the sample)")
    ap.add_argument("--also-fastest", type=int, default=1, help="1: after the measurement (N = 1, --score bf16) the same timed iterations once more with the fp16 x 2 scores, reported as `fastest_mode` beside the line, never as `value`")
    ap.add_argument("--two-streams", type=int, default=1, help="run the chunks of an iteration on two alternating streams (1) or on one stream (0)")
    ap.add_argument("--chunks", type=int, default=1, help="sub-batches an iteration's shard is cut into (alternating over two streams, one accumulator vector)")
    ap.add_argument("--min-var", type=float, default=0.01, help="HERest -v: variance floor of the update (the shard has ~8 frames per Gaussian)")
    ap.add_argument("--prewarm-seconds", type=float, default=2.0, help="wall time of untimed EM iterations before the warm-up steps (a freshly started box is slow for its first seconds)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="processes of the reference CPU baseline (0 = one per physical host core)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="1: also time forced alignment and network decoding at the same set after the timed region (N = 1 only; reported as other_paths)")
    ap.add_argument("--exchange-slices", type=int, default=4,
                    help="N > 1 only: the accumulator exchange of an iteration in this many parts by tied state -- a part travels (on a stream of its own) while the next "
                         "range of states' mixture statistics is still being summed (htkamd_fb_execute_begin / _mix); 1: one all-reduce behind the pass")
    ap.add_argument("--wire", choices=["f32", "f64"], default="f32",
                    help="N > 1: the accumulator statistics on the wire as fp32 (every rank rounds its fp64 partial sums once; counters stay fp64) or fp64")
    ap.add_argument("--dump-model", default=None, help="rank 0 writes the model of the last iteration (npz: mean, var, compWeight, transP) -- the multi-GPU tests compare it over rank counts")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here, as CHILDREN under torch.distributed.run, before this process has
        # imported torch or touched a device (a process that has initialised the GPU must never exec or be replaced on this pool); the
        # children's output is passed through (rank 0 prints the one JSON line) and their exit status is this process's.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["MASTER_ADDR"] = "127.0.0.1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    from htk_amd import synth, capi, herest

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("bench.py: --gpus %d under WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # test aid (tests/test_gpu_multi.py on a one-GPU box): every rank on device 0 and the exchange through gloo -- RCCL refuses two ranks
    # on one device; the loop around the collective is the same
    one_device = os.environ.get("HTKAMD_BENCH_ONE_DEVICE_GLOO") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    capi.check(capi.lib().htkamd_set_device(local_rank), "set_device")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    D = 39
    # same model on every rank (model_seed), a different 1250-utterance shard per rank (seed)
    if args.scaling == "strong":                             # config[2] as ONE job: rank r takes utterances r, r+world, ... of it (HERest -p semantics)
        ids = range(rank, args.total_utts, world)
        args.utts = len(ids)
        s = synth.generate_fast(args.states, args.mix, args.phones, 0, args.frames, seed=1000, model_seed=3, utt_ids=ids)
    else:
        s = synth.generate_fast(args.states, args.mix, args.phones, args.utts, args.frames, seed=1000 + rank, model_seed=3)
    pk = s.packed()
    model = capi.Model(pk)
    accs = capi.Accs(model)
    cfg = capi.fb_config(scoreMode={"exact": 0, "mfma": 1, "fast": 3, "bf16": 6, "fastest": 34}[args.score])                               # HERest defaults: pruning off, MINFORPROB 10, -u tmvw

    if args.ragged:                                      # not the headline workload: utterance lengths spread over [frames/2, frames], chains
        rr = np.random.default_rng(77 + rank)            # spread likewise (one model per 12 frames), in random order within the batch
        for u in range(len(s.feats)):
            T = int(rr.integers(args.frames // 2, args.frames + 1))
            s.feats[u] = s.feats[u][:T]; s.seqs[u] = s.seqs[u][: max(1, T // 12)]
    X = np.concatenate(s.feats)
    dX = torch.from_numpy(X).cuda()                      # features resident in HBM before the timed region
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    vec_ptr, vec_n = accs.device_vector()
    acc_t = herest.device_vector_as_tensor(accs, local_rank)
    wire_bulk = int(accs.lay.nEgs)
    wire_buf = torch.empty(wire_bulk, dtype=torch.float32, device=acc_t.device) if (world > 1 and args.wire == "f32") else None
    def exchange():
        herest.all_reduce_accumulators(acc_t, wire=args.wire, bulk=wire_bulk, staging=wire_buf)
    # The same exchange in parts (--exchange-slices): tied states [S i / n, S (i + 1) / n) own a range each of mu / muOcc / va / vaOcc / wt /
    # wtOcc (htkamd_accs_state_ranges); the last part also carries what no state owns (tr, trOcc) and the fp64 counters.  Part i is packed,
    # summed and unpacked on `comm_stream` as soon as the mixture statistics of its states are done, while part i + 1's are being summed.
    n_slices = max(1, args.exchange_slices) if (world > 1 and max(1, args.chunks) == 1) else 1
    part_ranges, part_pos, part_states = [], [], []
    if n_slices > 1:
        try:
            S_ = int(pk["numStates"]); pos_ = 0
            for i_ in range(n_slices):
                s0_, s1_ = S_ * i_ // n_slices, S_ * (i_ + 1) // n_slices
                rg_ = accs.state_ranges(s0_, s1_, with_rest=(i_ == n_slices - 1))
                part_states.append((s0_, s1_)); part_ranges.append(rg_); part_pos.append(pos_)
                pos_ += (sum(l_ for _, l_ in rg_) + 63) & ~63
            assert pos_ <= wire_bulk + 64 * n_slices
        except capi.HtkAmdError:                           # a set whose Gaussians are not in state order: the vector travels whole
            n_slices = 1
    comm_stream = torch.cuda.Stream() if n_slices > 1 else None
    ev_part = [torch.cuda.Event() for _ in range(n_slices)] if n_slices > 1 else []
    part_buf = (torch.empty(wire_bulk + 64 * n_slices, dtype=torch.float32 if args.wire == "f32" else torch.float64, device=acc_t.device) if n_slices > 1 else None)

    def pass_with_exchange(fb, ln):
        """The pass on stream `ln`; with --exchange-slices the exchange too, part by part behind the ranges of states.  True: the accumulators are summed
        over the ranks when the work queued here is done (`stream` waits for it); False: the caller exchanges them."""
        if n_slices <= 1:
            fb.execute(cfg, accs, ln.cuda_stream)
            return False
        if not fb.execute_begin(cfg, accs, ln.cuda_stream):
            return False                                   # a pass of another kind: complete, nothing deferred
        wcode = 1 if args.wire == "f32" else 0
        for i_, (s0_, s1_) in enumerate(part_states):
            fb.execute_mix(s0_, s1_, ln.cuda_stream)
            ev_part[i_].record(ln)
            comm_stream.wait_event(ev_part[i_])
            with torch.cuda.stream(comm_stream):
                n_ = sum(l_ for _, l_ in part_ranges[i_])
                buf_ = part_buf[part_pos[i_]:part_pos[i_] + n_]
                accs.pack_ranges(part_ranges[i_], wcode, buf_.data_ptr(), comm_stream.cuda_stream)
                dist.all_reduce(buf_, op=dist.ReduceOp.SUM)
                accs.unpack_ranges(part_ranges[i_], wcode, buf_.data_ptr(), comm_stream.cuda_stream)
                if i_ == n_slices - 1:
                    dist.all_reduce(acc_t[wire_bulk:], op=dist.ReduceOp.SUM)      # nEgs, totalPr, totalT, the counters: fp64
        ln.wait_stream(comm_stream)
        return True
    # A step is one EM ITERATION of HERest over the rank's shard, nothing left out and nothing carried over from the step before:
    #   ZeroAccs -> [CreateInsts/SetBeamTaper on the host, K1 scoring, K2 beta, K3 alpha + occupation/transition counts, K4 mixture
    #   statistics] -> all-reduce(sum) of the accumulator vector over the ranks -> UpdateModels + rebuild of every scoring table on the
    #   device -> per-utterance results on the host.
    # Iteration k+1 needs iteration k's model, so the device work of two iterations cannot overlap.  The one thing that does not depend
    # on the model's parameters -- the host-side batch tables (CreateInsts / SetBeamTaper: transcriptions and minimum durations) -- is
    # built for iteration k+1 while iteration k's kernels run, and rebuilt inside iteration k+1 should the update have changed a
    # minimum duration (`batch_tables_rebuilt_in_iteration` counts those).  `--chunks C` cuts the shard into C sub-batches alternating
    # over two streams into the one accumulator vector (measured: no gain with the state-per-lane recursions; default 1).
    NCH = max(1, args.chunks)
    U = len(s.feats)
    cuts = [U * c // NCH for c in range(NCH + 1)]
    frame_off_all = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int64)
    chunks = []
    for c in range(NCH):
        u0, u1 = cuts[c], cuts[c + 1]
        fo = (frame_off_all[u0:u1 + 1] - frame_off_all[u0]).astype(np.int32)
        lo = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs[u0:u1]])]).astype(np.int32)
        lb = np.concatenate(s.seqs[u0:u1]).astype(np.int32)
        # two batch contexts per chunk: while iteration k's kernels run on one, the host builds iteration k+1's tables in the other
        chunks.append(dict(fbs=[capi.ForwardBackward(model), capi.ForwardBackward(model)], ready=[False, False], frameOff=fo, labOff=lo, labs=lb,
                           x_ptr=dX.data_ptr() + int(frame_off_all[u0]) * D * 4, n=u1 - u0))
    lanes = [torch.cuda.Stream() for _ in range(min(2, NCH))] if (args.two_streams and NCH > 1) else [stream]
    copy_stream = torch.cuda.Stream()
    ev_chunk = [torch.cuda.Event() for _ in range(NCH)]
    ev_zero = torch.cuda.Event()
    upd = dict(minEgs=3, minVar=args.min_var)                                   # HERest -m 3 (default), -v
    t_parts = np.zeros(4)                                                      # pass, all-reduce, update, results (host clock, rank-local)
    n_reprepared = [0]
    it_no = [0]

    def prep(ch, k, sp):
        ch["fbs"][k].prepare(ch["x_ptr"], ch["frameOff"], ch["labOff"], ch["labs"], sp)
        ch["ready"][k] = True

    def em_iteration(timed: bool, parts=None):
        """One EM iteration.  `parts` (an array of four) asks for the wall-clock split pass | all-reduce | update | results: that costs three
        extra stream synchronisations, so it is taken in a few iterations AFTER the timed region, never inside it."""
        k = it_no[0] & 1
        it_no[0] += 1
        t = [time.perf_counter()]
        accs.zero(sptr)
        ev_zero.record(stream)
        for c, ch in enumerate(chunks):
            ln = lanes[c % len(lanes)]
            ln.wait_event(ev_zero)
            # CreateInsts / SetBeamTaper depend on the transcriptions and the models' minimum durations only: the tables were built
            # during the previous iteration (below) and are rebuilt here if its update changed a minimum duration
            if not (ch["ready"][k] and ch["fbs"][k].prepared_current()):
                prep(ch, k, ln.cuda_stream)
                n_reprepared[0] += int(timed)
            exchanged = pass_with_exchange(ch["fbs"][k], ln)
            ev_chunk[c].record(ln)
        for c, ch in enumerate(chunks):                                        # the kernels are running: next iteration's tables, uploaded on a
            prep(ch, k ^ 1, copy_stream.cuda_stream)                           # stream of their own (htkamd_fb_execute waits for the copy's event)
        for c in range(NCH):
            stream.wait_event(ev_chunk[c])
        if parts is not None:
            stream.synchronize(); t.append(time.perf_counter())
        if world > 1 and not exchanged:
            exchange()                              # the iteration's one exchange: RCCL sum over xGMI
        if parts is not None:
            stream.synchronize(); t.append(time.perf_counter())
        st_upd = model.update_device(accs, stream=sptr, **upd)                   # synchronises the stream
        if parts is not None:
            t.append(time.perf_counter())
        prs, sts = zip(*[ch["fbs"][k].results(sptr) for ch in chunks])
        if parts is not None:
            t.append(time.perf_counter())
            parts[:] += np.diff(t)
        return np.concatenate(prs), np.concatenate(sts), st_upd, k

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the pass with the INITIAL model, outside the timed region: its log-probabilities are checked against the oracle below
    # (run through BOTH batch contexts of every chunk: a context allocates its device workspace at its first pass, and that must not fall
    # into the timed region whatever --warmup is)
    for k0 in (1, 0):
        for attempt in (0, 1):
            accs.zero(sptr)
            for ch in chunks:
                prep(ch, k0, sptr); ch["fbs"][k0].execute(cfg, accs, sptr)
            try:
                pr_init = np.concatenate([ch["fbs"][k0].results(sptr)[0] for ch in chunks])
                break
            except capi.HtkAmdError as e:                   # the fp16 scores' range check (HTKAMD_ERANGE): the bf16 x 3 scores instead, said in the line
                if e.rc != capi.ERANGE or attempt or not (cfg.scoreMode & capi.SCORE_F16):
                    raise
                torch.cuda.synchronize()
                print("bench: %s -- the run goes on with --score bf16" % e, file=sys.stderr)
                cfg.scoreMode = (cfg.scoreMode & ~capi.SCORE_F16) | capi.SCORE_BF16
                args.score = "bf16"
    a_init = accs.download()
    units_local = sum(ch["fbs"][0].frame_states() for ch in chunks)           # (frame, chain state) evaluations of this rank's shard

    def any_rank(flag: bool) -> bool:
        """True on every rank if it is true on one: the ranks take the fp16 -> bf16 decision alike (no sum of statistics of two arithmetics)."""
        if world == 1:
            return flag
        t_ = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=acc_t.device)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return bool(t_.item() > 0)

    def to_bf16(why: str):
        torch.cuda.synchronize()
        if rank == 0:
            print("bench: %s -- the run goes on with --score bf16 on every rank" % why, file=sys.stderr)
        cfg.scoreMode = (cfg.scoreMode & ~capi.SCORE_F16) | capi.SCORE_BF16
        args.score = "bf16"

    asked_f16 = args.score == "fastest" or bool(cfg.scoreMode & capi.SCORE_F16)
    if any_rank(asked_f16 and not (cfg.scoreMode & capi.SCORE_F16)) and (cfg.scoreMode & capi.SCORE_F16):
        to_bf16("a rank's shard does not fit the fp16 scores' range")
    range_hit = [False]
    import gc

    # The timed iterations, host side pipelined: the update is queued in two halves (htkamd_model_update_device_begin / _end) and the NEXT
    # iteration's pass is queued behind its kernels before the host waits for the few bytes the update sends back (transition matrices for
    # the minimum durations, counters) -- the launches of an iteration are issued while the previous one still runs.  Same device work in
    # the same order as em_iteration(); should an update change a minimum duration, the pass queued on the old batch tables is repeated.
    def launch_pass(k, timed):
        accs.zero(sptr)
        ev_zero.record(stream)
        for c, ch in enumerate(chunks):
            ln = lanes[c % len(lanes)]
            ln.wait_event(ev_zero)
            if not (ch["ready"][k] and ch["fbs"][k].prepared_current()):
                prep(ch, k, ln.cuda_stream)
                n_reprepared[0] += int(timed)
            exchanged = pass_with_exchange(ch["fbs"][k], ln)
            ch["fbs"][k].results_begin(ln.cuda_stream)                          # the results' copy in stream order behind the pass
            ev_chunk[c].record(ln)
        for c in range(NCH):
            stream.wait_event(ev_chunk[c])
        return exchanged

    def collect(k):
        try:
            prs, sts = zip(*[ch["fbs"][k].results(sptr) for ch in chunks])
        except capi.HtkAmdError as e:                       # HTKAMD_ERANGE inside the measurement: noted, the loop (and its collectives) go on;
            if e.rc != capi.ERANGE:                         # the ranks settle it together behind the timed region and measure again as bf16
                raise
            range_hit[0] = True
            return None, None, np.zeros(5)
        kt = np.zeros(5)
        for ch in chunks:
            kt += np.array(ch["fbs"][k].kernel_times5())                       # per kernel: summed over the iteration's chunks
        return np.concatenate(prs), np.concatenate(sts), kt

    def set_events(mode):
        for ch in chunks:
            for f in ch["fbs"]:
                f.set_event_mode(mode)

    def measure():
        # the timed iterations record the DOMINANT kernel's own start / stop only (the scoring dispatch: `roofline` is priced on it, live);
        # the stream events between the other kernels are barrier packets worth 20 - 40 us of an iteration -- those kernels' durations come
        # from the un-chunked pass measured alone behind the timed region (`kernel_ms_isolated`, and `kernel_ms` says which is which)
        set_events(1)
        for i in range(args.warmup):
            em_iteration(False)
        ktimes = np.zeros(5)
        st_upd = None
        gc.collect(); gc.disable()                           # (no collector pause inside the few milliseconds that are timed; on again behind them)
        sync_all()
        t0 = time.perf_counter()
        pending, prev_k = False, None
        host_trace = os.environ.get("BENCH_HOST_TRACE")
        for i in range(args.steps):
            kk = it_no[0] & 1
            it_no[0] += 1
            th = [time.perf_counter()]
            exchanged = launch_pass(kk, True)
            th.append(time.perf_counter())
            if pending:
                st_upd = model.update_device_end()
                th.append(time.perf_counter())
                if not all(ch["fbs"][kk].prepared_current() for ch in chunks):      # a minimum duration changed under the pass just queued
                    stream.synchronize()
                    exchanged = launch_pass(kk, True)
                pr, st, kt = collect(prev_k)
                ktimes += kt
                th.append(time.perf_counter())
            if world > 1 and not exchanged:
                exchange()
            model.update_device_begin(accs, stream=sptr, **upd)
            pending, prev_k = True, kk
            th.append(time.perf_counter())
            for ch in chunks:                                                      # next iteration's tables while this one runs: the other context (its pass
                prep(ch, kk ^ 1, copy_stream.cuda_stream)                          # is over: update_device_end waited for it), uploaded on a stream of their own
            th.append(time.perf_counter())
            if host_trace and rank == 0:
                print("host it %d:" % i, " ".join("%.2f" % (1e3 * (b - a)) for a, b in zip(th, th[1:])), file=sys.stderr)
        if pending:
            st_upd = model.update_device_end()
            pr, st, kt = collect(prev_k)
            ktimes += kt
        sync_all()
        set_events(0)
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=acc_t.device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, ktimes, st_upd

    # a box that has just been started runs its first iterations slower (2.4 ms against 2.1 a few seconds later on the same box: clocks,
    # first touches): untimed iterations for a fixed wall time before the W warm-up steps and the K timed ones the contract asks for.
    # They move the model like any iteration; the parity check below starts from the initial parameters again
    if args.prewarm_seconds > 0:
        t_pw = time.perf_counter()
        while any_rank(time.perf_counter() - t_pw < args.prewarm_seconds):     # (the ranks agree on going on: every iteration holds a collective)
            for _ in range(20):
                em_iteration(False)
        model.set_params(mean=pk["mean"], var=pk["var"], compWeight=pk["compWeight"], transP=pk["transP"])
        for ch in chunks:
            ch["ready"] = [False, False]
    dt, ktimes, st_upd = measure()
    if any_rank(range_hit[0]) and (cfg.scoreMode & capi.SCORE_F16):
        # the range check tripped on a model the iterations themselves produced: initial parameters again, bf16 x 3 scores, the whole
        # measurement again (a pass whose scores overflowed has spoilt the model it fed)
        to_bf16("the fp16 scores' range check tripped inside the measurement")
        model.set_params(mean=pk["mean"], var=pk["var"], compWeight=pk["compWeight"], transP=pk["transP"])
        for ch in chunks:
            ch["ready"] = [False, False]
        range_hit[0] = False
        dt, ktimes, st_upd = measure()
        assert not any_rank(range_hit[0])
    ktimes /= max(args.steps, 1)
    a = accs.download()                                                        # the last iteration's summed statistics
    p_dump = model.get_params() if (args.dump_model and rank == 0) else None      # the model of the MEASURED iterations (the split iterations and the side run below move it)
    for i in range(3):                                                         # the split of an iteration's wall clock, outside the timed region
        em_iteration(False, parts=t_parts)
    t_parts /= 3.0

    # one un-chunked pass ALONE (outside the timed region): the kernels' own durations, the latency of a single pass
    fb1 = capi.ForwardBackward(model)
    frameOff = frame_off_all.astype(np.int32)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
    labs = np.concatenate(s.seqs).astype(np.int32)
    acc1 = capi.Accs(model)
    lat = []
    for rep in range(3):
        torch.cuda.synchronize(); tl = time.perf_counter()
        acc1.zero(sptr); fb1.prepare(dX.data_ptr(), frameOff, labOff, labs, sptr); fb1.execute(cfg, acc1, sptr); fb1.results(sptr)
        lat.append(time.perf_counter() - tl)
    ktimes_solo = np.array(fb1.kernel_times5())
    torch.cuda.synchronize()
    kernel_ms_source = {"score": "the dispatch's own start / stop, every timed iteration"}
    for i_, n_ in ((1, "beta"), (2, "alpha"), (3, "stats"), (4, "mix_stats")):
        if ktimes[i_] < 0:                                                     # not recorded in the timed iterations (set_events(1))
            ktimes[i_] = ktimes_solo[i_]
            kernel_ms_source[n_] = "stream events around the kernel in one un-chunked pass of the same shard behind the timed region"

    # the same timed iterations once more on the fp16 x 2 scores (from the initial model again): reported beside the line, never `value`
    fastest_side = None
    if args.also_fastest and world == 1 and args.score == "bf16":
        keep_mode = cfg.scoreMode
        try:
            model.set_params(mean=pk["mean"], var=pk["var"], compWeight=pk["compWeight"], transP=pk["transP"])
            for ch in chunks:
                ch["ready"] = [False, False]
            cfg.scoreMode = 34
            range_hit[0] = False
            dt_f, kt_f, _ = measure()
            kt_f = kt_f / max(args.steps, 1)
            if not range_hit[0]:
                fastest_side = {"score_mode": "fastest (fp16 x 2 split operands, k_score_f16w)", "ms_per_step": dt_f / args.steps * 1e3,
                                "value": float(units_local) * args.steps / dt_f, "kernel_ms_score": float(kt_f[0]) * 1e3,
                                "note": "parity of this mode: tests/test_gpu_headline_parity.py -- accumulator deviations 0.6 - 1.0 of the 1e-4 bar depending on the sample, "
                                        "against <= 0.5 for the bf16 x 3 scores the line is measured with"}
            else:
                fastest_side = {"score_mode": "fastest", "error": "the fp16 range check tripped (HTKAMD_ERANGE)"}
        except capi.HtkAmdError as e:
            fastest_side = {"score_mode": "fastest", "error": str(e)[:200]}
        cfg.scoreMode = keep_mode
        torch.cuda.synchronize()
    units_total = float(a["nEval"]) if world > 1 else float(units_local)
    utts_total = float(a["nUttDone"])
    value = units_total * args.steps / dt

    if rank == 0:
        flop_unit = FLOP_PER_FRAME_STATE(args.mix, D)
        kname = {"exact": "k_score_exact<39>", "bf16": "k_score_bf16w<5>" if 31 <= D <= 39 and args.mix <= 16 else "k_score_bf16w", "fastest": "k_score_f16w<3>"}.get(args.score, "k_score_mfma<20>")
        # HBM-side bytes per launch of every kernel from the committed PMC passes (FETCH_SIZE + WRITE_SIZE, separate runs: tools/prof_r03.sh),
        # same workload only
        traffic_of, traffic_upper = {}, {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TRAFFIC)))
            w = tj["workload"]
            if (w["states"], w["mix"], w["utts_per_gpu"], w["frames"], w.get("chunks", 1)) == (args.states, args.mix, args.utts, args.frames, NCH):
                # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies a 16-byte-per-lane streaming read at HALF its bytes -- doubled for the
                # kernels whose reads are of that width (the matrix-core scoring kernels' table tiles); WRITE_SIZE is exact for 16-byte
                # stores and float atomics; other widths (the recursions' 8-byte columns, 4-byte scores) are uncalibrated and left as counted
                for kn, k in tj["kernels"].items():
                    base = kn.split("<")[0]
                    traffic_of[base] = (k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0              # as counted
                    if base in ("k_score_bf16w", "k_score_f16w", "k_score_bf16", "k_score_f16"):     # an UPPER bound: every fetched byte taken as a 16-byte-per-lane read
                        traffic_upper[base] = (2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0
        except (OSError, KeyError, ValueError):
            traffic_of, traffic_upper = {}, {}
        # Per kernel of the pass (its launches of one iteration, one per chunk: algorithmic work of the iteration / summed duration).
        # Scoring: SURVEY §8(d)'s M (4D + 8) flop per frame-state against the fp32 matrix peak.  Recursions: §8(d)'s 36 bytes per
        # in-beam frame-state against HBM -- the score read and the beta column written in the beta pass (12), the beta column read and the
        # alpha column written in the alpha pass (16), the alpha column read back by the frame-parallel statistics (8).
        k1 = float(ktimes[0])
        achieved = units_local * flop_unit / k1 / 1e12 if k1 > 0 else 0.0
        per_kernel = {"score": {"kernel": kname, "bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_PEAK_TFLOPS,
                                "ms": k1 * 1e3, "traffic": traffic_of.get(kname.split("<")[0]), "flop_per_unit": flop_unit}}
        lean = os.environ.get("HTKAMD_LR_LEAN", "15") != "0"
        for key_, ki, kn, bytes_unit in (("beta", 1, "k_beta_np2" if lean else "k_beta_lr", 12), ("alpha", 2, "k_alpha_f2" if lean else "k_alpha_lr", 16),
                                         ("stats", 3, "k_stats_sp" if lean else "k_stats_lr", 8)):
            tk = float(ktimes[ki])
            ach = units_local * bytes_unit / tk / 1e9 if tk > 0 else 0.0
            per_kernel[key_] = {"kernel": kn, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "ms": tk * 1e3, "traffic": traffic_of.get(kn), "bytes_per_unit": bytes_unit}
        # mixture statistics: SURVEY §8(d)'s accumulation row -- 2 D FMAs and 2 x 2 D x 4 bytes of accumulator read-modify-write per retained
        # (frame, state, component) triple (624 bytes at D = 39); the triples are counted on the device (htkamd_fb_mix_counts)
        pairs_ = triples_ = 0
        for ch in chunks:                                   # the last pass of every sub-batch (either context: the same utterances under nearly the same model)
            p_c, t_c = max(ch["fbs"][0].mix_counts(), ch["fbs"][1].mix_counts())
            pairs_ += p_c; triples_ += t_c
        tmix = float(ktimes[4])
        per_kernel["mix"] = {"kernel": "k_mixstate (+ k_mixhits / k_rec_* for what its buckets turn away)", "bound": "hbm", "ms": tmix * 1e3,
                             "traffic": traffic_of.get("k_mixstate"), "bytes_per_unit": 16 * D, "unit_is": "(frame, state, component) triple past the MINFORPROB prune",
                             "units_per_launch": triples_, "pairs_per_launch": pairs_}
        if triples_ > 0 and tmix > 0:
            # `frac` is what the memory system really moved (counter bytes of the committed PMC passes) over the HBM peak; SURVEY §8(d)'s row -- 624 bytes of
            # accumulator read-modify-write per triple -- is kept beside it as an ALGORITHMIC figure: those bytes never reach memory here (a state's sums
            # stay in registers until one atomic per element), so it is no utilisation of anything.  The kernel is bound by latency (its ablations: DESIGN §4)
            am = triples_ * 16.0 * D / tmix / 1e9
            tr_ = traffic_of.get("k_mixstate")
            per_kernel["mix"].update({"bound": "latency (dependent phases per 32-pair chunk, one atomic per accumulator element and state)",
                                      "achieved": (tr_ / tmix / 1e9) if tr_ else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (tr_ / tmix / 1e9 / HBM_PEAK_GBS) if tr_ else None,
                                      "algorithmic": {"achieved": am, "unit": "GB/s", "over_hbm_peak": am / HBM_PEAK_GBS,
                                                      "note": "SURVEY §8(d)'s accumulation row; not a bandwidth the kernel draws"}})
        # The scoring kernel's `frac` is EXECUTED flops over the dense peak of the pipe it runs on (<= 1 by construction); SURVEY §8(d)'s
        # algorithmic count (the fp32 algorithm's M (4D + 8) flop per frame-state) stays beside it under `algorithmic`.
        sc = per_kernel["score"]
        sc["algorithmic"] = {"achieved": achieved, "unit": "TFLOP/s", "flop_per_unit": flop_unit, "over_fp32_matrix_peak": achieved / FP32_PEAK_TFLOPS,
                             "note": "SURVEY §8(d)'s unit; not a roofline fraction when the kernel runs on another pipe"}
        if args.score in ("fastest", "bf16"):
            # what the matrix pipe executes: three fp16 (six bf16) piece products over three K chunks of 32 (13 dimensions as
            # (x^2, x) pairs + the chunk's constant, padded), per component
            kpad = ((D + 14) // 15) * 32
            nprod = 3 if args.score == "fastest" else 6
            if args.score == "bf16" and 31 <= D <= 39 and args.mix <= 16 and not os.environ.get("HTKAMD_BF16_CHUNKED"):
                kpad = 80                                   # k_score_bf16w<5>: the 2 D + 2 terms in five k-steps of 16 (gmm_bf16.hip, the dense layout)
            exe = units_local * args.mix * kpad * 2 * nprod / k1 / 1e12 if k1 > 0 else 0.0
            # `frac` as SURVEY §8(d) defines it: ALGORITHMIC flops over the dense peak of the pipe the kernel runs on; what the pipe executes
            # (the split operands' piece products, padding included) beside it as executed_frac
            sc.update({"achieved": achieved, "peak": F16_PEAK_TFLOPS, "frac": achieved / F16_PEAK_TFLOPS, "flop_per_unit": flop_unit,
                       "executed": {"achieved": exe, "unit": "TFLOP/s", "flop_per_unit": args.mix * kpad * 2 * nprod}, "executed_frac": exe / F16_PEAK_TFLOPS,
                       "traffic_upper_bound": traffic_upper.get(kname.split("<")[0]),
                       "pipe": "v_mfma_f32_32x32x16_f16, operands split in two fp16 pieces, fp32 accumulate" if args.score == "fastest"
                               else "v_mfma_f32_32x32x16_bf16, operands split in three bf16 pieces, fp32 accumulate"})
        elif args.score == "exact":
            sc["pipe"] = "packed fp32 VALU (the reference's four roundings per dimension, no FMA)"
        else:
            sc["pipe"] = "v_mfma_f32_16x16x4_f32"
        # the matrix-core work of a pass, counted by the host from the task list (htkamd_fb_score_work): blocks of (32 frames of a wavefront, pair of
        # chain states) the kernel issues its products for, what it issued before Setotprob's per-state frame ranges reached it, what the units need
        sw = np.sum([ch["fbs"][0].score_work() for ch in chunks], axis=0)
        mi_blk = {"bf16": 30 if (31 <= D <= 39 and args.mix <= 16 and not os.environ.get("HTKAMD_BF16_CHUNKED")) else 36, "fastest": 15}.get(args.score)
        score_work = {"unit": "(wavefront of 32 frames, pair of chain states) blocks per pass", "issued": int(sw[0]), "issued_without_frame_ranges": int(sw[1]),
                      "needed": int(sw[2]), "needed_over_issued": float(sw[2]) / max(float(sw[0]), 1.0),
                      "frame_ranges_in_kernel": args.score == "bf16" and not os.environ.get("HTKAMD_NO_TAPER_SKIP"),      # (k_score_f16w works on every block of its tasks: the skip cost it 7 %)
                      "matrix_instructions_per_block": mi_blk, "mfma_issued": int(sw[0 if args.score == "bf16" else 1]) * mi_blk if mi_blk else None,
                      "mfma_needed": int(sw[2]) * mi_blk if mi_blk else None}
        dom = max(("score", "beta", "alpha", "stats"), key=lambda k_: per_kernel[k_]["ms"])
        out = {
            "metric": "herest_gmm_frame_state_loglik_per_sec",
            "value": value,
            "unit": "frame-state log-lik/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": {"exact": "f32 (scores, the reference's float arithmetic operation for operation) + f64 (alpha / beta / accumulators)",
                      "mfma": "f32 (scores: fp32 matrix-core products) + f64 (alpha / beta / accumulators)",
                      "fast": "f32 (scores: fp32 matrix-core products) + f64 (alpha / beta with fp32-transcendental log-add, accumulators)",
                      "bf16": "f32 (scores: bf16x3 split operands, fp32 accumulate) + f64 (alpha / beta with fp32-transcendental log-add, accumulators)",
                      "fastest": "f32 (scores: fp16x2 split operands, fp32 accumulate) + f64 (alpha / beta with fp32-transcendental log-add, accumulators)"}[args.score],
            "data": "synthetic",
            "config": {"workload": "HERest EM iteration (pass + accumulator all-reduce + model update), %d tied states x %d mix, D=39, "
                                   "%d x %d-frame utterances per GPU (BASELINE config[2]: 10k utterances sharded 8-way)" % (args.states, args.mix, args.utts, args.frames),
                       "states": args.states, "mix": args.mix, "utts_per_gpu": args.utts, "frames": args.frames, "chunks": NCH,
                       "update": "HERest -m 3 -v %g, on the device" % args.min_var,
                       "exchange_slices": (n_slices if world > 1 else None),      # parts the exchange travels in, each behind its range of states (1: one all-reduce behind the pass)
                       "wire": (args.wire if world > 1 else None),      # (the C library's htkamd_accs_allreduce defaults to fp64; tools/herest --wire f32 is this exchange)
                       "parallelism": "utterance shards, 1 all-reduce of %d accumulators per iteration (%s on the wire)" % (vec_n, args.wire if world > 1 else "no exchange at N = 1")},
            "herest_utterances_per_sec": utts_total * args.steps / dt,
            "utterances_ok": utts_total,
            "em_iteration_ms": dt / args.steps * 1e3,
            "batch_tables_rebuilt_in_iteration": n_reprepared[0],
            # split of an iteration's wall clock, from three iterations AFTER the timed region run with a stream synchronisation after every
            # part (the timed iterations carry none but the one the model update needs: their sum is above ms_per_step)
            "em_iteration_parts_ms": {"pass": t_parts[0] * 1e3, "allreduce": t_parts[1] * 1e3, "update_and_refresh": t_parts[2] * 1e3, "results": t_parts[3] * 1e3},
            "pass_latency_ms": float(np.median(lat)) * 1e3,
            "avg_logprob_per_frame": float(a_init["totalPr"] / a_init["totalT"]) if a_init["totalT"] else None,
            "avg_logprob_per_frame_last_iteration": float(a["totalPr"] / a["totalT"]) if a["totalT"] else None,
            "update_stats_last_iteration": st_upd,
            "kernel_ms": {"score": k1 * 1e3, "beta": ktimes[1] * 1e3, "alpha": ktimes[2] * 1e3, "stats": ktimes[3] * 1e3, "mix_stats": ktimes[4] * 1e3},
            "kernel_ms_source": kernel_ms_source,
            "score_work": score_work,
            "score_mode": args.score,
            "streams": len(lanes),
            # the kernel with the largest total time in the timed iterations
            "roofline": dict(per_kernel[dom], units_per_launch=units_local / NCH, launches_per_step=NCH,
                             traffic_source="profiles/%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, bytes per launch as counted; "
                                            "traffic_upper_bound: FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 tallies 16-byte-per-lane streaming reads at half; "
                                            "this kernel's table tiles are such reads, its feature rows and task words are not)" % PROFILE_TRAFFIC
                             if dom == "score" and args.score in ("bf16", "fastest") else "profiles/%s: FETCH_SIZE + WRITE_SIZE per launch as counted (8- and 4-byte accesses: uncalibrated on gfx950)" % PROFILE_TRAFFIC),
            "roofline_kernels": per_kernel,
        }
        if fastest_side is not None:
            out["fastest_mode"] = fastest_side
        if ktimes_solo[0] > 0:
            out["kernel_ms_isolated"] = {"score": ktimes_solo[0] * 1e3, "beta": ktimes_solo[1] * 1e3, "alpha": ktimes_solo[2] * 1e3, "stats": ktimes_solo[3] * 1e3, "mix_stats": ktimes_solo[4] * 1e3}
        if args.cpu_seconds > 0 and world == 1:               # the CPU leg runs at N = 1 only
            per_utt = units_local / max(len(s.feats), 1)
            n, cdt, opr, oacc, n_timed = cpu_baseline(s, pk, args.cpu_seconds)
            # the checker: the oracle's utterance log-probabilities for the same utterances under the same (initial) model
            tol = 1e-10 if args.score == "exact" else 1e-6
            worst = float(np.max(np.abs(pr_init[:n] - opr) / np.abs(opr))) if n else 0.0
            assert worst <= tol, "bench: utterance log-probabilities differ from the oracle: max relative %.3g over %d utterances" % (worst, n)
            # ... and its accumulators: the same utterances alone through the HIP path in the bench's mode, under the initial model.
            # Counts within 1e-4 * max(|ref|, 1e-3); first- and second-order sums (kept about the current mean: entries near zero are
            # cancellations) within 1e-4 * max(|ref|, the Gaussian's occupancy).
            m0 = capi.Model(pk); fbc = capi.ForwardBackward(m0); accc = capi.Accs(m0)
            fo_c = frame_off_all[:n + 1].astype(np.int32)
            lo_c = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs[:n]])]).astype(np.int32)
            fbc.prepare(dX.data_ptr(), fo_c, lo_c, np.concatenate(s.seqs[:n]).astype(np.int32), sptr)
            fbc.execute(cfg, accc, sptr); fbc.results(sptr)
            ac = accc.download()
            worst_acc = {}
            for k_ in ("muOcc", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
                ref_ = np.asarray(getattr(oacc, k_), np.float64).reshape(-1)
                worst_acc[k_] = float(np.max(np.abs(np.asarray(ac[k_], np.float64).reshape(-1) - ref_) / np.maximum(np.abs(ref_), 1e-3)))
            occ_ = np.maximum(np.asarray(oacc.muOcc, np.float64), 1e-3)[:, None]
            few_ = occ_[:, 0] < 3.0                              # Gaussians the sample gives fewer than three frames
            worst_few = {}
            pure_ = {}                                           # entries beyond 1e-4 of THEIR OWN value (no floor, no scale): counts, of how many
            for k_ in ("muOcc", "wt", "wtOcc", "tr", "trOcc"):
                ref_ = np.asarray(getattr(oacc, k_), np.float64).reshape(-1); got_ = np.asarray(ac[k_], np.float64).reshape(-1)
                nz_ = np.abs(ref_) > 0
                pure_[k_] = {"n": int(nz_.sum()), "n_above_1e4": int((np.abs(got_ - ref_)[nz_] > 1e-4 * np.abs(ref_)[nz_]).sum())}
            for k_ in ("mu", "va"):
                ref_ = np.asarray(getattr(oacc, k_), np.float64).reshape(occ_.shape[0], -1)
                rel_ = np.abs(np.asarray(ac[k_], np.float64).reshape(ref_.shape) - ref_) / np.maximum(np.abs(ref_), occ_)
                nz_ = np.abs(ref_) > 0
                pure_[k_] = {"n": int(nz_.sum()), "n_above_1e4": int((np.abs(np.asarray(ac[k_], np.float64).reshape(ref_.shape) - ref_)[nz_] > 1e-4 * np.abs(ref_)[nz_]).sum()),
                             "note": "sums about the current mean: an entry near zero is a cancellation, its own value no scale for it"}
                worst_acc[k_] = float(np.max(rel_[~few_])) if (~few_).any() else 0.0
                worst_few[k_] = float(np.max(rel_[few_])) if few_.any() else 0.0
                if os.environ.get("BENCH_ACC_DETAIL"):
                    order_ = np.argsort(-rel_.max(1))[:8]
                    print("# %s worst Gaussians: " % k_ + "; ".join("g=%d occ=%.4g rel=%.3g" % (g_, occ_[g_, 0], rel_[g_].max()) for g_ in order_), file=sys.stderr)
            assert np.array_equal(np.asarray(ac["nEgs"]).astype(np.int64), np.asarray(oacc.nEgs).astype(np.int64)), "bench: example counts differ from the oracle"
            # the bar: every count of every Gaussian, and the first- and second-order sums of the Gaussians with at least three frames of
            # occupancy in the sample (a sum over one or two frames carries the posterior noise of those frames undiminished; reported
            # below as `sums_of_gaussians_under_3_frames`, unasserted: the whole job gives every Gaussian 8x the shard's frames)
            assert max(worst_acc.values()) <= 1e-4, "bench: accumulators differ from the oracle: %r (Gaussians under three frames: %r)" % (worst_acc, worst_few)
            # ... and those sums themselves -- one or two frames' posteriors do not average out -- at 1e-4 too in the mode the line is measured with (bf16 x 3 scores:
            # observed 5.2e-5 over the ~11 000 such Gaussians of the shard; the exact mode far below); the fp16 x 2 scores of `--score fastest` have been seen
            # at 6e-5 .. 1e-4 there and keep 2e-4.  A real regression of the scores shows here first and fails the run
            bar_few = 2e-4 if (cfg.scoreMode & capi.SCORE_F16) else 1e-4
            assert max(worst_few.values()) <= bar_few, "bench: sums of the Gaussians under three frames differ from the oracle: %r" % (worst_few,)
            worst_acc["sums_of_gaussians_under_3_frames"] = dict(worst_few, gaussians=int(few_.sum()), asserted_at=bar_few)
            del fbc, accc, m0
            out["oracle_check"] = {"utterances": n, "max_rel_diff_logprob": worst, "tolerance": tol,
                                   "accumulators_max_rel_diff": worst_acc, "accumulators_tolerance": 1e-4,
                                   "accumulators_pure_relative_1e4": pure_,
                                   "avg_logprob_per_frame_oracle": float(np.sum(opr) / sum(s.feats[u].shape[0] for u in range(n))),
                                   "avg_logprob_per_frame_hip": float(np.sum(pr_init[:n]) / sum(s.feats[u].shape[0] for u in range(n)))}
            port = {"value": n_timed * per_utt / cdt, "unit": "frame-state log-lik/s", "cores": 1, "kind": "port",
                    "utterances_per_sec": n_timed / cdt,
                    "sample": "%d utterances of the same shard through oracle/htk_oracle.c (scalar C restatement of "
                              "HFB/HModel, bit-exact vs the reference), %.1f s on one host core (the oracle then goes on, untimed, "
                              "through the rest of the shard: the check in `oracle_check` covers all %d)" % (n_timed, cdt, n)}
            cores = host_cores() if args.cpu_workers <= 0 else args.cpu_workers
            n_ref = 150                                        # utterances per process and round: ~3 s of HERest per process in the first round, ~6 s in the second
            ref = cpu_baseline_reference(s, pk, n_ref, workers=cores)
            ref1 = cpu_baseline_reference(s, pk, 60, workers=1) if cores > 1 else ref
            if ref is not None:
                out["cpu_baseline"] = {"value": ref[0] * per_utt / ref[1], "unit": "frame-state log-lik/s", "cores": ref[2], "kind": "reference",
                                       "utterances_per_sec": ref[0] / ref[1],
                                       "sample": "the reference's own HERest (oracle/_ref) as %d parallel `-p k` processes, one per host core (affinity mask, physical cores, "
                                                 "cgroup quota: %s), %d utterances of the same shard each; model loading differenced out (round over 2n minus round over n "
                                                 "utterances per process)" % (ref[2], cgroup_cpu_quota(), n_ref)}
                if ref1 is not None:
                    one = ref1[0] * per_utt / ref1[1]
                    out["cpu_baseline"]["one_core"] = {"value": one, "utterances_per_sec": ref1[0] / ref1[1], "sample": "one HERest process, 60 utterances, same differencing"}
                    out["cpu_baseline"]["parallel_efficiency"] = (ref[0] * per_utt / ref[1]) / (one * ref[2]) if one > 0 else None
                out["cpu_baseline_port"] = port
            else:
                out["cpu_baseline"] = port
        if args.extras and world == 1:
            try:
                out["other_paths"] = other_paths(s, pk, dX, frame_off_all, cpu_utts=2 if args.cpu_seconds > 0 else 0)
                if "k_decode" in traffic_of and "hvite_decoding" in out["other_paths"]:          # counter bytes of the token kernel, per launch (same PMC passes)
                    out["other_paths"]["hvite_decoding"]["roofline"]["traffic"] = traffic_of["k_decode"]
                try:
                    out["other_paths"]["mfcc"] = mfcc_leg(pk, cpu_files=40 if args.cpu_seconds > 0 else 0)
                except Exception as e:  # noqa: BLE001
                    out["other_paths"]["mfcc"] = {"error": repr(e)[:300]}
                if args.extras >= 1 and args.cpu_seconds > 0:
                    # BASELINE config[2] as ONE job on ONE GPU (10 000 utterances in one batch: the fastest way one GPU does it -- eight sub-batches
                    # on two streams take 17.5 ms for this 14.7): the denominator of the 8-GPU speed-up
                    out["other_paths"]["strong_1gpu"] = subprocess_leg(
                        ["bench.py", "--gpus", "1", "--scaling", "strong", "--chunks", "1", "--cpu-seconds", "0", "--extras", "0", "--also-fastest", "0", "--steps", "10", "--warmup", "2"], 900,
                        lambda j: {k_: j.get(k_) for k_ in ("value", "unit", "ms_per_step", "herest_utterances_per_sec", "utterances_ok", "config", "kernel_ms")})
                    # the link-compatible boundary: the reference's unchanged HERest.o over the HFB shim, files per second (tools/shim_latency.py)
                    out["other_paths"]["boundary"] = subprocess_leg(["tools/shim_latency.py", "200"], 600, lambda j: j)
            except Exception as e:  # noqa: BLE001  (reported beside the line, never instead of it)
                out["other_paths"] = {"error": repr(e)[:300]}
        print(json.dumps(out))
    if args.dump_model and rank == 0:
        p_ = p_dump
        np.savez(args.dump_model, mean=p_["mean"], var=p_["var"], compWeight=p_["compWeight"], transP=p_["transP"], totalPr=a["totalPr"], nUttDone=a["nUttDone"],
                 accVec=a["vec"], accBulk=int(accs.lay.nEgs),
                 accLay=np.array([int(getattr(accs.lay, n)) for n in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs")]))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
