#!/usr/bin/env python3
"""bench.py -- HERest hot path (GMM scoring -> forward-backward -> statistics -> accumulator all-reduce)
on the configuration BASELINE.json's metric is quoted on: 5k tied states x 16 mixtures, 39-dim features,
500-frame synthetic utterances; config[2] sharded 8-way = 1250 utterances per GPU (weak scaling: every rank
processes its own 1250-utterance shard, the accumulator vector is summed with RCCL once per pass).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one embedded Baum-Welch pass over the rank's shard with the features already resident in HBM:
CreateInsts/beam taper on the host, K1 scoring, K2 beta, K3 alpha + occupation/transition statistics,
K4 mixture statistics, all-reduce(sum) of the fp64 accumulator vector, per-utterance results read back.
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


def cpu_baseline(s, pk, budget_s: float):
    """Oracle (CPU restatement of the reference, oracle/htk_oracle.c) timed on one host core over a bounded
    sample of the same workload.  Reported, never the thing measured as `value`."""
    from oracle import pyoracle as po
    om = po.Model(pk)
    acc = po.Accs(om)
    cfg = po.fb_cfg()
    n, nev, t0 = 0, 0, time.perf_counter()
    lib = po.lib()
    import ctypes as C
    while n < len(s.feats):
        rc, pr, _ = po.fb_utt(om, cfg, s.feats[n], s.seqs[n], acc)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return n, dt


def cpu_baseline_reference(s, pk, n_utt: int):
    """The reference's own HERest (oracle/_ref/HERest, built from /root/reference by oracle/Makefile and shipped with the tree)
    on one host core over the first utterances of the same shard.  Model loading (a 25 MB text MMF) is taken out by
    differencing a run over n and a run over 2n utterances.  Returns (utterances, seconds) or None if the binary is absent."""
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
        n2 = min(2 * n_utt, len(s.feats))
        for u in range(n2):
            synth.write_htk_param(os.path.join(d, "u%05d.mfc" % u), s.feats[u], kind=9)
            with open(os.path.join(d, "u%05d.lab" % u), "w") as f:
                f.write("\n".join(names[int(h)] for h in s.seqs[u]) + "\n")
        open(os.path.join(d, "config"), "w").close()
        times = []
        for n in (n2 // 2, n2):
            with open(os.path.join(d, "scp"), "w") as f:
                f.write("\n".join(os.path.join(d, "u%05d.mfc" % u) for u in range(n)) + "\n")
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-S", os.path.join(d, "scp"), "-L", d,
                                "-M", os.path.join(d, "out"), os.path.join(d, "hmmlist")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0:
                return None
            times.append(time.perf_counter() - t0)
        dt = times[1] - times[0]
        return (n2 - n2 // 2, dt) if dt > 0 else None
    except OSError:
        return None
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--states", type=int, default=5000)
    ap.add_argument("--mix", type=int, default=16)
    ap.add_argument("--phones", type=int, default=6000)
    ap.add_argument("--utts", type=int, default=1250, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--ragged", type=int, default=0, help="1: utterance lengths uniform in [frames/2, frames] (a look at mixed batches; the headline run uses 0)")
    ap.add_argument("--score", choices=["exact", "mfma", "fast"], default="fast",
                    help="scoring arithmetic: exact = bit-identical to the reference (packed FP32 VALU); mfma = fp32 matrix-core GEMM, 1e-4 tolerance class")
    ap.add_argument("--two-streams", type=int, default=1, help="run the alternating batch contexts on their own streams (1) or on one stream (0)")
    ap.add_argument("--contexts", type=int, default=2, help="batch contexts in flight (each with its own work space and accumulator vector)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from htk_amd import synth, capi, herest

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world), file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    capi.check(capi.lib().htkamd_set_device(local_rank), "set_device")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    D = 39
    # same model on every rank (model_seed), a different 1250-utterance shard per rank (seed)
    s = synth.generate_fast(args.states, args.mix, args.phones, args.utts, args.frames, seed=1000 + rank, model_seed=3)
    pk = s.packed()
    model = capi.Model(pk)
    accs = capi.Accs(model)
    fb = capi.ForwardBackward(model)
    cfg = capi.fb_config(scoreMode={"exact": 0, "mfma": 1, "fast": 3}[args.score])                               # HERest defaults: pruning off, MINFORPROB 10, -u tmvw

    if args.ragged:                                      # not the headline workload: utterance lengths spread over [frames/2, frames], chains
        rr = np.random.default_rng(77 + rank)            # spread likewise (one model per 12 frames), in random order within the batch
        for u in range(len(s.feats)):
            T = int(rr.integers(args.frames // 2, args.frames + 1))
            s.feats[u] = s.feats[u][:T]; s.seqs[u] = s.seqs[u][: max(1, T // 12)]
    X = np.concatenate(s.feats)
    frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int32)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
    labs = np.concatenate(s.seqs).astype(np.int32)
    dX = torch.from_numpy(X).cuda()                      # features resident in HBM before the timed region
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    vec_ptr, vec_n = accs.device_vector()
    # Two batch contexts AND two accumulator vectors alternate: the host-side preparation of a pass (CreateInsts /
    # SetBeamTaper on a worker pool, 0.7 ms) and its launches overlap the previous pass still running on the device, and the
    # previous pass's all-reduce (52 MB of fp64 over xGMI, on a side stream) overlaps this pass's kernels -- what a training
    # loop over many batches does.  Every pass does all of its work: zero, prepare, score, beta, alpha, statistics,
    # all-reduce; its per-utterance results are collected one pass later.  The same stream/event choreography runs at
    # N = 1 (without the collective), so the single-GPU run exercises it.
    NC = max(2, args.contexts)
    fbs = [fb] + [capi.ForwardBackward(model) for _ in range(NC - 1)]
    accs2 = [accs] + [capi.Accs(model) for _ in range(NC - 1)]
    acc_ts = [herest.device_vector_as_tensor(a, local_rank) for a in accs2]
    comm = torch.cuda.Stream()
    # ... and the two contexts run on two streams, so that the latency-bound recursions of one pass (1250 wavefronts, most
    # of the machine idle) share the GPU with the compute-bound scoring of the next
    lanes = [torch.cuda.Stream() for _ in range(NC)] if args.two_streams else [stream] * NC
    ev_done = [torch.cuda.Event() for _ in range(NC)]    # pass finished accumulating into accs2[k] (its stream)
    ev_red = [torch.cuda.Event() for _ in range(NC)]     # all-reduce of accs2[k] finished (side stream)
    red_pending = [False] * NC

    def launch(i):
        k = i % NC
        f = fbs[k]
        st_k = lanes[k]; sp = st_k.cuda_stream
        if red_pending[k]:
            st_k.wait_event(ev_red[k])                   # accs2[k] is still being summed from two passes ago
        accs2[k].zero(sp)
        f.prepare(dX.data_ptr(), frameOff, labOff, labs, sp)
        f.execute(cfg, accs2[k], sp)
        ev_done[k].record(st_k)
        with torch.cuda.stream(comm):
            comm.wait_event(ev_done[k])
            if world > 1:
                herest.all_reduce_accumulators(acc_ts[k])   # the pass's one exchange: RCCL sum over xGMI
            ev_red[k].record(comm)
        red_pending[k] = True
        return f

    def sync_all():
        torch.cuda.synchronize()                         # both streams
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        launch(i).results(sptr)
    ktimes = np.zeros(4)
    sync_all()
    t0 = time.perf_counter()
    inflight = []
    for i in range(args.steps):
        inflight.append(launch(i))
        if len(inflight) >= NC:                          # collect the oldest pass before its context is reused
            old = inflight.pop(0)
            pr, st = old.results(sptr)                   # waits for that pass only
            ktimes += np.array(old.kernel_times())
    for old in inflight:
        pr, st = old.results(sptr)
        ktimes += np.array(old.kernel_times())
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ktimes /= max(args.steps, 1)

    a = accs2[(args.steps - 1) % NC].download() if args.steps > 0 else accs.download()
    # one more pass ALONE on the device (outside the timed region): the kernels' durations without a neighbour stream
    ktimes_solo = None
    if args.two_streams:
        accs2[0].zero(sptr); fbs[0].prepare(dX.data_ptr(), frameOff, labOff, labs, sptr); fbs[0].execute(cfg, accs2[0], sptr)
        fbs[0].results(sptr)
        ktimes_solo = np.array(fbs[0].kernel_times())
        torch.cuda.synchronize()
    n_ok_local = int((st == capi.UTT_OK).sum())
    units_local = fb.frame_states()                      # (frame, chain state) evaluations of this rank's shard
    units_total = float(a["nEval"]) if world > 1 else float(units_local)
    utts_total = float(a["nUttDone"])
    value = units_total * args.steps / dt

    if rank == 0:
        flop_unit = FLOP_PER_FRAME_STATE(args.mix, D)
        kname = "k_score_mfma<20>" if args.score != "exact" else "k_score_exact<39>"
        traffic = None                                       # HBM-side bytes per launch from the committed PMC pass, same workload only
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")))
            w = tj["workload"]
            if (w["states"], w["mix"], w["utts_per_gpu"], w["frames"]) == (args.states, args.mix, args.utts, args.frames):
                k = tj["kernels"][kname]
                traffic = (k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0
        except (OSError, KeyError, ValueError):
            traffic = None
        k1 = float(ktimes[0])
        achieved = units_local * flop_unit / k1 / 1e12 if k1 > 0 else 0.0
        out = {
            "metric": "herest_gmm_frame_state_loglik_per_sec",
            "value": value,
            "unit": "frame-state log-lik/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "HERest pass, %d tied states x %d mix, D=39, %d x %d-frame utterances per GPU "
                                   "(BASELINE config[2]: 10k utterances sharded 8-way)" % (args.states, args.mix, args.utts, args.frames),
                       "states": args.states, "mix": args.mix, "utts_per_gpu": args.utts, "frames": args.frames,
                       "parallelism": "utterance shards, 1 all-reduce of %d fp64 accumulators per pass" % vec_n},
            "herest_utterances_per_sec": utts_total * args.steps / dt,
            "utterances_ok": utts_total,
            "avg_logprob_per_frame": float(a["totalPr"] / a["totalT"]) if a["totalT"] else None,
            "kernel_ms": {"score": ktimes[0] * 1e3, "beta": ktimes[1] * 1e3, "alpha_stats": ktimes[2] * 1e3, "mix_stats": ktimes[3] * 1e3},
            "score_mode": args.score,
            "streams": NC if args.two_streams else 1, "contexts": NC,
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": FP32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_PEAK_TFLOPS, "traffic": traffic,
                         "flop_per_unit": flop_unit, "units_per_launch": units_local},
        }
        if ktimes_solo is not None and ktimes_solo[0] > 0:
            # `roofline` above follows the contract (events over the timed region, where the kernel shares the GPU with the
            # other stream's recursions); this is the same kernel running alone
            ach = units_local * flop_unit / float(ktimes_solo[0]) / 1e12
            out["kernel_ms_isolated"] = {"score": ktimes_solo[0] * 1e3, "beta": ktimes_solo[1] * 1e3, "alpha_stats": ktimes_solo[2] * 1e3, "mix_stats": ktimes_solo[3] * 1e3}
            out["roofline_isolated"] = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS}
        if args.cpu_seconds > 0 and world == 1:               # the CPU leg runs at N = 1 only
            per_utt = units_local / max(len(s.feats), 1)
            n, cdt = cpu_baseline(s, pk, args.cpu_seconds)
            port = {"value": n * per_utt / cdt, "unit": "frame-state log-lik/s", "cores": 1, "kind": "port",
                    "utterances_per_sec": n / cdt,
                    "sample": "%d utterances of the same shard through oracle/htk_oracle.c (scalar C restatement of "
                              "HFB/HModel, bit-exact vs the reference), %.1f s on one host core" % (n, cdt)}
            ref = cpu_baseline_reference(s, pk, 60)
            if ref is not None:
                out["cpu_baseline"] = {"value": ref[0] * per_utt / ref[1], "unit": "frame-state log-lik/s", "cores": 1, "kind": "reference",
                                       "utterances_per_sec": ref[0] / ref[1],
                                       "sample": "the reference's own HERest (oracle/_ref, one process, one core) over %d utterances of the same "
                                                 "shard; model loading differenced out (run over 2n minus run over n utterances)" % (2 * ref[0])}
                out["cpu_baseline_port"] = port
            else:
                out["cpu_baseline"] = port
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
