"""Host-side mirror of HERest's pass structure for the multi-GPU path (HERest.c:502-557).

    rank r of R  ==  `HERest -p r+1`:  its shard of the script file, its own accumulator set
    all-reduce   ==  `HERest -p 0 HER*.acc`: LoadAccs adds every dump (HTrain.c:1625), then UpdateModels

The accumulators of a rank live in ONE flat fp64 device vector (include/htk_amd.h, htkamd_accs_layout), so the
merge is a single sum all-reduce (RCCL over xGMI when the backend is nccl; gloo in the CPU tests).  No numerics
here: statistics come from the HIP kernels (capi.ForwardBackward), the update from htk_amd/host/update.c.
"""
from __future__ import annotations

import numpy as np


def shard_indices(n_utts: int, rank: int, world: int) -> range:
    """Round-robin split of the script file, as one would split train.scp for `-p 1..R` (HTKBook train.tex:618-660)."""
    return range(rank, n_utts, world)


def layout_from_packed(pk: dict) -> dict:
    """Offsets of the flat accumulator vector; identical to htkamd_accs_get_layout (csrc/model.hip)."""
    G, D, C, S, H = int(pk["numGauss"]), int(pk["vecSize"]), int(pk["numComp"]), int(pk["numStates"]), int(pk["numPhys"])
    o = 0
    lay = {}
    for name, n in (("mu", G * D), ("muOcc", G), ("va", G * D), ("vaOcc", G), ("wt", C), ("wtOcc", S),
                    ("tr", int(pk["transOff"][-1])), ("trOcc", int(np.sum(pk["transN"]))), ("nEgs", H),
                    ("totalPr", 1), ("totalT", 1), ("nUttDone", 1), ("nUttSkipped", 1), ("nEval", 1)):
        lay[name] = o
        o += n
    lay["total"] = o
    return lay


def pack_vector(lay: dict, acc, total_pr: float, total_t: int, n_done: int, n_skipped: int = 0, n_eval: int = 0) -> np.ndarray:
    """Flat fp64 vector from per-field arrays (an object with .mu .muOcc .va .vaOcc .wt .wtOcc .tr .trOcc .nEgs)."""
    v = np.zeros(lay["total"], np.float64)
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs"):
        a = np.asarray(getattr(acc, k), np.float64).reshape(-1)
        v[lay[k]:lay[k] + a.size] = a
    v[lay["totalPr"]] = total_pr; v[lay["totalT"]] = total_t
    v[lay["nUttDone"]] = n_done; v[lay["nUttSkipped"]] = n_skipped; v[lay["nEval"]] = n_eval
    return v


def all_reduce_accumulators(vec_tensor, wire: str = "f64", bulk: int | None = None, staging=None) -> None:
    """The one exchange step of a pass: sum the accumulator vectors of all ranks in place.

    wire = "f32" (htkamd_accs_allreduce_wire / HTKAMD_WIRE_F32 in the C ABI): the statistics -- the first `bulk` = layout.nEgs entries --
    travel as floats: every rank rounds its fp64 partial sums once, the ring adds floats, the sums return to the fp64 vector; the
    counters behind them (nEgs, totalPr, totalT, ...) stay fp64 in a second, small all-reduce.  Half the bytes of the fp64 exchange and
    still tighter than the reference, which merges float dumps (LoadAccs, HTrain.c:1625-1687).  `staging`: a float32 tensor of `bulk`
    entries to reuse."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    if wire == "f64":
        dist.all_reduce(vec_tensor, op=dist.ReduceOp.SUM)
        return
    assert wire == "f32" and bulk is not None and 0 < bulk <= vec_tensor.numel()
    w = staging if staging is not None else torch.empty(bulk, dtype=torch.float32, device=vec_tensor.device)
    w.copy_(vec_tensor[:bulk])
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    if bulk < vec_tensor.numel():
        dist.all_reduce(vec_tensor[bulk:], op=dist.ReduceOp.SUM)
    vec_tensor[:bulk].copy_(w)


def state_ranges(pk: dict, lay: dict, state0: int, state1: int, with_rest: bool = False):
    """[(offset, length)] of the accumulator vector's ranges that belong to tied states [state0, state1): mu, muOcc, va, vaOcc of their Gaussians, wt of their components,
    wtOcc -- and with `with_rest` what no state owns and travels as statistics (tr, trOcc).  The host mirror of htkamd_accs_state_ranges (csrc/comm.hip): sets whose
    components own their Gaussians in state order only (compGauss[c] == c)."""
    cg = np.asarray(pk["compGauss"])
    if int(pk["numGauss"]) != int(pk["numComp"]) or not np.array_equal(cg, np.arange(cg.size)) or int(pk.get("numStreams", 1) or 1) > 1:
        raise ValueError("the set's components do not own their Gaussians in state order: exchange the vector whole")
    off = np.asarray(pk["stateCompOff"])
    g0, g1, D = int(off[state0]), int(off[state1]), int(pk["vecSize"])
    out = [(lay["mu"] + g0 * D, (g1 - g0) * D), (lay["muOcc"] + g0, g1 - g0), (lay["va"] + g0 * D, (g1 - g0) * D), (lay["vaOcc"] + g0, g1 - g0),
           (lay["wt"] + g0, g1 - g0), (lay["wtOcc"] + state0, state1 - state0)]
    if with_rest:
        out.append((lay["tr"], lay["nEgs"] - lay["tr"]))
    return [(o, n) for o, n in out if n > 0]


def all_reduce_accumulators_in_parts(vec_tensor, pk: dict, lay: dict, n_parts: int, wire: str = "f32") -> None:
    """The exchange of all_reduce_accumulators cut into `n_parts` by tied state (bench.py --exchange-slices; on the device the parts are packed by
    htkamd_accs_pack_ranges and travel while the next range of states is still being summed): part i = the ranges of states [S i / n, S (i + 1) / n), the last part with
    tr / trOcc and the fp64 counters behind them.  Leaves the vector the whole exchange leaves."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    S = int(pk["numStates"])
    dt = torch.float32 if wire == "f32" else torch.float64
    for i in range(n_parts):
        rg = state_ranges(pk, lay, S * i // n_parts, S * (i + 1) // n_parts, with_rest=(i == n_parts - 1))
        w = torch.cat([vec_tensor[o:o + n] for o, n in rg]).to(dt)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        at = 0
        for o, n in rg:
            vec_tensor[o:o + n] = w[at:at + n].to(vec_tensor.dtype); at += n
    if lay["nEgs"] < vec_tensor.numel():
        dist.all_reduce(vec_tensor[lay["nEgs"]:], op=dist.ReduceOp.SUM)


def device_vector_as_tensor(accs, device_index: int):
    """Zero-copy torch view of an htkamd_accs device vector (for the collective)."""
    import torch
    ptr, n = accs.device_vector()

    class _W:
        pass
    w = _W()
    w.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(w, device=torch.device("cuda", device_index))
