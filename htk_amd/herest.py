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


def device_vector_as_tensor(accs, device_index: int):
    """Zero-copy torch view of an htkamd_accs device vector (for the collective)."""
    import torch
    ptr, n = accs.device_vector()

    class _W:
        pass
    w = _W()
    w.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(w, device=torch.device("cuda", device_index))
