"""The collective of the HERest pass on the real backend: torch.distributed with backend nccl (= RCCL) in a one-rank group on the
GPU box (8-GPU runs are the driver's), on the zero-copy tensor view of an htkamd accumulator vector and on a side stream, the way
bench.py issues it.  With one rank the sum must leave the vector as it was; what this guards is that RCCL accepts the library's
own device allocation and the stream/event ordering, not the arithmetic (tests/test_dist_gloo.py covers that with two ranks)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_all_reduce_on_accumulator_vector():
    """Runs in a process of its own with torch imported first, as bench.py does: the HIP runtime torch ships and the one the
    library links must be initialised in that order for torch to see the device."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def _main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    import sys
    sys.path.insert(0, sys_path); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from htk_amd import capi as native, herest, synth
    from util import batch_arrays
    s = synth.generate(12, 2, 6, 3, 60, 5)
    pk = s.packed()
    model = native.Model(pk)
    acc = native.Accs(model)
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = native.DevArray(X)
    fb = native.ForwardBackward(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(native.fb_config(), acc)
    before = acc.download()["vec"].copy()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = herest.device_vector_as_tensor(acc, 0)
        assert t.dtype == torch.float64 and t.numel() == before.size
        # the tensor must BE the library's accumulator vector, not a copy of it: same address, and a write through either side is
        # seen by the other (a silent copy would leave every rank updating from its own un-reduced statistics)
        assert t.data_ptr() == acc.device_vector()[0]
        t.add_(1.0); torch.cuda.synchronize()
        assert np.array_equal(acc.download()["vec"], before + 1.0)
        acc.upload_add(-np.ones_like(before))
        assert np.array_equal(t.cpu().numpy(), before)
        comm = torch.cuda.Stream()
        done = torch.cuda.Event(); done.record(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            comm.wait_event(done)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)             # what herest.all_reduce_accumulators does for world > 1
        comm.synchronize()
        assert np.array_equal(acc.download()["vec"], before)
        assert np.array_equal(t.cpu().numpy(), before)
    finally:
        dist.destroy_process_group()
    print("RCCL_OK")


def test_two_ranks_with_hip_statistics_merge_to_the_single_rank_result():
    """Two processes share the one GPU of the box (RCCL refuses two ranks on one device, so the exchange itself goes through gloo on
    host copies -- the collective on the device vector is the test above): each rank runs ITS shard of the utterances through the HIP
    path, the accumulator vectors are summed across the ranks and added back, and every rank updates its model on the device.
    The merged statistics and the updated parameters must equal those of one process that saw all the utterances."""
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "shard", str(r), "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for r in range(2)]
    outs = [p.communicate(timeout=900) for p in ps]
    for p, (o, e) in zip(ps, outs):
        assert p.returncode == 0 and "SHARD_OK" in o, (o[-1500:], e[-3000:])


def _shard_main(rank, world):
    import torch
    import torch.distributed as dist
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from htk_amd import capi as native, herest, synth
    from util import batch_arrays, acc_close
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    s = synth.generate(40, 4, 24, 11, 100, 21)
    pk = s.packed()
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]

    def run(sel):
        model = native.Model(pk); acc = native.Accs(model)
        X, frameOff, labOff, labs = batch_arrays([utts[u] for u in sel])
        dX = native.DevArray(X)
        fb = native.ForwardBackward(model)
        fb.prepare(dX.ptr.value, frameOff, labOff, labs)
        fb.execute(native.fb_config(), acc)
        return model, acc, fb.results()[0], dX

    mine = list(herest.shard_indices(len(utts), rank, world))
    model, acc, pr, keep = run(mine)
    local = acc.download()["vec"].copy()
    t = torch.from_numpy(local.copy())
    herest.all_reduce_accumulators(t)                                  # sum over the ranks (gloo here, RCCL in bench.py)
    merged = t.numpy()
    acc.zero(); acc.upload_add(merged)                                 # LoadAccs of every rank's dump
    st = model.update_device(acc, minEgs=1, minVar=0.01)
    p = model.get_params()
    # the single-process run over all utterances
    model1, acc1, pr1, keep1 = run(list(range(len(utts))))
    whole = acc1.download()
    lay = herest.layout_from_packed(pk)
    assert merged[lay["nUttDone"]] == len(utts) and merged[lay["totalT"]] == whole["totalT"]
    assert np.array_equal(merged[lay["nEgs"]:lay["nEgs"] + int(pk["numPhys"])], whole["nEgs"])
    acc_close(merged[:lay["nEgs"]], whole["vec"][:lay["nEgs"]], "merged statistics", rtol=1e-9, floor=1e-6)    # same fp64 sums, another order
    assert np.array_equal(pr, pr1[mine])                               # an utterance's probability does not depend on its shard
    st1 = model1.update_device(acc1, minEgs=1, minVar=0.01)
    p1 = model1.get_params()
    assert st == st1
    for k in ("mean", "var", "compWeight", "transP"):
        assert np.allclose(p[k], p1[k], rtol=2e-6, atol=1e-7), k
    dist.destroy_process_group()
    print("SHARD_OK")


if __name__ == "__main__":
    import sys as _sys
    if len(_sys.argv) >= 4 and _sys.argv[1] == "shard":
        _shard_main(int(_sys.argv[2]), int(_sys.argv[3]))
    else:
        _main()
