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


if __name__ == "__main__":
    _main()
