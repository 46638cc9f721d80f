"""The multi-rank path on CPU: two processes (gloo), utterance shards, one all-reduce of the accumulator vector,
identical update on every rank -- the same code bench.py / the drivers use with backend nccl (= RCCL) on the GPUs.
Per-rank statistics come from the oracle here (there is no GPU in this container); what is under test is the
sharding, the vector layout, the collective and that the merged result equals the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q, wire="f64", nutt=9, parts=0):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from htk_amd import herest, synth
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = synth.generate(30, 3, 20, nutt, 90, 55)
    pk = s.packed()
    m = po.Model(pk)
    lay = herest.layout_from_packed(pk)
    mine = herest.shard_indices(len(s.feats), rank, world)
    acc = po.Accs(m); cfg = po.fb_cfg()
    tot_pr, tot_t, done = 0.0, 0, 0
    for u in mine:
        rc, pr, _ = po.fb_utt(m, cfg, s.feats[u], s.seqs[u], acc)
        if rc == 1:
            tot_pr += pr; tot_t += s.feats[u].shape[0]; done += 1
    vec = herest.pack_vector(lay, acc, tot_pr, tot_t, done)
    t = torch.from_numpy(vec)
    if parts:
        herest.all_reduce_accumulators_in_parts(t, pk, lay, parts, wire=wire)      # the same exchange in `parts` parts by tied state (bench.py --exchange-slices)
    else:
        herest.all_reduce_accumulators(t, wire=wire, bulk=lay["nEgs"])      # dist.all_reduce(SUM) on the flat vector (fp64, or floats on the wire)
    q.put((rank, t.numpy().copy(), list(mine)))
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, v0, s0), (r1, v1, s1) = res
    assert np.array_equal(v0, v1)                           # every rank holds the same sum
    assert sorted(s0 + s1) == list(range(9)) and not set(s0) & set(s1)
    # single process reference
    sys.path.insert(0, ROOT)
    from htk_amd import herest, synth
    from oracle import pyoracle as po
    s = synth.generate(30, 3, 20, 9, 90, 55)
    pk = s.packed(); m = po.Model(pk); acc = po.Accs(m); cfg = po.fb_cfg()
    tot_pr, tot_t = 0.0, 0
    for u in range(9):
        rc, pr, _ = po.fb_utt(m, cfg, s.feats[u], s.seqs[u], acc)
        tot_pr += pr; tot_t += s.feats[u].shape[0]
    lay = herest.layout_from_packed(pk)
    ref = herest.pack_vector(lay, acc, tot_pr, tot_t, 9)
    # float accumulators summed in a different order: the reference's own -p N merge differs at this level too
    assert np.allclose(v0, ref, rtol=2e-5, atol=1e-5)
    assert v0[lay["nUttDone"]] == 9 and v0[lay["totalT"]] == tot_t
    assert np.array_equal(v0[lay["nEgs"]:lay["nEgs"] + 20], acc.nEgs.astype(np.float64))


def test_eight_rank_fp32_wire_equals_single_process():
    """bench.py --wire f32 / htkamd_accs_allreduce_wire(HTKAMD_WIRE_F32): eight ranks, every rank's fp64 statistics rounded to float once,
    floats summed, counters in fp64 -- against the single process's sums: within float rounding of the partial sums (a few 1e-7), far
    inside the 1e-4 the re-estimated parameters are held to; the counters exactly."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    NU = 24
    ps = [ctx.Process(target=_worker, args=(r, 8, port, q, "f32", NU)) for r in range(8)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    v0 = res[0][1]
    for _, v, _ in res[1:]:
        assert np.array_equal(v0, v)                        # every rank holds the same sum
    sys.path.insert(0, ROOT)
    from htk_amd import herest, synth
    from oracle import pyoracle as po
    s = synth.generate(30, 3, 20, NU, 90, 55)
    pk = s.packed(); m = po.Model(pk); lay = herest.layout_from_packed(pk)
    # the same partial sums in fp64, added in fp64: what the float wire is held against
    ref = np.zeros(lay["total"])
    for r in range(8):
        acc = po.Accs(m); cfg = po.fb_cfg()
        tot_pr, tot_t, done = 0.0, 0, 0
        for u in herest.shard_indices(NU, r, 8):
            rc, pr, _ = po.fb_utt(m, cfg, s.feats[u], s.seqs[u], acc)
            if rc == 1:
                tot_pr += pr; tot_t += s.feats[u].shape[0]; done += 1
        ref += herest.pack_vector(lay, acc, tot_pr, tot_t, done)
    bulk = lay["nEgs"]
    ipr = lay["totalPr"] - bulk                             # nEgs, totalT, counters: integers in fp64, exact; totalPr: an fp64 sum in ring order
    assert np.array_equal(np.delete(v0[bulk:], ipr), np.delete(ref[bulk:], ipr))
    assert abs(v0[lay["totalPr"]] - ref[lay["totalPr"]]) <= 1e-12 * abs(ref[lay["totalPr"]])
    assert v0[lay["nUttDone"]] == NU
    scale = np.maximum(np.abs(ref[:bulk]), 1.0)
    assert np.max(np.abs(v0[:bulk] - ref[:bulk]) / scale) <= 1e-6
    assert np.all(v0[:bulk] == v0[:bulk].astype(np.float32))  # what came back are floats


@pytest.mark.parametrize("wire", ["f64", "f32"])
def test_exchange_in_parts_equals_the_whole_exchange_two_ranks(wire):
    """bench.py --exchange-slices on the host side: the accumulator vector travels in four parts by tied state (herest.state_ranges, the mirror of
    htkamd_accs_state_ranges) -- every rank ends with the vector ONE all-reduce leaves, bit for bit on either wire (two ranks: a sum of two is the same in either order);
    the parts are disjoint and cover the statistics' part of the vector."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from htk_amd import herest, synth
    ctx = mp.get_context("spawn")
    out = {}
    for parts in (0, 4):
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, q, wire, 9, parts)) for r in range(2)]
        for p in ps:
            p.start()
        res = sorted([q.get(timeout=120) for _ in ps], key=lambda x: x[0])
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
        assert np.array_equal(res[0][1], res[1][1])
        out[parts] = res[0][1]
    assert np.array_equal(out[0], out[4])
    pk = synth.generate(30, 3, 20, 1, 20, 55).packed()
    lay = herest.layout_from_packed(pk)
    seen = np.zeros(lay["total"], np.int32)
    S = int(pk["numStates"])
    for i in range(4):
        for o, n in herest.state_ranges(pk, lay, S * i // 4, S * (i + 1) // 4, with_rest=(i == 3)):
            seen[o:o + n] += 1
    assert (seen[:lay["nEgs"]] == 1).all() and not seen[lay["nEgs"]:].any()


def test_shards_are_balanced_and_disjoint():
    sys.path.insert(0, ROOT)
    from htk_amd import herest
    for n, w in ((10000, 8), (7, 8), (9, 2), (0, 4)):
        parts = [herest.shard_indices(n, r, w) for r in range(w)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
