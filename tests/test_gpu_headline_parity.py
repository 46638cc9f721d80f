"""The re-estimated model of the HEADLINE workload (bench.py's shard: 5k tied states x 16 mixtures, 1 250 x 500 frames, the arithmetic
mode the bench measures AND the exact mode, accumulate -> htkamd_model_update_device) against the model the reference's HERest writes
from the same files -- north_star's bar: re-estimated mean / variance within 1e-4 relative.

  fixture   tests/golden/c3_herest.npz (made by tests/golden/make_config3_herest_golden.py from oracle/_ref/HERest): a seeded sample of
            128 tied states = 2 048 Gaussians x 39 means and variances, their weights, the transition matrix; the reference's model
            from ONE process and from an 8-way `-p` merge, so that its own run-to-run difference is known per entry
  live      every one of the set's 3.1 M means and variances, where oracle/_ref/HERest is on the box (the driver's GPU box has it)

The bar per entry: |got - ref| <= 1e-4 * scale, scale = |ref| for weights / transition probabilities, max(|ref|, sigma) for means
(SURVEY.md §8c), and for variances the second moment about the previous mean, var + (mean_new - mean_old)^2 -- the quantity HERest's
accumulators hold (tests/c3_herest.py: compare); widened only where the reference's OWN 1-process-vs-8-way difference at that entry is
larger than half of it (one variance in 2.9 M on this workload: tests/golden/c3_herest.npz `whole_set_self`).  Against the variances'
own values: exact mode 1 entry of 2.9 M above 1e-4 (the very entry the reference does not reproduce itself), bf16 x 3 scores 3, fp16 x 2
scores 7, a float64 scorer 6 -- asserted at observed + 1 below (RAW_BARS)."""
import json
import os
import tempfile

import numpy as np
import pytest

import c3_herest as c3

pytestmark = pytest.mark.gpu

MODES = [(0, "exact"), (6, "bf16x3fast"), (34, "fastest")]
# variances beyond 1e-4 of their own value, whole set: (count, worst) for the Gaussians of two frames or more, (count, worst) for the others --
# observed in round 6 (profiles/r06_headline_parity_*.json: exact 1 / 1.31e-4, 0; bf16 x 3 3 / 1.41e-4, 5 / 1.70e-4; fp16 x 2 7 / 1.46e-4, 3 / 1.70e-4) + 1 entry, + ~0.1e-4
RAW_BARS = {"exact": (1, 1.4e-4, 0, 1e-6), "bf16x3fast": (4, 1.5e-4, 6, 1.8e-4), "fastest": (8, 1.6e-4, 4, 1.8e-4)}


def msg_raw(r):
    return "var: %d above 1e-4 of their own value (worst %.3g); under two frames: %d (worst %.3g)" % (
        r["var"]["n_above_1e4"], r["var"]["worst_rel"], r["var_low_occ"]["n_above_1e4"], r["var_low_occ"]["worst_rel"])


def _hip_model(native, s, pk, mode):
    """One EM iteration of the whole shard through the C ABI the way bench.py runs it: pass -> device update."""
    from util import batch_arrays
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    X, frameOff, labOff, labs = batch_arrays(utts)
    model = native.Model(pk)
    dX = native.DevArray(X)
    fb, acc = native.ForwardBackward(model), native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(native.fb_config(scoreMode=mode), acc)
    pr, st = fb.results()
    assert (st == 1).all()
    a = acc.download()
    stats = model.update_device(acc, minEgs=c3.MIN_EGS, minVar=c3.MIN_VAR)
    p = model.get_params()
    return p, a, stats, pr


_cache = {}


def _run(native, mode):
    if "wl" not in _cache:
        _cache["wl"] = c3.workload()
    s, pk = _cache["wl"]
    if mode not in _cache:
        _cache[mode] = _hip_model(native, s, pk, mode)
    return (s, pk) + _cache[mode]


def _assert_report(r, what):
    msg = "%s: %s" % (what, json.dumps(r))
    for k in ("mean", "var", "weight", "mean_low_occ", "var_low_occ"):
        if k in r:
            assert r[k]["n_fail"] == 0, msg
    assert r["trans"]["worst_rel"] <= 1e-4 and r["trans"]["zeros_equal"], msg


@pytest.mark.parametrize("mode,name", MODES, ids=[m[1] for m in MODES])
def test_headline_model_vs_reference_fixture(native, mode, name):
    z = np.load(c3.GOLDEN, allow_pickle=False)
    s, pk, p, a, stats, pr = _run(native, mode)
    # the inputs are the ones the fixture was made from
    assert abs(float(pk["mean"].astype(np.float64).sum()) - float(z["init_mean_sum"])) <= 1e-9 * abs(float(z["init_mean_sum"]))
    assert abs(sum(float(x.astype(np.float64).sum()) for x in s.feats) - float(z["x_sum"])) <= 1e-9 * abs(float(z["x_sum"]))
    # HERest's summary lines
    log1 = str(z["log1"]).splitlines()
    avg = [l for l in log1 if "average log prob" in l][0].split("=")[1].strip()
    got_avg = a["totalPr"] / a["totalT"]
    if mode == 0:
        assert "%e" % got_avg == avg, (got_avg, avg)
    assert abs(got_avg - float(avg)) <= 1e-6 * abs(float(avg)), (got_avg, avg)        # 7 digits printed; utterance log-probabilities to 1e-6 relative
    fl = [l for l in log1 if "floored variance" in l][0].split()
    assert (stats["nFloorVar"], stats["nFloorVarMix"]) == (int(fl[1]), int(fl[6])), (stats, fl)
    assert stats["nSkippedHmm"] == sum("copied: only" in l for l in log1)
    # the sampled states
    g = (z["states"][:, None].astype(np.int64) * c3.M + np.arange(c3.M)[None, :]).reshape(-1)
    got = dict(mean=p["mean"][g], var=p["var"][g], compWeight=p["compWeight"][g], transP=p["transP"])
    r1 = dict(mean=z["mean1"], var=z["var1"], compWeight=z["w1"], transP=z["trans1"])
    r8 = dict(mean=z["mean8"], var=z["var8"], compWeight=z["w8"], transP=z["trans8"])
    r = c3.compare(got, r1, r8, z["occ"].astype(np.float64), init_mean=pk["mean"][g])
    print(name, json.dumps(r))
    _assert_report(r, "sample of %d Gaussians, mode %s" % (g.size, name))
    assert r["var"]["n_above_1e4"] == 0 and r["mean"]["n_above_1e4"] == 0          # on the sample nothing needs the second-moment scale
    # occupancies: the reference's float accumulators against fp64 sums
    lay_occ = a["muOcc"][g]
    assert np.allclose(lay_occ, z["occ"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode,name", MODES, ids=[m[1] for m in MODES])
def test_headline_model_vs_reference_live(native, mode, name):
    """Every entry of the set, against oracle/_ref/HERest run here (one process and 8-way)."""
    if not os.path.exists(os.path.join(c3.REF, "HERest")):
        pytest.skip("oracle/_ref/HERest is not on this box")
    s, pk, p, a, stats, pr = _run(native, mode)
    if "live" not in _cache:
        with tempfile.TemporaryDirectory(prefix="c3herest_") as d:
            c3.write_files(d, s, pk)
            o1, log1, _ = c3.run_reference(d, c3.NU, 1)
            o8, log8, accs = c3.run_reference(d, c3.NU, 8)
            r1, r8 = c3.read_model(os.path.join(o1, "MMF"), pk), c3.read_model(os.path.join(o8, "MMF"), pk)
            vec = c3.load_accs(pk, accs)
        _cache["live"] = (r1, r8, vec)
    r1, r8, vec = _cache["live"]
    lay = native.accs_layout(pk)
    G = int(pk["numGauss"])
    occ = vec[lay.muOcc:lay.muOcc + G]
    r = c3.compare(p, r1, r8, occ, init_mean=pk["mean"])
    print(name, json.dumps(r))
    os.makedirs(os.path.join(c3.ROOT, "gpurun_out"), exist_ok=True)
    json.dump(r, open(os.path.join(c3.ROOT, "gpurun_out", "headline_parity_%s.json" % name), "w"))
    _assert_report(r, "all %d Gaussians, mode %s" % (G, name))
    # Against the variances' OWN values (no second-moment scale): the counts and worst ratios observed on MI355X in round 6, plus one entry /
    # a tenth -- RAW_BARS below.  What they can be: the reference's float arithmetic is itself 2.1e-5 rms away from exact arithmetic per
    # score, and a scorer WITHOUT rounding error (a diagnostic build that scores in float64, tools/r06_parvar.sh `truth`) has 6 entries
    # above 1e-4 (worst 1.81e-4) and 6 among the Gaussians under two frames (2.12e-4): only the bit-identical exact mode can have fewer
    # than a handful, and a tolerance-class mode that counts 3 is inside that scatter, not better than it.
    n_max, worst_max, n_low_max, worst_low_max = RAW_BARS[name]
    assert r["var"]["n_above_1e4"] <= n_max and r["var"]["worst_rel"] <= worst_max and r["mean"]["n_above_1e4"] == 0, msg_raw(r)
    # ... and the 193 089 entries of the Gaussians with fewer than two frames, every one that exists in the reference's model: means all inside
    # 1e-4; variances -- a difference of two sums over one or two frames -- inside 1e-4 of the second moment the accumulators carry (n_fail, above)
    assert r["mean_low_occ"]["n_above_1e4"] == 0 and r["mean_low_occ"]["n"] > 150000
    assert r["var_low_occ"]["n_above_1e4"] <= n_low_max and r["var_low_occ"]["worst_rel"] <= worst_low_max, msg_raw(r)
    # the accumulators themselves: occupancies and weight counts of the whole set
    assert np.allclose(a["muOcc"], occ, rtol=1e-4, atol=1e-4)


def _shard_accs(native, s, pk, mode, utts, wire_round=False):
    """The accumulators of the utterances `utts` of the headline workload under the INITIAL model in scoring mode `mode`."""
    from util import batch_arrays
    X, frameOff, labOff, labs = batch_arrays([dict(seq=s.seqs[u], feat=s.feats[u]) for u in utts])
    model = native.Model(pk)
    dX = native.DevArray(X)
    fb, acc = native.ForwardBackward(model), native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(native.fb_config(scoreMode=mode), acc)
    pr, st = fb.results()
    assert (st == 1).all()
    if wire_round:
        acc.wire_round()
    return acc.download(), model, acc


def _acc_deviation(a, ref):
    """bench.py's rule (oracle_check): counts against max(|ref|, 1e-3); first- and second-order sums against max(|ref|, occupancy), the
    Gaussians under three frames of occupancy apart."""
    worst = {}
    for k in ("muOcc", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
        r = np.asarray(ref[k], np.float64).reshape(-1)
        worst[k] = float(np.max(np.abs(np.asarray(a[k], np.float64).reshape(-1) - r) / np.maximum(np.abs(r), 1e-3)))
    occ = np.maximum(np.asarray(ref["muOcc"], np.float64), 1e-3)[:, None]
    few = occ[:, 0] < 3.0
    worst_few = {}
    for k in ("mu", "va"):
        r = np.asarray(ref[k], np.float64).reshape(occ.shape[0], -1)
        rel = np.abs(np.asarray(a[k], np.float64).reshape(r.shape) - r) / np.maximum(np.abs(r), occ)
        worst[k] = float(np.max(rel[~few])) if (~few).any() else 0.0
        worst_few[k] = float(np.max(rel[few])) if few.any() else 0.0
    return worst, worst_few, int(few.sum())


@pytest.mark.parametrize("mode,name", MODES[1:], ids=[m[1] for m in MODES[1:]])
@pytest.mark.parametrize("nu", [400, 800])
def test_tolerance_modes_at_smaller_shards(native, mode, name, nu):
    """The tolerance-class scoring modes are not accepted at the one sample size where they happen to pass: the accumulators of the
    first 400 and the first 800 utterances of the shard (fewer frames per Gaussian: less averaging of a score's deviation) against the
    exact mode's (= the oracle's to float accumulation noise, tests/test_gpu_parity.py), by bench.py's rule -- counts and the sums of
    Gaussians with three frames or more at 1e-4, the sums of the Gaussians under three frames at 2e-4."""
    if "wl" not in _cache:
        _cache["wl"] = c3.workload()
    s, pk = _cache["wl"]
    key = ("exact_acc", nu)
    if key not in _cache:
        _cache[key] = _shard_accs(native, s, pk, 0, range(nu))[0]
    ref = _cache[key]
    a = _shard_accs(native, s, pk, mode, range(nu))[0]
    worst, worst_few, n_few = _acc_deviation(a, ref)
    print(name, nu, json.dumps(dict(worst=worst, under_3_frames=worst_few, gaussians_under_3=n_few)))
    os.makedirs(os.path.join(c3.ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(worst=worst, under_3_frames=worst_few, gaussians_under_3=n_few), open(os.path.join(c3.ROOT, "gpurun_out", "shard_parity_%s_%d.json" % (name, nu)), "w"))
    assert max(worst.values()) <= 1e-4, (worst, worst_few)
    assert max(worst_few.values()) <= 2e-4, worst_few


def test_fp32_wire_eight_way_merge_vs_reference_fixture(native):
    """The multi-GPU exchange with fp32 on the wire (htkamd_accs_allreduce_wire HTKAMD_WIRE_F32, bench.py --wire f32), emulated on one
    device: the shard cut 8-way by utterance id (HERest -p semantics), every part's statistics rounded to float once
    (htkamd_accs_wire_round), the parts added in float, the device update on the sum -- against the reference's model (fixture)."""
    z = np.load(c3.GOLDEN, allow_pickle=False)
    if "wl" not in _cache:
        _cache["wl"] = c3.workload()
    s, pk = _cache["wl"]
    mode = 34
    total, model, acc = None, None, None
    for r in range(8):
        a, model, acc = _shard_accs(native, s, pk, mode, range(r, c3.NU, 8), wire_round=True)
        v = a["vec"].copy()
        bulk = int(acc.lay.nEgs)
        if total is None:
            total = v
            total32 = v[:bulk].astype(np.float32)
        else:
            total32 = (total32 + v[:bulk].astype(np.float32)).astype(np.float32)         # the ring adds floats
            total[bulk:] += v[bulk:]
    total[:bulk] = total32.astype(np.float64)
    acc.zero(None)
    acc.upload_add(total)
    stats = model.update_device(acc, minEgs=c3.MIN_EGS, minVar=c3.MIN_VAR)
    p = model.get_params()
    g = (z["states"][:, None].astype(np.int64) * c3.M + np.arange(c3.M)[None, :]).reshape(-1)
    got = dict(mean=p["mean"][g], var=p["var"][g], compWeight=p["compWeight"][g], transP=p["transP"])
    r1 = dict(mean=z["mean1"], var=z["var1"], compWeight=z["w1"], transP=z["trans1"])
    r8 = dict(mean=z["mean8"], var=z["var8"], compWeight=z["w8"], transP=z["trans8"])
    r = c3.compare(got, r1, r8, z["occ"].astype(np.float64), init_mean=pk["mean"][g])
    print("wire f32, 8-way", json.dumps(r))
    _assert_report(r, "sample, fp32 wire, 8-way")
    # and next to the one-batch model of the same mode: what the float wire itself moves (means and weights to float rounding of the
    # sums; a variance is the difference of two sums, so it is held to its second moment about the old mean, as in c3.compare)
    s_, pk_, p1, a1, st1, pr1 = _run(native, mode)
    for k in ("mean", "compWeight"):
        d = np.abs(p[k].astype(np.float64) - p1[k].astype(np.float64)) / np.maximum(np.abs(p1[k].astype(np.float64)), 1.0 if k == "mean" else 1e-3)
        assert float(d.max()) <= 2e-5, (k, float(d.max()))
    m2 = p1["var"].astype(np.float64) + (p1["mean"].astype(np.float64) - pk["mean"].astype(np.float64)) ** 2
    d = np.abs(p["var"].astype(np.float64) - p1["var"].astype(np.float64)) / m2
    assert float(d.max()) <= 2e-5, ("var", float(d.max()))
