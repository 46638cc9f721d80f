"""Parity of the HIP path (through the C ABI) against the reference's golden vectors and the oracle.
Runs on the MI355X box:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

from util import acc_close, batch_arrays, load_case, mmf_fmt

pytestmark = pytest.mark.gpu

CASES = ["fb_small", "fb_small_prune", "fb_topo", "fb_topo_prune"]


PATHS = ["lr", "state", "wave", "general"]  # left-to-right chains on fb_lr.hip where the set allows (else as "state") | lane per chain state (fb_state.hip) | lane per model (fb_wave.hip) | workgroup per utterance


def run_fb(native, pk, utts, prune=None, debug=True, general=False, uFlags=15, scoreMode=0, path=None, stats_list="auto"):
    model = native.Model(pk)
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = native.DevArray(X)
    if path is not None:
        general = path == "general"
    fb = native.ForwardBackward(model, debug=debug, force_general=general, no_state_path=(path == "wave"), stats_list=stats_list, no_lr_path=(path == "state"))
    acc = native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(native.fb_config(uFlags=uFlags, scoreMode=scoreMode, **(prune or {})), acc)
    pr, st = fb.results()
    return model, fb, acc, pr, st


# ----------------------------------------------------------------------------------------- scoring (K1)
@pytest.mark.parametrize("name", ["fb_small", "fb_topo"])
def test_outp_block_bit_exact(native, oracle, name):
    """IDOutP + ShStrP/cSOutP arithmetic: every float identical to the oracle (which is bit-exact vs the reference)."""
    case = load_case(name)
    gm, om = native.Model(case["pk"]), oracle.Model(case["pk"])
    X = np.concatenate([u["feat"] for u in case["utts"]])
    states = np.arange(case["pk"]["numStates"], dtype=np.int32)
    got, ref = gm.outp_block(X, states), om.score_block(X, states)
    assert np.array_equal(got, ref)
    prep = gm.get_prepared()
    assert np.array_equal(prep["ivar"], om.ivar) and np.array_equal(prep["gconst"], om.gconst)
    assert np.array_equal(prep["compLogWt"], om.compLogWt)


@pytest.mark.parametrize("D,M", [(20, 3), (26, 2), (39, 16), (7, 1)])
def test_outp_block_other_sizes(native, oracle, D, M):
    """D = 26/39 use the register-resident kernels, D = 20/7 the any-D kernel; M = 1 skips the LAdd; ragged T."""
    from htk_amd import synth
    s = synth.generate(25, M, 10, 1, 333, 40 + D, D=D)
    pk = s.packed()
    gm, om = native.Model(pk), oracle.Model(pk)
    X = s.feats[0]
    states = np.array([3, 3, 0, 24, 7, 11], np.int32)          # repeats and arbitrary order are allowed
    assert np.array_equal(gm.outp_block(X, states), om.score_block(X, states))
    assert gm.outp_block(X[:0], states).shape == (0, 6)
    assert np.array_equal(gm.outp_block(X[:1], states), om.score_block(X[:1], states))


@pytest.mark.parametrize("D,M", [(39, 16), (26, 3), (13, 1), (20, 5)])
def test_outp_block_soutp_and_diagc_forms_bit_exact(native, oracle, D, M):
    """The forms a direct caller of OutP / POutP / SOutP / MOutP sees (HRest, HInit): SOutP's rounding (HModel.c:5538-5552: double
    log-sum, one float at the end) and DOutP's xmm*xmm/var for sets that were not put through ConvDiagC (HModel.c:5347) -- every float
    identical to the oracle's restatement, and the SOutP form does differ from ShStrP's in a few per cent of the mixture scores."""
    from htk_amd import synth, capi
    s = synth.generate(25, M, 10, 1, 200, 70 + D + M, D=D)
    pk = s.packed()
    gm, om = native.Model(pk), oracle.Model(pk)
    X = s.feats[0]
    states = np.arange(25, dtype=np.int32)
    base = gm.outp_block(X, states, mode=0)
    so = gm.outp_block(X, states, mode=capi.SCORE_SOUTP)
    assert np.array_equal(so, om.soutp_block(X, states))
    assert np.array_equal(gm.outp_block(X, states, mode=capi.SCORE_SOUTP | capi.SCORE_DIAGC), om.soutp_block(X, states, diagc=True))
    assert np.array_equal(gm.outp_block(X, states, mode=capi.SCORE_DIAGC), om.score_block_diagc(X, states))
    if M > 1:
        frac = float((so != base).mean())
        assert 0.0 < frac < 0.3, frac
    else:
        assert np.array_equal(so, base)


def test_outp_large_block_statistics(native, oracle):
    """BASELINE config-2 shape (1k states x 8 mix, 2 x 500 frames x all states = 1M scores): bit-exact on a sample of rows,
    and every score finite and negative."""
    from htk_amd import synth
    s = synth.generate(1000, 8, 2000, 2, 500, 1)
    pk = s.packed()
    gm, om = native.Model(pk), oracle.Model(pk)
    X = np.concatenate(s.feats)
    states = np.arange(1000, dtype=np.int32)
    got = gm.outp_block(X, states)
    assert np.isfinite(got).all() and (got < 0).all()
    rows = np.arange(0, 1000, 37)
    assert np.array_equal(got[rows], om.score_block(X[rows], states))


# ----------------------------------------------------------------------------------------- scoring on the matrix cores (K1m)
@pytest.mark.parametrize("mode", [1, 4, 32], ids=["f32mfma", "bf16x3", "f16x2"])
@pytest.mark.parametrize("D,M", [(39, 16), (39, 20), (39, 5), (26, 2), (13, 1), (13, 33), (36, 8), (20, 3), (7, 2), (40, 4), (45, 3), (16, 6), (17, 2)])
def test_outp_block_mfma_tolerance(native, oracle, D, M, mode):
    """HTKAMD_SCORE_MFMA: expanded-form fp32 GEMM + float log-sum-exp.  Tolerance class: |score - reference| <= 1e-3
    absolute (scores are O(100); the reference's own float rounding is ~1e-4), typical error far smaller.
    M = 20/33 span two/three column tiles, M = 5/2/1 leave unused columns; T = 150 exercises a ragged second pass."""
    from htk_amd import synth
    if mode == 1 and D > 40:
        pytest.skip("the fp32 matrix-core kernel takes vector sizes up to 40")
    s = synth.generate(25, M, 10, 1, 150, 900 + D + M, D=D)
    pk = s.packed()
    gm, om = native.Model(pk), oracle.Model(pk)
    X = s.feats[0]
    states = np.arange(25, dtype=np.int32)[::-1].copy()
    got, ref = gm.outp_block(X, states, mode=mode), om.score_block(X, states)
    assert got.shape == ref.shape and np.isfinite(got).all()
    err = np.abs(got.astype(np.float64) - ref)
    assert err.max() <= 1e-3, err.max()
    assert err.mean() <= 1e-4, err.mean()
    assert np.array_equal(gm.outp_block(X, states, mode=0), ref)          # the exact mode is untouched
    for T in (1, 64, 65):
        g1 = gm.outp_block(X[:T], states, mode=mode)
        assert np.abs(g1 - ref[:T]).max() <= 1e-3


def test_f16_path_detects_what_it_cannot_represent(native, oracle):
    """HTKAMD_SCORE_F16 scales every term by a power of two chosen from the model; a feature value far outside anything the model
    describes (here 3e4 in one dimension of one frame: x^2 = 9e8, scaled beyond 65504) must raise HTKAMD_ERANGE -- from
    htkamd_model_f16_check for block scoring, from htkamd_fb_results for a forward-backward pass -- and never come out as a wrong
    score; the same data through HTKAMD_SCORE_BF16 is within tolerance, and ordinary data afterwards passes again."""
    from htk_amd import synth
    s = synth.generate(25, 4, 10, 3, 60, 77, D=39)
    pk = s.packed()
    gm, om = native.Model(pk), oracle.Model(pk)
    states = np.arange(25, dtype=np.int32)
    X = s.feats[0].copy()
    ok = gm.outp_block(X, states, mode=32)
    assert np.abs(ok - om.score_block(X, states)).max() <= 1e-3
    Xb = X.copy(); Xb[17, 5] = 3.0e4
    with pytest.raises(native.HtkAmdError) as ei:
        gm.outp_block(Xb, states, mode=32)
    assert ei.value.rc == native.ERANGE and "feature" in str(ei.value)
    ref = om.score_block(Xb, states)
    got = gm.outp_block(Xb, states, mode=4)
    assert np.abs(got - ref)[np.arange(len(Xb)) != 17].max() <= 1e-3
    assert np.allclose(got[17], ref[17], rtol=1e-5)
    assert np.abs(gm.outp_block(X, states, mode=32) - om.score_block(X, states)).max() <= 1e-3      # the flag was cleared
    # a model whose coefficients span more than the format (one variance of 1e-10 among ordinary ones): flagged when the table is built
    pk2 = dict(pk); pk2["var"] = pk["var"].copy(); pk2["var"][3, 4] = 1.0e-10
    gm2 = native.Model(pk2)
    with pytest.raises(native.HtkAmdError) as ei:
        gm2.outp_block(X, states, mode=32)
    assert ei.value.rc == native.ERANGE and "model coefficient" in str(ei.value)
    assert np.isfinite(gm2.outp_block(X, states, mode=4)).all()
    # forward-backward: the pass's own flag
    feats = [f.copy() for f in s.feats]
    feats[1][3, 0] = -5.0e4
    fb = native.ForwardBackward(gm); acc = native.Accs(gm)
    Xall = np.concatenate(feats)
    frameOff = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.int32)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
    labs = np.concatenate(s.seqs).astype(np.int32)
    dX = native.DevArray(Xall)
    for mode, bad in ((34, True), (6, False)):
        acc.zero(None); fb.prepare(dX.ptr.value, frameOff, labOff, labs, None); fb.execute(native.fb_config(scoreMode=mode), acc, None)
        if bad:
            with pytest.raises(native.HtkAmdError) as ei:
                fb.results(None)
            assert ei.value.rc == native.ERANGE
        else:
            pr, st = fb.results(None)
            assert (st == 1).all()


def test_outp_block_mfma_rejects_other_sizes(native):
    from htk_amd import synth
    s = synth.generate(5, 2, 4, 1, 20, 5, D=45)
    gm = native.Model(s.packed())
    with pytest.raises(native.HtkAmdError):
        gm.outp_block(s.feats[0], np.arange(5, dtype=np.int32), mode=1)
    with pytest.raises(native.HtkAmdError):
        gm.outp_block(s.feats[0], np.arange(5, dtype=np.int32), mode=7)


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 6, 32, 34], ids=["mfma", "fastladd", "fast", "bf16x3", "bf16x3fast", "f16x2", "fastest"])
@pytest.mark.parametrize("name", ["fb_small", "fb_topo", "fb_small_prune", "fb_topo_prune"])
def test_mfma_forward_backward_within_tolerance(native, name, mode):
    """HERest through the tolerance-class kernels (matrix-core scores and / or the fp32-transcendental LAdd of the recursions):
    utterance log-probabilities within 1e-6 relative, alpha/beta within 1e-4 relative, and the parameters re-estimated from
    its accumulators within 1e-4 of the MMF the reference's HERest wrote (BASELINE.json north_star tolerance)."""
    case = load_case(name)
    model, fb, acc, pr, st = run_fb(native, case["pk"], case["utts"], case["prune"], scoreMode=mode)
    for u, ut in enumerate(case["utts"]):
        assert st[u] == 1
        assert abs(pr[u] - float(ut["pr"])) <= 1e-6 * abs(float(ut["pr"]))
        g = fb.trellis(u)
        if "beta" in ut:
            for k in ("beta", "alpha"):
                ref, got = ut[k], g[k]
                ok = ~np.isnan(ref) & (ref > -1e9) & ~np.isnan(got)
                assert np.allclose(got[ok], ref[ok], rtol=1e-4, atol=0), k
    a = acc.download()
    ref = case["acc"]
    for k in ("muOcc", "wtOcc", "trOcc"):
        assert np.allclose(a[k], ref[k], rtol=1e-4, atol=1e-6), k
    model.update(acc, a["vec"], minEgs=1, singleProcess=True)
    p = model.get_params()
    upd = case["upd"]
    ref_mean = np.asarray(upd["mean"], np.float64); ref_var = np.asarray(upd["var"], np.float64)
    ok = ~np.isnan(ref_mean)
    sigma = np.sqrt(np.where(ok, np.abs(ref_var), 1.0))
    assert (np.abs(p["mean"] - ref_mean)[ok] <= 1e-4 * np.maximum(np.abs(ref_mean), sigma)[ok] + 1e-6).all()
    # Two groups, both held to 1e-4 (as tests/c3_herest.py: compare does at the headline size): the Gaussians of two frames of occupancy or more
    # against the variance's own value; the others -- var = sum(g (x - mu)^2) / occ - (mu_new - mu)^2 is a difference of two sums over one or
    # two frames -- against the second moment about the previous mean, var + (mu_new - mu_old)^2, the quantity the accumulators carry
    # (HFB.c:1671-1678).  No entry is left out.
    low = ok & ~(np.asarray(ref["muOcc"])[:, None] >= 2.0)
    well = ok & ~low
    assert well.sum() > 0.3 * ok.sum()
    assert np.allclose(p["var"][well], ref_var[well], rtol=1e-4, atol=1e-6)
    moment = np.abs(ref_var) + (ref_mean - np.asarray(case["pk"]["mean"], np.float64)) ** 2
    assert (np.abs(p["var"] - ref_var)[low] <= 1e-4 * moment[low] + 1e-6).all(), float((np.abs(p["var"] - ref_var)[low] / moment[low]).max())
    assert np.allclose(p["compWeight"], np.asarray(upd["compWeight"], np.float64), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("mode", [6, 4], ids=["bf16x3fast", "bf16x3"])
def test_scoring_sits_out_what_setotprob_never_evaluates(native, oracle, mode, monkeypatch):
    """The pair kernels leave out, wavefront by wavefront, the (32 frames, pair of chain states) blocks that lie outside Setotprob's ranges
    of the un-pruned pass (HFB.c:1014 with 1177 / 1215: model q at frame t only for qLo-1 <= q <= qHi).  Nothing that the recursions read
    may change: log-probabilities and every beta / alpha the reference holds are bit-equal with and without the skip, the accumulators equal
    to the order of the atomics, and the log-probabilities are the oracle's."""
    from htk_amd import synth
    from util import batch_arrays
    s = synth.generate(60, 4, 45, 5, 420, 77)                      # 35 models = 105 chain states: two chunks of states, four tiles of frames, a taper that cuts through both
    s.feats[3] = s.feats[3][:300]; s.seqs[3] = s.seqs[3][:25]      # ragged
    pk = s.packed()
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    runs = {}
    for skip in (False, True):
        if skip:
            monkeypatch.delenv("HTKAMD_NO_TAPER_SKIP", raising=False)
        else:
            monkeypatch.setenv("HTKAMD_NO_TAPER_SKIP", "1")
        model, fb, acc, pr, st = run_fb(native, pk, utts, scoreMode=mode)
        assert (st == 1).all()
        runs[skip] = (pr, [fb.trellis(u) for u in range(len(utts))], acc.download(), fb.score_work())
    (pr0, tr0, a0, w0), (pr1, tr1, a1, w1) = runs[False], runs[True]
    assert np.array_equal(pr0, pr1)
    for g0, g1 in zip(tr0, tr1):
        for k in ("beta", "alpha"):
            assert np.array_equal(np.isnan(g0[k]), np.isnan(g1[k])) and np.array_equal(g0[k][~np.isnan(g0[k])], g1[k][~np.isnan(g1[k])]), k
        for k in ("qLo", "qHi", "aLo", "aHi"):
            assert np.array_equal(g0[k], g1[k])
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
        assert np.allclose(a0[k], a1[k], rtol=1e-11, atol=1e-13), k
    issued, everything, needed = w1
    assert needed <= issued < everything and w0[:2] == (issued, everything), (w0, w1)      # (the count is the batch's, not the switch's)
    om, oacc, ocfg = oracle.Model(pk), None, oracle.fb_cfg()
    oacc = oracle.Accs(om)
    for u, ut in enumerate(utts):
        rc, opr, _ = oracle.fb_utt(om, ocfg, ut["feat"], ut["seq"], oacc)
        assert rc == 1 and abs(opr - pr1[u]) <= 1e-6 * abs(opr)


def test_pass_in_two_phases_and_exchange_ranges(native, oracle, monkeypatch):
    """htkamd_fb_execute_begin / _mix (the multi-GPU hosts' form of the pass: a range of states' statistics travels while the next is summed):
    the states in three uneven ranges, last first, leave the accumulators of htkamd_fb_execute; htkamd_accs_state_ranges cuts the statistics'
    part of the vector into disjoint ranges that cover it; pack / unpack move them unchanged (fp64) or rounded to float once (fp32).
    With two-pair buckets (HTKAMD_ST_CAP) nearly every pair goes through the list kernel in front of the states: same sums, the oracle's."""
    from htk_amd import synth
    s = synth.generate(50, 4, 40, 6, 200, 91)
    pk = s.packed()
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    from util import batch_arrays
    model = native.Model(pk)
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = native.DevArray(X)                                        # (kept alive: the passes below run on this prepared batch)
    fb, acc = native.ForwardBackward(model), native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    cfg = native.fb_config(scoreMode=6)
    fb.execute(cfg, acc)
    pr, st = fb.results()
    assert (st == 1).all()
    whole = acc.download()
    cuts = [(31, 50), (0, 7), (7, 31)]
    for cap in (None, "2"):
        if cap:
            monkeypatch.setenv("HTKAMD_ST_CAP", cap)
        acc2 = native.Accs(model)
        assert fb.execute_begin(cfg, acc2)
        part = acc2.download()
        assert np.array_equal(part["tr"], whole["tr"]) or np.allclose(part["tr"], whole["tr"], rtol=1e-12)      # what no state owns is there behind _begin
        if not cap:
            assert not part["mu"].any()                                                                          # ... and nothing of the states yet
        for s0, s1 in cuts:
            fb.execute_mix(s0, s1)
        pr2, st2 = fb.results()
        assert np.array_equal(pr2, pr)
        got = acc2.download()
        # (the list kernel takes exp() where the state kernel of the fast class takes v_exp_f32: posteriors equal to 1e-7)
        for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
            assert np.allclose(got[k], whole[k], rtol=2e-6 if cap else 1e-11, atol=1e-6 if cap else 1e-12), (cap, k)
        monkeypatch.delenv("HTKAMD_ST_CAP", raising=False)
    om = oracle.Model(pk); oacc = oracle.Accs(om); ocfg = oracle.fb_cfg()
    for ut in utts:
        oracle.fb_utt(om, ocfg, ut["feat"], ut["seq"], oacc)
    for k in ("muOcc", "vaOcc", "wt", "wtOcc"):
        acc_close(got[k], getattr(oacc, k), "tiny buckets / %s" % k)
    occ = np.maximum(np.asarray(oacc.muOcc, np.float64), 1e-3)[:, None]
    for k in ("mu", "va"):                                             # sums about the current mean: an entry near zero is a cancellation, the Gaussian's occupancy its scale (bench.py's rule)
        ref = np.asarray(getattr(oacc, k), np.float64).reshape(occ.shape[0], -1)
        assert (np.abs(np.asarray(got[k]).reshape(ref.shape) - ref) <= 1e-4 * np.maximum(np.abs(ref), occ)).all(), k
    # the ranges: disjoint, and together the statistics' part of the vector (everything in front of nEgs)
    L = acc.lay
    seen = np.zeros(L.total, np.int32)
    for i, (s0, s1) in enumerate(cuts):
        for off, ln in acc.state_ranges(s0, s1, with_rest=(i == 2)):
            seen[off:off + ln] += 1
    assert (seen[:L.nEgs] == 1).all() and not seen[L.nEgs:].any()
    from htk_amd import herest                                             # the host mirror the CPU (gloo) tests use names the same ranges
    hl = herest.layout_from_packed(pk)
    for i, (s0, s1) in enumerate(cuts):
        assert herest.state_ranges(pk, hl, s0, s1, with_rest=(i == 2)) == acc.state_ranges(s0, s1, with_rest=(i == 2))
    # pack / unpack, on a stream of the library's making ordered behind the default stream (htkamd_stream_create / _wait: what a C host of the exchange in parts uses)
    import ctypes as C
    cs = C.c_void_p()
    native.check(native.lib().htkamd_stream_create(C.byref(cs)), "stream_create")
    native.check(native.lib().htkamd_stream_wait(cs, None), "stream_wait")
    v0 = acc.download()["vec"].copy()
    rg = acc.state_ranges(7, 31, with_rest=True)
    n = sum(l for _, l in rg)
    for wire, dt in ((0, np.float64), (1, np.float32)):
        buf = native.DevArray(np.zeros(n, dt))
        acc.pack_ranges(rg, wire, buf.ptr.value, stream=cs.value)
        native.check(native.lib().htkamd_stream_wait(None, cs), "stream_wait")
        flat = buf.to_host(dt, (n,))
        assert np.array_equal(flat, np.concatenate([v0[off:off + ln] for off, ln in rg]).astype(dt))
        acc.zero(None)
        acc.unpack_ranges(rg, wire, buf.ptr.value)
        v1 = acc.download()["vec"]
        want = np.zeros_like(v0)
        for off, ln in rg:
            want[off:off + ln] = v0[off:off + ln].astype(dt).astype(np.float64)
        assert np.array_equal(v1, want)
        acc.zero(None); acc.upload_add(v0)
    native.check(native.lib().htkamd_stream_destroy(cs), "stream_destroy")
    with pytest.raises(native.HtkAmdError):
        acc.state_ranges(5, 51)


# ----------------------------------------------------------------------------------------- forward-backward
@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", CASES)
def test_forward_backward_vs_reference(native, name, path):
    case = load_case(name)
    model, fb, acc, pr, st = run_fb(native, case["pk"], case["utts"], case["prune"], path=path)
    for u, ut in enumerate(case["utts"]):
        assert st[u] == ut["ok"] == 1
        assert abs(pr[u] - float(ut["pr"])) <= 1e-10 * abs(float(ut["pr"]))       # alpha/beta bar is 1e-4 relative
        g = fb.trellis(u)
        for k in ("qLo", "qHi", "aLo", "aHi"):
            assert np.array_equal(g[k], ut[k]), k
        if "beta" in ut:
            for k in ("beta", "alpha"):
                ref, got = ut[k], g[k]
                if k == "beta":
                    assert np.array_equal(np.isnan(ref), np.isnan(got))
                ok = ~np.isnan(ref) & (ref > -1e9)
                assert np.allclose(got[ok], ref[ok], rtol=1e-10, atol=0), k
                zero = ~np.isnan(ref) & (ref <= -1e9)                              # log-zero stays log-zero
                assert (got[zero] < -0.5e10).all()
            ok = ~np.isnan(ut["outp"])
            assert np.array_equal(g["outp"][ok], ut["outp"][ok])                   # output probabilities: bit-exact
    a = acc.download()
    ref = case["acc"]
    assert np.array_equal(a["nEgs"], ref["nEgs"])
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
        acc_close(a[k], ref[k], "%s/%s" % (name, k))
    assert a["nUttDone"] == len(case["utts"]) and a["nUttSkipped"] == 0
    assert abs(a["totalPr"] - float(ref["totalPr"])) <= 1e-5 * abs(float(ref["totalPr"]))   # HERest keeps it as float
    assert a["totalT"] == int(ref["totalT"])


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("prune", [dict(pruneInit=0.5, pruneInc=2.0, pruneLim=12.0), dict(pruneInit=0.01, pruneInc=0.0, pruneLim=0.01)],
                         ids=["retry", "overprune"])
def test_pruning_retry_and_skip(native, oracle, prune, path):
    """StepBack's retry loop (HFB.c:1332-1361) and the skip of over-pruned utterances, against the oracle."""
    from htk_amd import synth
    po = oracle
    s = synth.generate(30, 3, 20, 6, 90, 103)
    pk = s.packed()
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    model, fb, acc, pr, st = run_fb(native, pk, utts, prune, path=path)
    om = po.Model(pk); oacc = po.Accs(om); cfg = po.fb_cfg(**prune)
    nskip = 0
    for u in range(6):
        rc, opr, d = po.fb_utt(om, cfg, s.feats[u], s.seqs[u], oacc, dump=True)
        assert st[u] == rc
        if rc == 1:
            assert abs(pr[u] - opr) <= 1e-10 * abs(opr)
            g = fb.trellis(u)
            for k in ("qLo", "qHi", "aLo", "aHi"):
                assert np.array_equal(g[k], d[k]), k
        else:
            nskip += 1
    a = acc.download()
    assert a["nUttSkipped"] == nskip and a["nUttDone"] == 6 - nskip
    acc_close(a["muOcc"], oacc.muOcc, "muOcc"); acc_close(a["tr"], oacc.tr, "tr")


def test_ragged_batch_skips_and_empty(native, oracle):
    """Utterances of different lengths in one batch; one too short for its transcription (qt > T, HFB.c:1339);
    an empty batch is a no-op."""
    from htk_amd import synth
    po = oracle
    s = synth.generate(30, 2, 20, 5, 96, 9)
    pk = s.packed()
    lens = [96, 40, 5, 77, 24]                                 # Q = 8 models x minDur 3 = 24 frames minimum
    utts = [dict(seq=q, feat=x[:n]) for q, x, n in zip(s.seqs, s.feats, lens)]
    model, fb, acc, pr, st = run_fb(native, pk, utts)
    om = po.Model(pk); oacc = po.Accs(om); cfg = po.fb_cfg()
    for u, ut in enumerate(utts):
        rc, opr, _ = po.fb_utt(om, cfg, ut["feat"], ut["seq"], oacc)
        assert st[u] == rc
        if rc == 1:
            assert abs(pr[u] - opr) <= 1e-10 * abs(opr)
    assert list(st) == [1, 1, 0, 1, 1]
    a = acc.download()
    acc_close(a["mu"], oacc.mu, "mu"); acc_close(a["va"], oacc.va, "va"); acc_close(a["wt"], oacc.wt, "wt")
    assert np.array_equal(a["nEgs"], oacc.nEgs)
    # empty batch
    fb2 = native.ForwardBackward(model)
    dX = native.DevArray(np.zeros((1, 39), np.float32))
    fb2.prepare(dX.ptr.value, np.array([0], np.int32), np.array([0], np.int32), np.zeros(0, np.int32))
    acc2 = native.Accs(model)
    fb2.execute(native.fb_config(), acc2)
    assert acc2.download()["nUttDone"] == 0


def test_update_flags_subset(native, oracle):
    """-u m / -u v / -u t style passes only touch the requested accumulators (HFB.c:1663-1721)."""
    from htk_amd import capi
    case = load_case("fb_small")
    om = oracle.Model(case["pk"])
    for flags in (capi.UPMEANS, capi.UPVARS, capi.UPTRANS, capi.UPMIXES | capi.UPMEANS):
        model, fb, acc, pr, st = run_fb(native, case["pk"], case["utts"], uFlags=flags)
        oacc = oracle.Accs(om); cfg = oracle.fb_cfg(uFlags=flags)
        for ut in case["utts"]:
            oracle.fb_utt(om, cfg, ut["feat"], ut["seq"], oacc)
        a = acc.download()
        for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
            acc_close(a[k], getattr(oacc, k), "flags=%d %s" % (flags, k))


@pytest.mark.parametrize("stats_list", ["tiny", "off"])
@pytest.mark.parametrize("flags", [15, 1, 2])
def test_mixture_statistics_fallback_paths(native, oracle, stats_list, flags):
    """K4's record list (per-Gaussian reduction) is the default everywhere else; here the list is 128 records long, so that almost
    everything takes the overflow path, or absent (direct atomics): same accumulators (UpMixParms HFB.c:1426-1721)."""
    case = load_case("fb_topo")
    om = oracle.Model(case["pk"])
    model, fb, acc, pr, st = run_fb(native, case["pk"], case["utts"], uFlags=flags, stats_list=stats_list)
    oacc = oracle.Accs(om); cfg = oracle.fb_cfg(uFlags=flags)
    for ut in case["utts"]:
        oracle.fb_utt(om, cfg, ut["feat"], ut["seq"], oacc)
    a = acc.download()
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc"):
        acc_close(a[k], getattr(oacc, k), "%s flags=%d %s" % (stats_list, flags, k))


# ----------------------------------------------------------------------------------------- update (host C)
@pytest.mark.parametrize("name", ["fb_small", "fb_topo"])
def test_model_update_vs_reference_mmf(native, name):
    """Accumulate on the GPU, update with htk_amd/host/update.c, compare with the MMF the reference's HERest -m 1 wrote.
    Means within 1e-4*max(|ref|, sigma), variances/weights/gconst within 1e-4 relative."""
    case = load_case(name)
    model, fb, acc, pr, st = run_fb(native, case["pk"], case["utts"], case["prune"], debug=False)
    a = acc.download()
    stats = model.update(acc, a["vec"], minEgs=1, singleProcess=True)
    p = model.get_params()
    upd = case["upd"]
    ref_mean = np.asarray(upd["mean"], np.float64); ref_var = np.asarray(upd["var"], np.float64)
    ok = ~np.isnan(ref_mean)
    sigma = np.sqrt(np.where(ok, np.abs(ref_var), 1.0))
    assert (np.abs(p["mean"] - ref_mean)[ok] <= 1e-4 * np.maximum(np.abs(ref_mean), sigma)[ok] + 1e-6).all()
    assert np.allclose(p["var"][ok], ref_var[ok], rtol=1e-4, atol=1e-6)
    w = np.asarray(upd["compWeight"], np.float64)
    assert np.allclose(p["compWeight"], w, rtol=1e-4, atol=2e-6)
    gc = np.asarray(upd["gconst"], np.float64); okg = ~np.isnan(gc)
    assert np.allclose(p["gconst"][okg], gc[okg], rtol=1e-5, atol=1e-4)
    lin = np.where(p["transP"] > -0.5e10, np.exp(p["transP"].astype(np.float64)), 0.0)
    assert np.allclose(lin, np.asarray(upd["transLin"], np.float64), rtol=1e-4, atol=1e-6)
    assert "floored variance" not in case["log"] or stats["nFloorVar"] >= 0


@pytest.mark.parametrize("variant", ["plain", "single", "floors", "flags", "minegs"])
@pytest.mark.parametrize("name", ["fb_small", "fb_topo", "c3"])
def test_device_update_equals_host_update(native, name, variant):
    """htkamd_model_update_device against htkamd_model_update (host C, itself pinned to the MMF the reference's HERest wrote) from the
    SAME device accumulators: parameters equal to the last float bit but for the odd value where the device's double log()/exp()
    rounds the other way (<= 1 ulp, a handful at most), the update counters equal, and every derived table the kernels read -- so the
    scores of the next pass -- identical in both score modes."""
    from htk_amd import synth, capi
    if name == "c3":
        s = synth.generate(5000, 16, 6000, 24, 500, 3)
        pk, utts, prune = s.packed(), [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)], None
    else:
        case = load_case(name)
        pk, utts, prune = case["pk"], case["utts"], case["prune"]
    kw = dict(plain=dict(minEgs=1), single=dict(minEgs=1, singleProcess=True), floors=dict(minEgs=1, minVar=0.7, mixWeightFloor=2e-5 * 3),
              flags=dict(minEgs=1, uFlags=capi.UPMEANS | capi.UPTRANS, singleProcess=True), minegs=dict(minEgs=3))[variant]
    if variant == "floors":
        kw["varFloor"] = np.linspace(0.4, 1.2, int(pk["vecSize"])).astype(np.float32)
    mh, fb, acc, pr, st = run_fb(native, pk, utts, prune, debug=False)
    md = native.Model(pk)
    a = acc.download()
    sh = mh.update(acc, a["vec"], **kw)
    accd = native.Accs(md); accd.upload_add(a["vec"])
    sd = md.update_device(accd, **kw)
    assert sh == sd, (sh, sd)
    ph, pd = mh.get_params(), md.get_params()
    for k in ("mean", "var", "gconst", "compWeight", "transP"):
        x, y = ph[k].reshape(-1), pd[k].reshape(-1)
        ne = x != y
        assert ne.sum() <= max(3, 1e-5 * x.size), (k, int(ne.sum()))
        assert np.allclose(x[ne], y[ne], rtol=2.5e-7, atol=0), k
    qh, qd = mh.get_prepared(), md.get_prepared()
    assert np.array_equal(qh["minDur"], qd["minDur"])
    if all((ph[k] == pd[k]).all() for k in ph):
        for k in ("ivar", "gconst", "compLogWt"):
            assert np.array_equal(qh[k], qd[k]), k
        X = np.concatenate([u["feat"] for u in utts[:2]])[:300]
        states = np.arange(min(int(pk["numStates"]), 64), dtype=np.int32)
        for mode in (0, 1, 4):
            if mode == 1 and int(pk["vecSize"]) > 40:
                continue
            assert np.array_equal(mh.outp_block(X, states, mode=mode), md.outp_block(X, states, mode=mode)), mode
    # a second pass runs on the refreshed tables
    model2, fb2, acc2, pr2, st2 = None, None, None, None, None
    X, frameOff, labOff, labs = batch_arrays(utts[:4])
    dX = native.DevArray(X)
    res = []
    for m_ in (mh, md):
        fbm = native.ForwardBackward(m_); am = native.Accs(m_)
        fbm.prepare(dX.ptr.value, frameOff, labOff, labs)
        fbm.execute(native.fb_config(**(prune or {})), am)
        res.append(fbm.results()[0])
    assert np.allclose(res[0], res[1], rtol=1e-9, atol=0)


@pytest.mark.parametrize("variant", ["plain", "single", "floors", "means_only"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_update_with_tied_vectors_equals_host_update(native, seed, variant):
    """Sets with tied mean / variance vectors (~u ~v): htkamd_model_update_device pools the statistics of a group and decides "whose
    variance keeps the mean-shift term" by minimum scan position (update.hip: k_upd_first) -- against the host update, which walks the
    models in scan order as UpdateModels does (HERest.c:974-1122, 1262-1321; pinned to the reference's MMFs by tests/test_cli_tools.py).
    Random groups across states and models, a random scan order, models below minEgs."""
    from htk_amd import synth, capi
    rng = np.random.default_rng(seed)
    s = synth.generate(40, 3, 30, 20, 90, 11 + seed)
    pk = dict(s.packed())
    G, D = int(pk["numGauss"]), int(pk["vecSize"])
    mean, var = np.array(pk["mean"], np.float32).reshape(G, D), np.array(pk["var"], np.float32).reshape(G, D)
    ms, vs = np.full(G, -1, np.int32), np.full(G, -1, np.int32)
    for share, vec in ((ms, mean), (vs, var)):
        perm = rng.permutation(G)
        pos, gid = 0, 0
        while pos < G // 2:
            n = int(rng.integers(2, 5))
            grp = perm[pos:pos + n]; pos += n
            share[grp] = gid; gid += 1
            vec[grp] = vec[grp.min()]
    pk["mean"], pk["var"] = mean.reshape(np.shape(pk["mean"])), var.reshape(np.shape(pk["var"]))
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    kw = dict(plain=dict(minEgs=2), single=dict(minEgs=1, singleProcess=True), floors=dict(minEgs=1, minVar=0.6, mixWeightFloor=2e-5 * 3),
              means_only=dict(minEgs=1, uFlags=capi.UPMEANS | capi.UPMIXES))[variant]
    models, order = [], None
    for _ in range(2):
        m = native.Model(pk); m.set_sharing(ms, vs)
        if order is None:
            order = rng.permutation(m.H).astype(np.int32)
        m.set_scan_order(order)
        models.append(m)
    mh, md = models
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = native.DevArray(X)
    fb, acc = native.ForwardBackward(mh), native.Accs(mh)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(native.fb_config(), acc)
    a = acc.download()
    sh = mh.update(acc, a["vec"], **kw)
    accd = native.Accs(md); accd.upload_add(a["vec"])
    sd = md.update_device(accd, **kw)
    assert sh == sd, (sh, sd)
    ph, pd = mh.get_params(), md.get_params()
    for k in ("mean", "var", "gconst", "compWeight", "transP"):
        x, y = ph[k].reshape(-1), pd[k].reshape(-1)
        ne = x != y
        assert ne.sum() <= max(3, 1e-5 * x.size), (k, int(ne.sum()), np.abs(x - y).max())
        assert np.allclose(x[ne], y[ne], rtol=2.5e-7, atol=0), k
    # every user of a vector holds the group's value
    for share, k in ((ms, "mean"), (vs, "var")):
        v = pd[k].reshape(G, D)
        for gid in range(int(share.max()) + 1):
            grp = np.nonzero(share == gid)[0]
            assert (v[grp] == v[grp[0]]).all()
    assert (pd["mean"] != np.asarray(pk["mean"]).reshape(pd["mean"].shape)).any()


def test_em_iterations_increase_likelihood(native):
    """Three Baum-Welch iterations through prepare/execute/update: the total log-likelihood must not decrease."""
    from htk_amd import synth
    s = synth.generate(20, 2, 12, 24, 72, 77)
    pk = s.packed()
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    X, frameOff, labOff, labs = batch_arrays(utts)
    model = native.Model(pk); dX = native.DevArray(X)
    fb = native.ForwardBackward(model); acc = native.Accs(model)
    tot = []
    for it in range(3):
        acc.zero()
        fb.prepare(dX.ptr.value, frameOff, labOff, labs)
        fb.execute(native.fb_config(), acc)
        pr, st = fb.results()
        assert (st == 1).all()
        tot.append(pr.sum())
        a = acc.download()
        model.update(acc, a["vec"], minEgs=1, minVar=0.01)
    assert tot[1] >= tot[0] - 1e-6 and tot[2] >= tot[1] - 1e-6


# ----------------------------------------------------------------------------------------- full-size properties
def test_config2_properties(native):
    """1k states x 8 mix, 64 x 500-frame utterances (BASELINE config 2): size-independent invariants.
    sum_j gamma_j(t) = 1 per frame -> total occupancy = frames; weight counts = occupancy; transitions out of the
    entry state = one per model instance; first utterances match the reference's printed per-frame log probabilities."""
    from htk_amd import synth
    import os, util
    s = synth.generate(1000, 8, 2000, 64, 500, 1)
    pk = s.packed()
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    model, fb, acc, pr, st = run_fb(native, pk, utts, debug=False)
    assert (st == 1).all()
    known = np.load(os.path.join(util.GOLDEN, "c2_known.npz"))["per_frame"]
    for u in range(3):
        assert float("%e" % (pr[u] / 500)) == float("%e" % known[u])
    a = acc.download()
    frames = 64 * 500
    assert abs(a["muOcc"].sum() - frames) < 1e-3 * frames          # components below the MINFORPROB prune are dropped
    assert abs(a["wt"].sum() - a["muOcc"].sum()) < 1e-6 * frames
    assert abs(a["wtOcc"].sum() - a["muOcc"].sum()) < 1e-6 * frames
    assert a["nEgs"].sum() == 64 * 41 and a["totalT"] == frames
    tr = a["tr"].reshape(5, 5)
    assert abs(tr[0, 1] - 64 * 41) < 1e-6 * 64 * 41               # every model instance is entered exactly once
    assert abs(tr[1:4, 4].sum() - 64 * 41) < 1e-6 * 64 * 41       # ... and left exactly once
    assert abs(a["trOcc"][1:4].sum() - frames) < 1e-6 * frames    # emitting-state occupancy sums to the frame count
    assert "%.6e" % (a["totalPr"] / a["totalT"]) == "-6.108214e+01"  # HERest: "average log prob per frame" 
    assert a["nEval"] == fb.frame_states() == 64 * 47220


@pytest.mark.parametrize("mode", [0, 3, 6, 34], ids=["exact", "fast", "bf16x3fast", "fastest"])
def test_config3_headline_size(native, oracle, mode):
    """The configuration bench.py measures (BASELINE config[2] per GPU: 5k tied states x 16 mix, D = 39, 500-frame utterances of 41
    models), 64 utterances, in the exact mode and in the mode the bench runs (matrix-core scores + fast LAdd): utterance
    log-probabilities against the oracle on a sample (1e-10 exact / 1e-6 tolerance class), accumulators of the sampled
    utterances' states through acc_close on a batch of just those utterances, and the size-independent invariants on all 64."""
    from htk_amd import synth
    s = synth.generate(5000, 16, 6000, 64, 500, 3)
    pk = s.packed()
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    model, fb, acc, pr, st = run_fb(native, pk, utts, debug=False, scoreMode=mode)
    assert (st == 1).all()
    om = oracle.Model(pk)
    cfg = oracle.fb_cfg()
    sample = [0, 17, 63]
    oacc = oracle.Accs(om)
    tol = 1e-6 if mode else 1e-10
    for u in sample:
        rc, opr, _ = oracle.fb_utt(om, cfg, utts[u]["feat"], utts[u]["seq"], oacc)
        assert rc == 1 and abs(pr[u] - opr) <= tol * abs(opr), (u, pr[u], opr)
    a = acc.download()
    frames = 64 * 500
    assert abs(a["muOcc"].sum() - frames) < 1e-3 * frames
    assert abs(a["wt"].sum() - a["muOcc"].sum()) < 1e-6 * frames and abs(a["wtOcc"].sum() - a["muOcc"].sum()) < 1e-6 * frames
    assert a["nEgs"].sum() == 64 * 41 and a["totalT"] == frames
    tr = a["tr"].reshape(5, 5)
    assert abs(tr[0, 1] - 64 * 41) < 1e-5 * 64 * 41 and abs(tr[1:4, 4].sum() - 64 * 41) < 1e-5 * 64 * 41
    assert abs(a["trOcc"][1:4].sum() - frames) < 1e-5 * frames
    assert a["nEval"] == fb.frame_states() == 64 * 47220
    assert abs(a["totalPr"] - pr.sum()) <= 1e-9 * abs(pr.sum())
    # the sampled utterances alone: every accumulator against the oracle's
    model2, fb2, acc2, pr2, st2 = run_fb(native, pk, [utts[u] for u in sample], debug=False, scoreMode=mode)
    a2 = acc2.download()
    assert np.array_equal(a2["nEgs"], oacc.nEgs)
    rt = 1e-4
    for k in ("muOcc", "vaOcc", "wt", "wtOcc", "tr", "trOcc") + (() if mode else ("mu", "va")):
        acc_close(a2[k], getattr(oacc, k), "c3/%s" % k, rtol=rt)
    if mode:
        # first- and second-order sums are kept about the current mean (HFB.c:1671-1678): an entry near zero is a cancellation of
        # terms of the size of its Gaussian's occupancy, which is therefore the scale a tolerance-class change of the posteriors
        # moves it by (the exact mode above holds every entry to 1e-4 of its own value)
        occ = np.maximum(np.asarray(oacc.muOcc, np.float64), 1e-3)[:, None]
        for k in ("mu", "va"):
            got = np.asarray(a2[k], np.float64).reshape(occ.shape[0], -1); ref = np.asarray(getattr(oacc, k), np.float64).reshape(occ.shape[0], -1)
            assert (np.abs(got - ref) <= rt * np.maximum(np.abs(ref), occ)).all(), k
    for k, u in enumerate(sample):
        assert pr2[k] == pr[u]                              # a score does not depend on the batch it is in


def test_accumulator_merge_is_load_accs(native):
    """upload_add == LoadAccs (HTrain.c:1625): adding a dumped vector to a zeroed set reproduces it; adding twice doubles."""
    case = load_case("fb_small")
    model, fb, acc, pr, st = run_fb(native, case["pk"], case["utts"], debug=False)
    v = acc.download()["vec"].copy()
    acc2 = native.Accs(model)
    acc2.upload_add(v); acc2.upload_add(v)
    assert np.array_equal(acc2.download()["vec"], 2 * v)


# ----------------------------------------------------------------------------------------- Viterbi alignment (K5)
def _align(native, pk, seqs, feats, beam=1.0e10):
    model = native.Model(pk)
    utts = [dict(seq=q, feat=x) for q, x in zip(seqs, feats)]
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = native.DevArray(X)
    v = native.Viterbi(model)
    return v.align(dX.ptr.value, frameOff, labOff, labs, genBeam=beam)


def _check_vs_oracle(oracle, om, got, seqs, feats, beam):
    for u, g in enumerate(got):
        r = oracle.viterbi_align(om, feats[u], seqs[u], genBeam=beam)
        if r is None:
            assert g["status"] == 0
            continue
        assert g["status"] == 1 and g["total"] == r["total"]            # token likes: bit-identical doubles
        visited = g["segStart"] >= 0
        assert visited.sum() == r["n"]
        assert np.array_equal(g["segStart"][visited], r["start"]) and np.array_equal(g["segEnd"][visited], r["end"])
        assert np.array_equal(g["segScore"][visited], r["score"])
        assert np.array_equal(g["modStart"], r["modStart"]) and np.array_equal(g["modEnd"], r["modEnd"])
        assert np.array_equal(g["modScore"], r["modScore"])


@pytest.mark.parametrize("beam", [1.0e10, 40.0, 15.0])
def test_viterbi_alignment_bit_identical(native, oracle, beam):
    """State alignments must be bit-identical to the reference's (north star): compared with the committed HVite
    label files line by line and with the oracle's segment table (also under pruning beams)."""
    import os
    import util
    from htk_amd import synth
    s = synth.generate(60, 4, 40, 4, 120, 5)
    pk = s.packed()
    got = _align(native, pk, s.seqs, s.feats, beam)
    _check_vs_oracle(oracle, oracle.Model(pk), got, s.seqs, s.feats, beam)
    z = np.load(os.path.join(util.GOLDEN, "hvite_rec.npz"))
    names = ["p%d" % i for i in range(40)]
    tag = {1.0e10: "small", 40.0: "small_t40"}.get(beam)
    if tag:
        for u in range(4):
            want = [l for l in str(z["%s_%d" % (tag, u)]).split("\n") if l.strip()]
            assert native.format_rec(got[u], names) == want


def test_viterbi_mixed_topologies_and_tee(native, oracle):
    """3/4/5-state models with a skip transition vs the reference's label files; with the tee model in the chain
    (which the reference's label writer cannot emit) vs the oracle's token passing."""
    import os
    import util
    from htk_amd import synth
    pk, names, seqs, feats = synth.make_topo_set()
    z = np.load(os.path.join(util.GOLDEN, "hvite_rec.npz"))
    noTee = [np.array([h for h in q if h != 2], np.int32) for q in seqs]
    got = _align(native, pk, noTee, feats)
    for u in range(len(seqs)):
        want = [l for l in str(z["topo_%d" % u]).split("\n") if l.strip()]
        assert native.format_rec(got[u], names) == want
    got = _align(native, pk, seqs, feats)
    _check_vs_oracle(oracle, oracle.Model(pk), got, seqs, feats, 1.0e10)


def test_viterbi_config2_size(native, oracle):
    """1k x 8, 500-frame utterances: first lines of the alignment the survey recorded from the reference, the
    per-utterance score HVite prints, and segment bookkeeping (contiguous cover of all frames)."""
    from htk_amd import synth
    s = synth.generate(1000, 8, 2000, 8, 500, 1)
    got = _align(native, s.packed(), s.seqs, s.feats)
    names = ["p%d" % i for i in range(2000)]
    lines = native.format_rec(got[0], names)
    assert lines[0] == "0 400000 s2 -228.724319 p63 -742.978210 p63"
    assert lines[1] == "400000 800000 s3 -255.099365" and lines[2] == "800000 1200000 s4 -259.154510"
    assert lines[3] == "1200000 1600000 s2 -249.230179 p447 -742.860291 p447"
    assert "%.4f" % (got[6]["total"] / 500) == "-61.1013"                   # "[500 frames] -61.1013" for u00006
    for g in got:
        assert g["status"] == 1
        assert g["segStart"][0] == 0 and g["segEnd"][-1] == 500
        assert np.array_equal(g["segStart"][1:], g["segEnd"][:-1])
        assert abs(g["segScore"].sum() - g["total"]) < 1e-6 * abs(g["total"])


# ----------------------------------------------------------------------------------------- MFCC front end (K6)
def _test_wave(n=48000, seed=7):
    rng = np.random.default_rng(seed); t = np.arange(n) / 16000
    return (3000 * np.sin(2 * np.pi * 440 * t) * np.sin(2 * np.pi * 3 * t) + rng.normal(0, 800, n)).clip(-32768, 32767).astype("<i2")


@pytest.mark.parametrize("kind,kw", [("MFCC_0_D_A", {}), ("MFCC_E_D_A", {}), ("MFCC_E_D_A_Z", {}), ("MFCC_0", {}),
                                     ("MFCC_E_D", dict(rawEnergy=False, zMeanSource=True)),
                                     ("MFCC_0_D_A", dict(loFreq=300.0, hiFreq=3400.0, numChans=20, numCeps=10, usePower=True))])
def test_mfcc_matches_reference_front_end(native, oracle, kind, kw):
    """WAV -> MFCC on the device vs the oracle (bit-equal to the reference's HCopy, tests/test_oracle_golden.py).
    Bit for bit: the kernels keep the reference's operation order and its double-precision detours (FFT twiddles tabulated on the host)."""
    waves = [_test_wave(48000, 7), _test_wave(12345, 8), _test_wave(400, 9), _test_wave(399, 10)]     # ragged, 1 frame, 0 frames
    got, frameOff = native.Mfcc(native.mfcc_config(kind, **kw)).compute_host(waves)
    ocfg = oracle.mfcc_cfg(kind, **kw)
    refs = [oracle.mfcc(w, ocfg) for w in waves]
    assert list(frameOff) == list(np.concatenate([[0], np.cumsum([r.shape[0] for r in refs])]))
    assert refs[2].shape[0] == 1 and refs[3].shape[0] == 0
    ref = np.concatenate(refs)
    assert got.shape == ref.shape
    # (round 5: measured over 1.3 M floats of these configurations and longer waveforms, tools/mfcc_diag.py -- not one differs)
    assert np.array_equal(got, ref)


def test_mfcc_known_answer_config5(native):
    """3 s synthetic WAV, MFCC_0_D_A: 298 frames, first-frame statics and C0 as HCopy/HList print them."""
    got, frameOff = native.Mfcc(native.mfcc_config("MFCC_0_D_A")).compute_host([_test_wave()])
    assert got.shape == (298, 39)
    want = [-20.591204, -2.0929291, -4.9431148, -4.984157, -9.7785, -11.552636, -8.135165, -4.291196, 0.3099544, 4.2238774, 3.6917002, 8.413123]
    assert np.allclose(got[0, :12], want, rtol=1e-6) and abs(got[0, 12] - 75.76163) < 1e-4
    assert np.allclose(got[0, 13:18], [-0.199, -0.417, -0.302, -0.200, -0.803], atol=1e-3)


def test_wav_to_scores_on_device(native, oracle):
    """BASELINE config 5: raw 16 kHz waveform -> on-device MFCC_0_D_A -> GMM scoring, no host round trip of the features."""
    import ctypes as C
    from htk_amd import synth
    s = synth.generate(25, 4, 10, 0, 10, 3)
    pk = s.packed()
    gm, om = native.Model(pk), oracle.Model(pk)
    mf = native.Mfcc(native.mfcc_config("MFCC_0_D_A"))
    dX, frameOff = mf.compute([_test_wave()])
    T = int(frameOff[-1])
    states = np.arange(25, dtype=np.int32)
    dS = native.DevArray(states); dO = native.DevArray(nbytes=4 * T * 25)
    native.check(native.lib().htkamd_outp_block(gm.h, dX.ptr, C.c_int(T), dS.ptr, C.c_int(25), dO.ptr, C.c_int(T), None), "outp_block")
    got = dO.to_host(np.float32, (25, T)).T
    X = dX.to_host(np.float32, (T, 39))
    assert np.array_equal(got, om.score_block(X, states))                 # scoring of the device features: bit-exact
    ref = om.score_block(oracle.mfcc(_test_wave(), oracle.mfcc_cfg("MFCC_0_D_A")), states)
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-2)


def test_viterbi_long_chains_general_kernel(native, oracle):
    """Transcriptions of more than 64 models (one wavefront cannot hold them) go through the workgroup-per-utterance
    kernel: same tokens, same segments as the oracle; short and long utterances mixed in one batch, with and without a beam."""
    from htk_amd import synth
    s = synth.generate(60, 3, 40, 3, 700, 23, D=13)
    pk = s.packed()
    rng = np.random.default_rng(4)
    seqs = [np.asarray(rng.integers(0, 40, size=n), np.int32) for n in (90, 30, 150)]
    feats = []
    for q in seqs:                                                  # 4 frames per state of the chain
        fr = []
        for h in q:
            for st in pk["hmmState"][pk["hmmStateOff"][h]:pk["hmmStateOff"][h + 1]]:
                c = pk["stateCompOff"][st]
                fr.append(pk["mean"][pk["compGauss"][c]] + rng.normal(size=(4, 13)) * np.sqrt(pk["var"][pk["compGauss"][c]]))
        feats.append(np.concatenate(fr).astype(np.float32))
    om = oracle.Model(pk)
    for beam in (1.0e10, 60.0):
        got = _align(native, pk, seqs, feats, beam=beam)
        _check_vs_oracle(oracle, om, got, seqs, feats, beam)
        assert all(g["status"] == 1 for g in got)


def test_tied_mixture_components_accumulate_jointly(native, oracle):
    """A model set with shared mixture pdfs (~m macros, read from the reference's HHEd output): statistics of a shared Gaussian
    are the sum over the components that use it, as in the reference (one MuAcc/VaAcc per MixPDF); HIP path vs oracle."""
    import os
    from htk_amd import synth
    gold = os.path.join(os.path.dirname(__file__), "golden", "mmf")
    m = native.Mmf(files=[os.path.join(gold, "syn_tied.mmf")], hmm_list=os.path.join(gold, "syn_list"))
    pk = m.packed()
    s = synth.generate(10, 3, 6, 4, 60, 77, D=5)                    # the set the fixture was made from: use its utterances
    utts = [dict(seq=np.asarray(q, np.int32), feat=f) for q, f in zip(s.seqs, s.feats)]
    model, fb, acc, pr, st = run_fb(native, pk, utts, debug=False)
    om = oracle.Model(pk); oacc = oracle.Accs(om)
    for u, ut in enumerate(utts):
        rc, opr, _ = oracle.fb_utt(om, oracle.fb_cfg(), ut["feat"], ut["seq"], oacc)
        assert rc == 1 and st[u] == 1 and abs(pr[u] - opr) <= 1e-10 * abs(opr)
    a = acc.download()
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
        acc_close(a[k], getattr(oacc, k), k)


@pytest.mark.parametrize("general", [False, True], ids=["wave", "general"])
def test_beta_beam_bottom_model_above_tapered_top(native, oracle, general):
    """A case the randomised sweep found (tests/fuzz_parity.py): tight pruning with retries on a chain with tee models, where the
    taper pulls the top of the beta beam BELOW its bottom model at t = 1.  The reference keeps that beam (qLo > qHi: its test
    against the top only runs while the bottom is being raised, HFB.c:1259-1272) and then fails in the alpha pass (error 7390);
    both kernel families must follow it -- same status, pr and beams as the oracle -- instead of retrying with a wider beam."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fuzz", "alpha_prune_case.npz"))
    pk = {k[3:]: (z[k] if z[k].ndim else z[k].item()) for k in z.files if k.startswith("pk_")}
    if pk["gconst"].size == 0:
        pk["gconst"] = None
    for k in ("vecSize", "numStates", "numComp", "numGauss", "numTrans", "numPhys"):
        pk[k] = int(pk[k])
    prune = dict(pruneInit=float(z["prune"][0]), pruneInc=float(z["prune"][1]), pruneLim=float(z["prune"][2]))
    utts = [dict(seq=z["seq"], feat=z["feat"])]
    model, fb, acc, pr, st = run_fb(native, pk, utts, prune, general=general)
    om = oracle.Model(pk); oacc = oracle.Accs(om)
    rc, opr, d = oracle.fb_utt(om, oracle.fb_cfg(**prune), z["feat"], z["seq"], oacc, dump=True)
    assert rc == -7390 and st[0] == rc
    g = fb.trellis(0, want_alpha=False)
    assert np.array_equal(g["qLo"], d["qLo"]) and np.array_equal(g["qHi"], d["qHi"]) and g["qLo"][0] > g["qHi"][0]
    a = acc.download()
    assert a["nUttSkipped"] == 1 and a["nUttDone"] == 0 and not a["muOcc"].any()
    # without pruning the utterance is fine on every path
    model, fb, acc, pr, st = run_fb(native, pk, utts, general=general)
    rc, opr, _ = oracle.fb_utt(om, oracle.fb_cfg(), z["feat"], z["seq"], oracle.Accs(om))
    assert rc == 1 and st[0] == 1 and abs(pr[0] - opr) <= 1e-10 * abs(opr)


# ----------------------------------------------------------------------------------------- long chains: 2 / 4 wavefronts per utterance
def _concat_chains(rng, seqs, feats, targets):
    """Utterances with chains of about `targets` models: whole utterances of the pool strung together (labels and frames)."""
    out = []
    for tgt in targets:
        q, x = [], []
        while sum(len(a) for a in q) < tgt:
            k = int(rng.integers(0, len(seqs)))
            q.append(np.asarray(seqs[k], np.int32)); x.append(feats[k])
        out.append(dict(seq=np.concatenate(q), feat=np.concatenate(x)))
    return out


@pytest.mark.parametrize("prune", [None, dict(pruneInit=150.0, pruneInc=0.0, pruneLim=150.0), dict(pruneInit=30.0, pruneInc=60.0, pruneLim=400.0)],
                         ids=["noprune", "beam", "retry"])
@pytest.mark.parametrize("topo", [False, True], ids=["3state", "topo"])
def test_long_chains_every_kernel_class(native, oracle, prune, topo):
    """Chains of 40 .. 500 models in ONE batch: <= 64 models run on one wavefront, <= 128 on two, <= 256 on four, <= 512 on eight
    (values that cross a 64-model boundary go through LDS).  Same pr, beams, beta/alpha and accumulators as the oracle."""
    from htk_amd import synth
    rng = np.random.default_rng(11 + int(topo))
    if topo:
        pk, names, seqs, feats = synth.make_topo_set(seed=77, D=13, NU=8)
    else:
        s = synth.generate(30, 3, 20, 12, 60, 4242, D=13)
        pk, seqs, feats = s.packed(), s.seqs, s.feats
    utts = _concat_chains(rng, seqs, feats, [40, 64, 66, 120, 128, 131, 200, 255, 258, 300, 500])
    if topo:                                            # tee models on the wavefront boundaries of some chains (lanes 63|64, 127|128 ...)
        tee = names.index("sp")
        for ut in utts[2:]:
            q = ut["seq"]
            for pos in (63, 64, 127, 128, 191, 256):
                if pos + 1 < len(q) and q[pos - 1] != tee and q[pos + 1] != tee and rng.random() < 0.7:
                    q[pos] = tee
    Qs = [len(u["seq"]) for u in utts]
    assert min(Qs) <= 64 and any(64 < q <= 128 for q in Qs) and any(128 < q <= 256 for q in Qs) and 256 < max(Qs) <= 512
    model, fb, acc, pr, st = run_fb(native, pk, utts, prune)
    om = oracle.Model(pk); oacc = oracle.Accs(om); cfg = oracle.fb_cfg(**(prune or {}))
    ndone = 0
    for u, ut in enumerate(utts):
        rc, opr, d = oracle.fb_utt(om, cfg, ut["feat"], ut["seq"], oacc, dump=True)
        assert st[u] == rc, (u, Qs[u], st[u], rc)
        if rc != 1:
            continue
        ndone += 1
        assert abs(pr[u] - opr) <= 1e-10 * abs(opr), (u, Qs[u])
        g = fb.trellis(u)
        for k in ("qLo", "qHi", "aLo", "aHi"):
            assert np.array_equal(g[k], d[k]), (k, u, Qs[u])
        for k in ("beta", "alpha"):
            ref, got = d[k], g[k]
            ok = ~np.isnan(got) & ~np.isnan(ref) & (ref > -1e9)
            assert ok.any() and np.allclose(got[ok], ref[ok], rtol=1e-10, atol=0), (k, u, Qs[u])
    assert ndone >= 5
    a = acc.download()
    assert a["nUttDone"] == ndone
    for k in ("muOcc", "wtOcc", "trOcc", "tr", "wt", "mu", "va"):
        acc_close(a[k], getattr(oacc, k), k)
    assert np.array_equal(a["nEgs"], oacc.nEgs)


@pytest.mark.parametrize("beam", [1.0e10, 60.0], ids=["nobeam", "beam60"])
@pytest.mark.parametrize("topo", [False, True], ids=["3state", "topo"])
def test_viterbi_long_chains_every_kernel_class(native, oracle, beam, topo):
    """Alignment chains of 40 .. 600 models in one batch: 1 / 2 / 4 / 8 wavefronts per utterance, beyond 512 models the workgroup kernel.
    The topology set puts tee models anywhere in the chain, also in the first and last lane of a wavefront (lanes 63|64, 127|128 ...),
    where the entry/exit hand-over crosses wavefronts.  Tokens, segments and scores identical to the oracle's."""
    from htk_amd import synth
    rng = np.random.default_rng(5 + int(topo))
    if topo:
        pk, names, seqs, feats = synth.make_topo_set(seed=78, D=13, NU=8)
        tee = names.index("sp")
    else:
        s = synth.generate(30, 3, 20, 12, 60, 4243, D=13)
        pk, seqs, feats = s.packed(), s.seqs, s.feats
    utts = _concat_chains(rng, seqs, feats, [40, 64, 66, 120, 128, 131, 200, 255, 258, 300, 500, 600])
    if topo:                                            # force tee models onto the wavefront boundaries of some chains
        for ut in utts[2:]:
            q = ut["seq"]
            for pos in (63, 64, 127, 128, 191, 256):
                if pos + 1 < len(q) and q[pos - 1] != tee and q[pos + 1] != tee and rng.random() < 0.7:
                    q[pos] = tee
    sq, ft = [u["seq"] for u in utts], [u["feat"] for u in utts]
    got = _align(native, pk, sq, ft, beam=beam)
    _check_vs_oracle(oracle, oracle.Model(pk), got, sq, ft, beam)
    assert sum(g["status"] == 1 for g in got) >= 6


@pytest.mark.parametrize("mode", [0, 6, 34], ids=["exact", "bf16x3fast", "fastest"])
def test_run_to_run_reproducibility(native, mode):
    """The statistics are summed with fp64 atomics, so their LAST bits depend on the order in which wavefronts arrive; everything computed
    per utterance (log-probabilities, beams, trellis) does not.  Two passes over one batch: `pr` bit-identical, accumulators equal to
    1e-12 relative (the summation-order noise of an fp64 sum; the reference's float accumulators carry 1e-7), counters identical."""
    from htk_amd import synth
    s = synth.generate(40, 4, 30, 12, 120, 4242, D=39)
    model = native.Model(s.packed())
    X = np.concatenate(s.feats)
    frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int32)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
    labs = np.concatenate(s.seqs).astype(np.int32)
    dX = native.DevArray(X)
    runs = []
    for k in range(3):
        fb = native.ForwardBackward(model); acc = native.Accs(model)
        fb.prepare(dX.ptr.value, frameOff, labOff, labs)
        fb.execute(native.fb_config(uFlags=15, scoreMode=mode, pruneInit=250.0, pruneInc=150.0, pruneLim=1000.0), acc)
        pr, st = fb.results()
        runs.append((np.array(pr), np.array(st), acc.download()["vec"].copy()))
    for pr, st, vec in runs[1:]:
        assert np.array_equal(pr, runs[0][0]) and np.array_equal(st, runs[0][1])
        ref = runs[0][2]
        scale = np.maximum(np.abs(ref), 1e-6)
        assert (np.abs(vec - ref) <= 1e-12 * scale + 1e-14).all(), float(np.max(np.abs(vec - ref) / scale))
        lay = native.accs_layout(s.packed())
        for k in ("nEgs", "totalT", "nUttDone", "nUttSkipped", "nEval"):
            o = getattr(lay, k)
            assert vec[o] == ref[o]
