"""Multi-stream forward-backward on the device (HFB.c:1026-1066 Setotprob with S > 1, :1499-1602 UpMixParms' stream
loop): a row of scores per (stream, chain state), their float sum as the state's log probability (k_combine_streams), per-stream
posteriors with "the other streams" added (k_mixstats_ms), one WtAcc per (state, stream), Gaussians re-estimated on their stream's
dimensions only.

Against what:
  * S = 3 demo set: the reference's own accumulators (`HERest -p 1`) and re-estimated MMF (tests/golden/make_streams_golden.py).  The
    reference meets a tied state a second time at a frame through `sum/2` of replaced values (HFB.c:1059), which for S = 3 is the same
    number as the first visit up to float rounding -- so its accumulators are matched like every other set's (1e-4 against float sums), not bit for bit.
  * any S: oracle/htk_oracle.c with `ms_intended` (every visit computes the first visit's values); the oracle WITHOUT it is pinned to
    the reference float for float in tests/test_streams.py, S = 2 with the reference's defect included.
  * random multi-stream sets: S = 2, 3, 4, streams with one and with several components, all kernels paths."""
import os

import numpy as np
import pytest

import streams_util as su
import test_cli_tools as cli
from util import batch_arrays, acc_close

pytestmark = pytest.mark.gpu
DEMO = su.DEMO
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(native, pk, utts, path=None, mode=0, prune=None, uFlags=15, compat=0):
    model = native.Model(pk)
    if compat:
        model.set_compat(compat)
    X, frameOff, labOff, labs = batch_arrays(utts)
    dX = native.DevArray(X)
    fb = native.ForwardBackward(model, debug=True, force_general=(path == "general"), no_state_path=(path == "wave"))
    acc = native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(native.fb_config(uFlags=uFlags, scoreMode=mode, **(prune or {})), acc)
    pr, st = fb.results()
    return model, fb, acc, pr, st


def _oracle_accs(oracle, pk, utts, prune=None, uFlags=15, intended=True):
    om = oracle.Model(pk, ms_intended=intended)
    acc = oracle.Accs(om)
    prs = []
    for u in utts:
        rc, pr, _ = oracle.fb_utt(om, oracle.fb_cfg(uFlags=uFlags, **(prune or {})), u["feat"], u["seq"], acc)
        prs.append(pr if rc == 1 else np.nan)
    return om, acc, np.array(prs)


def _close_stat(got, ref, occ, sigma2, k, rtol, what):
    """First- / second-order sums about the OLD mean: sum L (x - mu) and sum L (x - mu)^2 are sums of terms of size L sigma and L sigma^2
    that largely cancel in the first case, so their natural scale is occ sigma (occ sigma^2), not the net value."""
    s2 = np.where(np.isfinite(sigma2), sigma2, 0.0)
    scale = np.maximum(np.abs(ref), occ[:, None] * (np.sqrt(s2) if k == "mu" else s2))
    err = np.abs(got - ref)
    bad = err > rtol * np.maximum(scale, 1e-3)
    assert not bad.any(), "%s %s: %d outside tolerance, worst %g of scale %g" % (what, k, int(bad.sum()), err[bad].max(), scale[bad][np.argmax(err[bad])])


def _compare(a, ref, rtol, what, var):
    """ref: dict of arrays (oracle accumulators or the reference's file)"""
    G = var.shape[0]
    for k in ("mu", "va"):
        r = np.asarray(ref[k], np.float64).reshape(G, -1)
        _close_stat(a[k].reshape(r.shape), r, np.asarray(ref[k + "Occ"], np.float64), var.astype(np.float64), k, rtol, what)
    for k in ("muOcc", "vaOcc", "wt", "wtOcc", "tr", "trOcc"):
        acc_close(a[k], np.asarray(ref[k], np.float64), "%s %s" % (what, k), rtol=rtol, floor=1e-3)


def _odict(oacc):
    return {k: getattr(oacc, k) for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc")}


@pytest.mark.parametrize("path", ["state", "wave", "general"])
@pytest.mark.parametrize("S", [3, 2])
def test_demo_stream_sets_against_oracle_and_reference(native, oracle, S, path):
    d = os.path.join(DEMO, "hmm_streams%d" % S)
    mmf = native.Mmf(files=[os.path.join(d, "newMacros")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    utts = su.demo_utterances(native, oracle, mmf)
    prune = dict(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0)
    model, fb, acc, pr, st = _run(native, pk, utts, path=path, prune=prune)
    assert (st == 1).all()
    a = acc.download()
    om, oacc, opr = _oracle_accs(oracle, pk, utts, prune=dict(pruneInit=2000.0))
    assert np.allclose(pr, opr, rtol=1e-9, atol=0)
    _compare(a, _odict(oacc), 1e-4, "S=%d %s vs oracle" % (S, path), pk["var"])        # float accumulators on the oracle side (as in the reference), fp64 sums here
    # the dimensions outside a Gaussian's stream collect nothing
    gs = np.zeros(pk["numGauss"], np.int32)
    for e in range(pk["numStates"] * S):
        gs[pk["compGauss"][pk["stateCompOff"][e]:pk["stateCompOff"][e + 1]]] = e % S
    outside = pk["dimStream"][None, :] != gs[:, None]
    assert (a["mu"].reshape(outside.shape)[outside] == 0).all() and (a["va"].reshape(outside.shape)[outside] == 0).all()
    if S == 3:
        # the reference itself: accumulators of `HERest -p 1`, its summary line
        lay = native.accs_layout(pk)
        v = np.zeros(lay.total, np.float64)
        native.accs_load_file(pk, v, list(mmf.phys_names), os.path.join(d, "HER1.acc"))
        _compare(a, {k: v[getattr(lay, k):getattr(lay, k) + a[k].size] for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc")}, 1e-4,
                 "S=3 %s vs the reference's accumulators" % path, pk["var"])
        assert "average log prob per frame = %e" % (a["totalPr"] / a["totalT"]) in open(os.path.join(d, "herest.log")).read()


@pytest.mark.parametrize("path", ["state", "general"])
def test_demo_two_streams_with_the_reference_s_second_visit_arithmetic(native, oracle, path):
    """htkamd_model_set_compat(HTKAMD_COMPAT_STREAM_REVISIT): a tied state met again in one Setotprob call gets the halved sum of the
    streams' replaced values (HFB.c:1059) -- for two streams HALF its log probability.  Against the oracle WITHOUT `ms_intended` (pinned
    to the reference float for float, tests/test_streams.py) and against the reference's own `HERest -p 1` accumulators and summary line
    (-33.6 per frame where the sum of the streams gives -59.1)."""
    d = os.path.join(DEMO, "hmm_streams2")
    mmf = native.Mmf(files=[os.path.join(d, "newMacros")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    utts = su.demo_utterances(native, oracle, mmf)
    prune = dict(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0)
    model, fb, acc, pr, st = _run(native, pk, utts, path=path, prune=prune, compat=native.COMPAT_STREAM_REVISIT)
    assert (st == 1).all()
    a = acc.download()
    om, oacc, opr = _oracle_accs(oracle, pk, utts, prune=dict(pruneInit=2000.0), intended=False)
    assert np.allclose(pr, opr, rtol=1e-9, atol=0)
    _compare(a, _odict(oacc), 1e-4, "S=2 compat %s vs oracle" % path, pk["var"])
    lay = native.accs_layout(pk)
    v = np.zeros(lay.total, np.float64)
    native.accs_load_file(pk, v, list(mmf.phys_names), os.path.join(d, "HER1.acc"))
    _compare(a, {k: v[getattr(lay, k):getattr(lay, k) + a[k].size] for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc")}, 1e-4,
             "S=2 compat %s vs the reference's accumulators" % path, pk["var"])
    assert "average log prob per frame = %e" % (a["totalPr"] / a["totalT"]) in open(os.path.join(d, "herest.log")).read()
    # and without the switch the same set gives the sum of the streams (what the test above holds against `ms_intended`)
    _, _, acc0, pr0, _ = _run(native, pk, utts, path=path, prune=prune)
    assert pr0.sum() < pr.sum() - 1000.0


@pytest.mark.parametrize("seed", list(range(11, 27)))
def test_random_stream_sets_with_the_reference_s_second_visit_arithmetic(native, oracle, seed):
    """The same on random sets of two and four streams with repeated models in the transcriptions (every utterance meets tied states
    again), beams that prune, and a retry of StepBack (pruneInc): against the oracle without `ms_intended`.  (Four streams: the second
    visit's value, 1.5 times the log probability, lies BELOW it -- it is no upper bound for a component's posterior; the sweep of
    tests/fuzz_parity.py found the alpha kernel pruning mixture seeds with it.)"""
    from htk_amd import synth
    rng = np.random.default_rng(seed)
    widths = (10, 10) if seed % 2 else (6, 6, 6, 2)
    s = synth.generate(12, 1, 5, 6, 60, 40 + seed, D=sum(widths))          # five models: every transcription repeats some
    pk = su.make_multistream(s.packed(), list(widths), rng, max_mix=4, single=())
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    prune = dict(pruneInit=float(rng.choice([60.0, 150.0, 2000.0])), pruneInc=50.0, pruneLim=2000.0)
    model, fb, acc, pr, st = _run(native, pk, utts, prune=prune, compat=native.COMPAT_STREAM_REVISIT)
    om, oacc, opr = _oracle_accs(oracle, pk, utts, prune=prune, intended=False)
    ok = st == 1
    assert (ok == np.isfinite(opr)).all() and ok.any()
    assert np.allclose(pr[ok], opr[ok], rtol=1e-9, atol=0)
    _compare(acc.download(), _odict(oacc), 1e-4, "compat, %d streams" % len(widths), pk["var"])


@pytest.mark.parametrize("device_update", [False, True])
def test_demo_three_streams_reestimated_model_equals_the_reference(native, oracle, tmp_path, device_update):
    d = os.path.join(DEMO, "hmm_streams3")
    mmf = native.Mmf(files=[os.path.join(d, "newMacros")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    utts = su.demo_utterances(native, oracle, mmf)
    model, fb, acc, pr, st = _run(native, pk, utts, prune=dict(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0))
    kw = dict(minEgs=3, minVar=0.05, mixWeightFloor=3 * 1.0e-5)
    stats = model.update_device(acc, **kw) if device_update else model.update(acc, acc.download()["vec"], **kw)
    assert "Total %d floored variance elements in %d different mixes" % (stats["nFloorVar"], stats["nFloorVarMix"]) in open(os.path.join(d, "herest.log")).read()
    p = model.get_params()
    out = str(tmp_path / "newMacros")
    mmf.write(p, one_file=out)
    cli._mmf_close(cli._mmf_numbers(out), cli._mmf_numbers(os.path.join(d, "after_herest")))
    # the structure of the file is the reference's: same keywords in the same order
    kw_of = lambda path: [t for t in open(path).read().split() if t.startswith("<") or t.startswith("~")]
    assert kw_of(out) == kw_of(os.path.join(d, "after_herest"))


def test_herest_cli_three_streams(native, tmp_path):
    tools = os.path.join(ROOT, "tools", "bin")
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    out = tmp_path / "next"; out.mkdir()
    d = os.path.join(DEMO, "hmm_streams3")
    r = cli.run([os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(d, "newMacros"), "-M", str(out),
                 "-s", str(tmp_path / "stats"), "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + cli.demo_train_files())
    assert r.returncode == 0, r.stderr
    for line in open(os.path.join(d, "herest.log")).read().splitlines():
        assert line in r.stdout, (line, r.stdout[-400:])
    cli._mmf_close(cli._mmf_numbers(str(out / "newMacros")), cli._mmf_numbers(os.path.join(d, "after_herest")))
    # the occupation statistics file: the first stream's WtAcc per state (PrintStats HERest.c:680-695)
    ours, theirs = (tmp_path / "stats").read_text().split(), open(os.path.join(d, "stats")).read().split()
    assert len(ours) == len(theirs)
    for x, y in zip(ours, theirs):
        assert x == y or abs(float(x) - float(y)) <= 1e-4 * max(abs(float(y)), 1.0), (x, y)
    # parallel mode: the accumulator file of this run is the reference's to the float
    acc = tmp_path / "acc"; acc.mkdir()
    r = cli.run([os.path.join(tools, "herest"), "-p", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(d, "newMacros"), "-M", str(acc),
                 "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + cli.demo_train_files())
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(str(acc / "HER1.acc")) == os.path.getsize(os.path.join(d, "HER1.acc"))


def test_herest_cli_two_streams_compat(native, tmp_path):
    """tools/bin/herest --compat on the demo set split 13 | 13: the reference HERest's summary lines, statistics file and re-estimated
    MMF -- the numbers its second-visit arithmetic gives (-33.6 per frame); without the switch the sum of the streams (-59.1)."""
    tools = os.path.join(ROOT, "tools", "bin")
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    d = os.path.join(DEMO, "hmm_streams2")
    def run(extra, out):
        out.mkdir()
        return cli.run([os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(d, "newMacros"), "-M", str(out),
                        "-s", str(out / "stats"), "-L", os.path.join(DEMO, "labels"), "-t", "2000.0"] + extra + [os.path.join(DEMO, "bcplist")] + cli.demo_train_files())
    r = run(["--compat"], tmp_path / "c")
    assert r.returncode == 0, r.stderr
    for line in open(os.path.join(d, "herest.log")).read().splitlines():
        assert line in r.stdout, (line, r.stdout[-400:])
    cli._mmf_close(cli._mmf_numbers(str(tmp_path / "c" / "newMacros")), cli._mmf_numbers(os.path.join(d, "after_herest")))
    ours, theirs = (tmp_path / "c" / "stats").read_text().split(), open(os.path.join(d, "stats")).read().split()
    assert len(ours) == len(theirs)
    for x, y in zip(ours, theirs):
        assert x == y or abs(float(x) - float(y)) <= 1e-4 * max(abs(float(y)), 1.0), (x, y)
    r0 = run([], tmp_path / "p")
    assert r0.returncode == 0 and "average log prob per frame = -5.9" in r0.stdout, r0.stdout[-300:]


@pytest.mark.parametrize("mode", [0, 6, 34])
@pytest.mark.parametrize("widths,single,seed", [((10, 10), (), 1), ((8, 8, 4), (2,), 2), ((6, 6, 6, 2), (), 3), ((12, 8), (0, 1), 4), ((5, 15), (1,), 5)])
def test_random_stream_sets(native, oracle, widths, single, seed, mode):
    """Random sets on random transcriptions; `single`: streams with one Gaussian (their posterior is the state's occupation, HFB.c:1584);
    (0, 1) of two streams = a set without any mixture (maxM == 1: the seed is log alpha + log beta - pr)."""
    from htk_amd import synth
    rng = np.random.default_rng(seed)
    s = synth.generate(12, 1, 10, 6, 60, 40 + seed, D=sum(widths))
    pk = su.make_multistream(s.packed(), list(widths), rng, max_mix=4, single=single)
    utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
    om, oacc, opr = _oracle_accs(oracle, pk, utts)
    ok = ~np.isnan(opr)
    assert ok.sum() >= len(utts) // 2
    for path in ("state", "wave", "general"):
        model, fb, acc, pr, st = _run(native, pk, utts, path=path, mode=mode)
        assert ((st == 1) == ok).all()
        rt = 1e-9 if mode == 0 else 2e-5
        assert np.allclose(pr[ok], opr[ok], rtol=rt, atol=0), (path, np.abs(pr[ok] / opr[ok] - 1).max())
        _compare(acc.download(), _odict(oacc), 1e-4, "%s mode %d %s" % (widths, mode, path), pk["var"])


TMIX = os.path.join(DEMO, "hmm_tmix")


@pytest.mark.parametrize("path", ["state", "wave", "general"])
@pytest.mark.parametrize("kind", ["tiedhs", "tiedhs3"])
def test_tied_mixture_sets_against_oracle_and_reference(native, oracle, kind, path):
    """hsKind TIEDHS on the device (k_tm_pool / k_tm_state / k_mixstats_tm): utterance log probabilities and accumulators against the
    oracle (itself equal to the reference's accumulator file float for float: tests/test_streams.py) and against that file."""
    mmf = native.Mmf(files=[os.path.join(TMIX, kind + "_newMacros")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    utts = su.demo_utterances(native, oracle, mmf)
    model, fb, acc, pr, st = _run(native, pk, utts, path=path, prune=dict(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0))
    assert (st == 1).all()
    a = acc.download()
    om, oacc, opr = _oracle_accs(oracle, pk, utts, prune=dict(pruneInit=2000.0), intended=False)
    assert np.allclose(pr, opr, rtol=1e-7, atol=0), np.abs(pr / opr - 1).max()          # float scores: a last bit of the double sum's order
    var = np.where(np.isfinite(pk["var"]), pk["var"], 0.0)
    _compare(a, _odict(oacc), 1e-4, "%s %s vs oracle" % (kind, path), var)
    lay = native.accs_layout(pk)
    v = np.zeros(lay.total, np.float64)
    native.accs_load_file(pk, v, list(mmf.phys_names), os.path.join(TMIX, kind + "_HER1.acc"))
    _compare(a, {k: v[getattr(lay, k):getattr(lay, k) + a[k].size] for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc")}, 1e-4,
             "%s %s vs the reference's accumulators" % (kind, path), var)
    assert "average log prob per frame = %e" % (a["totalPr"] / a["totalT"]) in open(os.path.join(TMIX, kind + ".log")).read()


@pytest.mark.parametrize("device_update", [False, True])
@pytest.mark.parametrize("kind", ["tiedhs", "tiedhs3"])
def test_tied_mixture_reestimated_model_equals_the_reference(native, oracle, tmp_path, kind, device_update):
    mmf = native.Mmf(files=[os.path.join(TMIX, kind + "_newMacros")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    utts = su.demo_utterances(native, oracle, mmf)
    model, fb, acc, pr, st = _run(native, pk, utts, prune=dict(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0))
    kw = dict(minEgs=3, minVar=0.05, mixWeightFloor=3 * 1.0e-5)
    stats = model.update_device(acc, **kw) if device_update else model.update(acc, acc.download()["vec"], **kw)
    assert "Total %d floored variance elements in %d different mixes" % (stats["nFloorVar"], stats["nFloorVarMix"]) in open(os.path.join(TMIX, kind + ".log")).read()
    out = str(tmp_path / "newMacros")
    mmf.write(model.get_params(), one_file=out)
    cli._mmf_close(cli._mmf_numbers(out), cli._mmf_numbers(os.path.join(TMIX, kind + "_after_herest")))


def test_herest_cli_tied_mixtures(native, tmp_path):
    tools = os.path.join(ROOT, "tools", "bin")
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    for kind in ("tiedhs", "tiedhs3"):
        out = tmp_path / kind; out.mkdir()
        r = cli.run([os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(TMIX, kind + "_newMacros"),
                     "-M", str(out), "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + cli.demo_train_files())
        assert r.returncode == 0, r.stderr
        for line in open(os.path.join(TMIX, kind + ".log")).read().splitlines():
            assert line in r.stdout, (line, r.stdout[-400:])
        cli._mmf_close(cli._mmf_numbers(str(out / (kind + "_newMacros"))), cli._mmf_numbers(os.path.join(TMIX, kind + "_after_herest")))


@pytest.mark.parametrize("name", ["after_herest", "sw_after_herest"])
def test_hvite_cli_on_a_three_stream_set(native, tmp_path, name):
    """Recognition and forced alignment of a multi-stream set: state output probability = sum over streams of w_s x (the stream's mixture
    log likelihood) (cPOutP HRec.c:510-548), here with <SWEIGHTS> 1 1 1 and 1 0.5 2 -- the label files of the reference's HVite, line for
    line (tests/golden/make_streams_hvite_golden.py)."""
    import json
    tools = os.path.join(ROOT, "tools", "bin")
    d3 = os.path.join(DEMO, "hmm_streams3")
    exp = json.load(open(os.path.join(d3, "hvite_expected.json")))[name]
    conf = tmp_path / "hvite.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    path = lambda u: os.path.join(DEMO, "test" if u.startswith("te") else "train", u + ".mfc")
    for what, opts in (("rec", ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0", "-m", "-f"]),
                       ("align", ["-a", "-m", "-f", "-L", os.path.join(DEMO, "labels"), "-t", "300.0"])):
        out = tmp_path / what; out.mkdir()
        names = sorted(exp[what])
        r = cli.run([os.path.join(tools, "hvite"), "-C", str(conf), "-H", os.path.join(d3, name), "-l", str(out)] + opts +
                    [os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + [path(u) for u in names])
        assert r.returncode == 0, r.stderr
        for u in names:
            assert (out / (u + ".rec")).read_text().splitlines() == exp[what][u], (what, u)


@pytest.mark.parametrize("name", ["tiedhs_after_herest", "tiedhs3_after_herest"])
def test_hvite_cli_on_tied_mixture_sets(native, tmp_path, name):
    """Recognition and forced alignment of <TMIX> sets, one stream and three: per frame PrecomputeTMix with HVite's -c threshold
    (HRec.c:1987), per state SOutP's sum over the kept pool entries, stream-weighted (cPOutP) -- the label files of the reference's
    HVite line for line (tests/golden/make_tmix_hvite_golden.py), with the default threshold and with -c 3.0."""
    import json
    tools = os.path.join(ROOT, "tools", "bin")
    exp = json.load(open(os.path.join(TMIX, "hvite_expected.json")))[name]
    conf = tmp_path / "hvite.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    path = lambda u: os.path.join(DEMO, "test" if u.startswith("te") else "train", u + ".mfc")
    for what, opts in (("rec", ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0", "-m", "-f"]),
                       ("rec_c3", ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0", "-c", "3.0"]),
                       ("align", ["-a", "-m", "-f", "-L", os.path.join(DEMO, "labels"), "-t", "300.0"])):
        out = tmp_path / what; out.mkdir()
        names = sorted(exp[what])
        r = cli.run([os.path.join(tools, "hvite"), "-C", str(conf), "-H", os.path.join(TMIX, name), "-l", str(out)] + opts +
                    [os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + [path(u) for u in names])
        assert r.returncode == 0, r.stderr
        for u in names:
            assert (out / (u + ".rec")).read_text().splitlines() == exp[what][u], (what, u)


def test_hvite_cli_nbest_on_stream_and_tied_mixture_sets(native, tmp_path):
    """N-best decoding (token sets, -n 3 2) of a three-stream tied-mixture set and of a three-stream set with stream weights 1 0.5 2: the
    label files of the reference's HVite run beside ours (oracle/_ref travels to the GPU box), byte for byte."""
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref", "HVite")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/HVite is not on this box")
    conf = tmp_path / "hvite.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    files = [os.path.join(DEMO, "test", "te1.mfc"), os.path.join(DEMO, "test", "te2.mfc"), os.path.join(DEMO, "train", "tr3.mfc")]
    for k, mset in enumerate([os.path.join(TMIX, "tiedhs3_after_herest"), os.path.join(DEMO, "hmm_streams3", "sw_after_herest")]):
        outs = []
        for who, exe in (("ours", os.path.join(ROOT, "tools", "bin", "hvite")), ("ref", ref)):
            out = tmp_path / ("%s%d" % (who, k)); out.mkdir(); outs.append(out)
            r = subprocess.run([exe, "-C", str(conf), "-H", mset, "-w", os.path.join(DEMO, "monLattice"), "-l", str(out), "-t", "300.0", "-p", "5.0", "-s", "0.0",
                                "-n", "3", "2", os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + files, capture_output=True, text=True)
            assert r.returncode == 0, (who, r.stdout[-400:], r.stderr[-400:])
        for f in ("te1.rec", "te2.rec", "tr3.rec"):
            assert (outs[0] / f).read_text() == (outs[1] / f).read_text(), (mset, f)
            assert "///" in (outs[0] / f).read_text()
