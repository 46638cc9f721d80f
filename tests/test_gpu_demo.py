"""BASELINE config[0]: the reference's own CPU-runnable case -- HTKDemo's monophone system (5 models, 1 Gaussian per
state, D = 26 = MFCC_E on disk + deltas at load), first embedded re-estimation pass:
    HERest -w 3 -v 0.05 -C herest.conf(TARGETKIND = MFCC_E_D) -u tmvw -d hmm.1 -M hmm.2 -L labels -t 2000.0 bcplist tr*.mfc
Inputs and the reference's outputs are the committed fixtures of tests/golden/make_demo_golden.py; every step runs through
the C ABI: MMF files -> model, parameter files -> device table with deltas, label files -> model sequences,
forward-backward + accumulators on the GPU, UpdateModels, MMF files out."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DEMO = os.path.join(os.path.dirname(__file__), "golden", "demo")


def _run_pass(native):
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm1"))
    pk = mmf.packed()
    model = native.Model(pk)
    files = sorted(f for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    stat, seqs = [], []
    for f in files:
        X, period, kind = native.parm_read(os.path.join(DEMO, "train", f))
        assert X.shape[1] == 13 and period == 100000 and kind == (6 | 0o100)          # MFCC_E
        stat.append(X)
        labs = native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab")))
        seqs.append(np.array([mmf.logical[n] for n, _, _, _ in labs], np.int32))
    dX, frameOff, cols = native.parm_add_qualifiers(stat, hasD=True)                  # TARGETKIND = MFCC_E_D
    assert cols == 26 == pk["vecSize"]
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    fb = native.ForwardBackward(model)
    acc = native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, np.concatenate(seqs))
    fb.execute(native.fb_config(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0), acc)   # -t 2000.0
    pr, st = fb.results()
    a = acc.download()
    stats = model.update(acc, a["vec"], minEgs=3, minVar=0.05, mixWeightFloor=3 * 1.0e-5)   # defaults + -v 0.05 -w 3
    return mmf, model, pr, st, a, stats, dX, frameOff, stat


def test_htkdemo_first_herest_pass(native, oracle, tmp_path):
    mmf, model, pr, st, a, stats, dX, frameOff, stat = _run_pass(native)
    log = open(os.path.join(DEMO, "herest_pass1.log")).read()
    # the deltas computed on the device are the reference's (oracle pinned against HCopy in this container)
    got = dX.to_host(np.float32, (int(frameOff[-1]), 26))
    for u, X in enumerate(stat):
        assert np.array_equal(got[frameOff[u]:frameOff[u + 1]], oracle.add_qualifiers(X, True, False))
    assert (st == 1).all() and int(a["totalT"]) == 1811 and "1.811000e+03" in log
    ref_avg = float(re.search(r"average log prob per frame = (\S+)", log).group(1))
    assert ref_avg == -5.900196e+01
    assert "%e" % (a["totalPr"] / a["totalT"]) == "%e" % ref_avg                      # the line HERest prints
    m = re.search(r"Total (\d+) floored variance elements in (\d+) different mixes", log)
    assert (stats["nFloorVar"], stats["nFloorVarMix"]) == (int(m.group(1)), int(m.group(2))) == (27, 15)
    # re-estimated models against the files the reference wrote
    ref = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm2_expected")).packed()
    p = model.get_params()
    sigma = np.sqrt(ref["var"])
    assert (np.abs(p["mean"] - ref["mean"]) <= 1e-4 * np.maximum(np.abs(ref["mean"]), sigma) + 1e-6).all()
    assert np.allclose(p["var"], ref["var"], rtol=1e-4, atol=1e-7)
    assert np.allclose(p["gconst"], ref["gconst"], rtol=1e-5)
    lin = lambda t: np.where(t > -0.5e10, np.exp(t.astype(np.float64)), 0.0)
    assert np.allclose(lin(p["transP"]), lin(ref["transP"]), rtol=1e-4, atol=1e-7)
    # and out through the MMF writer: same files up to the last printed digit
    mmf.write(p, out_dir=str(tmp_path))
    for name in "SCVNL":
        ours = (tmp_path / name).read_text().split()
        theirs = open(os.path.join(DEMO, "hmm2_expected", name)).read().split()
        assert len(ours) == len(theirs)
        for x, y in zip(ours, theirs):
            if x != y:
                assert abs(float(x) - float(y)) <= 1e-4 * max(abs(float(y)), 1e-3), (name, x, y)


def test_htkdemo_pass_is_the_same_through_the_mfma_scores(native):
    """D = 26 is one of the MFMA path's sizes: log-likelihood and models agree with the exact path to 1e-4."""
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm1"))
    model = native.Model(mmf.packed())
    files = sorted(f for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    stat = [native.parm_read(os.path.join(DEMO, "train", f))[0] for f in files]
    seqs = [np.array([mmf.logical[n] for n, _, _, _ in native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab")))], np.int32) for f in files]
    dX, frameOff, cols = native.parm_add_qualifiers(stat, hasD=True)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    out = []
    for mode in (0, 1):
        model = native.Model(mmf.packed())
        fb = native.ForwardBackward(model); acc = native.Accs(model)
        fb.prepare(dX.ptr.value, frameOff, labOff, np.concatenate(seqs))
        fb.execute(native.fb_config(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0, scoreMode=mode), acc)
        pr, st = fb.results()
        a = acc.download()
        model.update(acc, a["vec"], minEgs=3, minVar=0.05, mixWeightFloor=3e-5)
        out.append((pr, a, model.get_params()))
    assert np.allclose(out[0][0], out[1][0], rtol=1e-6)
    for k in ("muOcc", "trOcc"):
        assert np.allclose(out[0][1][k], out[1][1][k], rtol=1e-4, atol=1e-6), k
    p0, p1 = out[0][2], out[1][2]
    sigma = np.sqrt(p0["var"])
    err_m = np.abs(p1["mean"] - p0["mean"]) / np.maximum(np.abs(p0["mean"]), sigma)
    err_v = np.abs(p1["var"] - p0["var"]) / p0["var"]
    assert err_m.max() <= 1e-4 and err_v.max() <= 1e-4, (err_m.max(), err_v.max())


def test_examples_on_fixture_files(native, tmp_path):
    """The two example drivers (HERest-pass and HVite data flows over HTK files) run on the committed fixtures and print /
    write what the reference's tools did."""
    import glob
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "examples"))
    import herest_pass, hvite_decode
    from htk_amd import synth
    a, stats = herest_pass.main(["--hmmlist", os.path.join(DEMO, "bcplist"), "--hmmdir", os.path.join(DEMO, "hmm1"), "--labdir", os.path.join(DEMO, "labels"),
                                 "--target-kind", "MFCC_E_D", "--prune", "2000", "--minvar", "0.05", "--mixfloor", "3", "--outdir", str(tmp_path / "hmm2"),
                                 "--data"] + sorted(glob.glob(os.path.join(DEMO, "train", "*.mfc"))))
    assert "%e" % (a["totalPr"] / a["totalT"]) == "-5.900196e+01" and (stats["nFloorVar"], stats["nFloorVarMix"]) == (27, 15)
    assert sorted(os.listdir(tmp_path / "hmm2")) == sorted("SCVNL")
    d = os.path.join(os.path.dirname(__file__), "golden", "decode", "bigram")
    z = np.load(os.path.join(d, "feats.npz"))
    files = []
    for u in range(len(z.files)):
        fn = str(tmp_path / ("u%d.mfc" % u)); synth.write_htk_param(fn, z["u%d" % u], kind=9); files.append(fn)
    exp = json.load(open(os.path.join(d, "expected.json")))
    hvite_decode.main(["--mmf", os.path.join(d, "MMF"), "--hmmlist", os.path.join(d, "hmmlist"), "--dict", os.path.join(d, "dict"), "--net", os.path.join(d, "net.slf"),
                       "--beam", "250", "--lmscale", "5", "--wordpen", "-10", "--models", "--out", str(tmp_path / "rec.mlf")] + files)
    got = open(tmp_path / "rec.mlf").read().split("\n")
    want = exp["-m -t 250.0 -s 5.0 -p -10.0"]
    lines = [l for l in got[1:] if l and not l.startswith('"') and l != "."]
    assert lines == [l for u in range(len(files)) for l in want["u%d" % u]]


@pytest.mark.parametrize("name", list("SCVNL"))
def test_hrest_isolated_unit_reestimation(native, oracle, name):
    """The demo's HRest step (HRest -u tmvw -w 3 -v 0.05 -i 10 -l X hmm.0/X -> hmm.1/X) through the library: the tokens of one
    model as single-model utterances, forward-backward + update iterated like ReEstimateModel.  Per-iteration average log
    probability against the reference's trace, final model against the file it wrote."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from examples.hrest_model import hrest
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm0"))
    files = sorted(f for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    tables, labels = [], []
    for f in files:
        X, _, _ = native.parm_read(os.path.join(DEMO, "train", f))
        tables.append(oracle.parm_qualify(X, hasD=True))                             # TARGETKIND = MFCC_E_D over the whole file
        labels.append(native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab"))))
    ref_lines = [l for l in open(os.path.join(DEMO, "hinit_hrest.log")) if l.startswith("HRest %s: Ave LogProb" % name)]
    ref = [(float(re.search(r"= +(-?[\d.]+) using", l).group(1)), int(re.search(r"using (\d+) examples", l).group(1))) for l in ref_lines]
    # the stopping rule |change| < 1e-4 is applied to a float near -600 (spacing 6e-5) whose last bits depend on HRest's own
    # output-probability rounding: our own stop may come a pass or two earlier or later ...
    _, own = hrest(native, mmf, name, tables, labels, max_iter=10, min_var=0.05, mix_weight_floor=3 * 1.0e-5)
    assert abs(len(own) - len(ref)) <= 3, (own, ref)
    # ... so the numbers are compared pass by pass over the reference's number of passes
    model, hist = hrest(native, mmf, name, tables, labels, max_iter=len(ref), epsilon=0.0, min_var=0.05, mix_weight_floor=3 * 1.0e-5)
    assert len(hist) == len(ref), (hist, ref)
    for (p, n), (rp, rn) in zip(hist, ref):
        assert n == rn and abs(p - rp) <= 1e-6 * abs(rp), (name, p, rp)
    # final parameters of this model against hmm.1 (the other four models are untouched: no tokens of theirs in the batch)
    rmmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm1"))
    rq, pk, p = rmmf.packed(), mmf.packed(), model.get_params()
    h, rh = mmf.logical[name], rmmf.logical[name]
    for s, rs in zip(pk["hmmState"][pk["hmmStateOff"][h]:pk["hmmStateOff"][h + 1]], rq["hmmState"][rq["hmmStateOff"][rh]:rq["hmmStateOff"][rh + 1]]):
        g, rg = int(pk["compGauss"][pk["stateCompOff"][s]]), int(rq["compGauss"][rq["stateCompOff"][rs]])
        sigma = np.sqrt(rq["var"][rg])
        assert (np.abs(p["mean"][g] - rq["mean"][rg]) <= 1e-4 * np.maximum(np.abs(rq["mean"][rg]), sigma) + 1e-6).all()
        assert np.allclose(p["var"][g], rq["var"][rg], rtol=1e-4, atol=1e-7)
    t, rt = int(pk["hmmTrans"][h]), int(rq["hmmTrans"][rh])
    N = int(pk["transN"][t])
    lin = lambda v: np.where(v > -0.5e10, np.exp(v.astype(np.float64)), 0.0)
    assert np.allclose(lin(p["transP"][pk["transOff"][t]:pk["transOff"][t] + N * N]), lin(rq["transP"][rq["transOff"][rt]:rq["transOff"][rt] + N * N]), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("name", list("SCVNL"))
def test_hinit_viterbi_training(native, oracle, name):
    """The demo's HInit step (HInit -i 10 -l X -o X proto/X -> hmm.0/X): uniform segmentation, then Viterbi alignment on the
    device (the batch alignment kernel, one single-model utterance per token) and re-estimation from the aligned frames, pass
    by pass against the reference's trace ("Iteration k: Average LogP") and the model it wrote."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from examples.hinit_model import hinit
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "proto"))
    files = sorted(f for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    tables, labels = [], []
    for f in files:
        X, _, _ = native.parm_read(os.path.join(DEMO, "train", f))
        tables.append(oracle.parm_qualify(X, hasD=True))
        labels.append(native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab"))))
    pk, model, hist, converged = hinit(native, mmf, name, tables, labels, max_iter=10)
    lines = [l for l in open(os.path.join(DEMO, "hinit_hrest.log")) if l.startswith("HInit %s:" % name)]
    ref = [float(re.search(r"Average LogP = *(-?[\d.]+)", l).group(1)) for l in lines if "Average LogP" in l]
    assert len(hist) == len(ref), (hist, ref)
    for p, rp in zip(hist, ref):
        assert abs(p - rp) <= 1e-6 * abs(rp), (name, hist, ref)
    assert converged == any("converged" in l for l in lines)
    rmmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm0"))
    rq, p = rmmf.packed(), model.get_params()
    q0 = mmf.packed()
    h, rh = mmf.logical[name], rmmf.logical[name]
    for s, rs in zip(q0["hmmState"][q0["hmmStateOff"][h]:q0["hmmStateOff"][h + 1]], rq["hmmState"][rq["hmmStateOff"][rh]:rq["hmmStateOff"][rh + 1]]):
        g, rg = int(q0["compGauss"][q0["stateCompOff"][s]]), int(rq["compGauss"][rq["stateCompOff"][rs]])
        sigma = np.sqrt(rq["var"][rg])
        assert (np.abs(p["mean"][g] - rq["mean"][rg]) <= 1e-4 * np.maximum(np.abs(rq["mean"][rg]), sigma) + 1e-6).all()
        assert np.allclose(p["var"][g], rq["var"][rg], rtol=1e-4, atol=1e-7)
    t, rt = int(q0["hmmTrans"][h]), int(rq["hmmTrans"][rh])
    N = int(q0["transN"][t])
    lin = lambda v: np.where(v > -0.5e10, np.exp(v.astype(np.float64)), 0.0)
    assert np.allclose(lin(p["transP"][q0["transOff"][t]:q0["transOff"][t] + N * N]), lin(rq["transP"][rq["transOff"][rt]:rq["transOff"][rt] + N * N]), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("name", list("SCVNL"))
def test_hinit_mixture_training(native, oracle, name):
    """HInit on prototypes with 2 / 3 / 2 mixture components per state (tests/golden/make_hinit_mix_golden.py): uniform segmentation +
    FlatCluster (HTrain.c:763), Viterbi alignment and best-component choice on the device, hard-assignment re-estimation -- pass by pass
    against the reference's trace and the model it wrote (weights, means, variances, transitions)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from examples.hinit_model import hinit
    gold = os.path.join(DEMO, "hinit_mix")
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(gold, "proto"))
    files = sorted(f for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    tables, labels = [], []
    for f in files:
        X, _, _ = native.parm_read(os.path.join(DEMO, "train", f))
        tables.append(oracle.parm_qualify(X, hasD=True))
        labels.append(native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab"))))
    pk, model, hist, converged = hinit(native, mmf, name, tables, labels, max_iter=10)
    lines = [l for l in open(os.path.join(gold, "hinit.log")) if l.startswith("HInit %s:" % name)]
    ref = [float(re.search(r"Average LogP = *(-?[\d.]+)", l).group(1)) for l in lines if "Average LogP" in l]
    assert len(hist) == len(ref), (hist, ref)
    for p, rp in zip(hist, ref):
        assert abs(p - rp) <= 1e-6 * abs(rp), (name, hist, ref)
    assert converged == any("converged" in l for l in lines)
    rmmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(gold, "hmm0"))
    rq, p = rmmf.packed(), model.get_params()
    q0 = mmf.packed()
    h, rh = mmf.logical[name], rmmf.logical[name]
    nmix = []
    for s, rs in zip(q0["hmmState"][q0["hmmStateOff"][h]:q0["hmmStateOff"][h + 1]], rq["hmmState"][rq["hmmStateOff"][rh]:rq["hmmStateOff"][rh + 1]]):
        c0, c1, r0, r1 = int(q0["stateCompOff"][s]), int(q0["stateCompOff"][s + 1]), int(rq["stateCompOff"][rs]), int(rq["stateCompOff"][rs + 1])
        assert c1 - c0 == r1 - r0
        nmix.append(c1 - c0)
        for c, rc in zip(range(c0, c1), range(r0, r1)):
            g, rg = int(q0["compGauss"][c]), int(rq["compGauss"][rc])
            sigma = np.sqrt(rq["var"][rg])
            assert abs(p["compWeight"][c] - rq["compWeight"][rc]) <= 1e-5
            assert (np.abs(p["mean"][g] - rq["mean"][rg]) <= 1e-4 * np.maximum(np.abs(rq["mean"][rg]), sigma) + 1e-6).all()
            assert np.allclose(p["var"][g], rq["var"][rg], rtol=1e-4, atol=1e-7)
    assert nmix == [2, 3, 2]
    t, rt = int(q0["hmmTrans"][h]), int(rq["hmmTrans"][rh])
    N = int(q0["transN"][t])
    lin = lambda v: np.where(v > -0.5e10, np.exp(v.astype(np.float64)), 0.0)
    assert np.allclose(lin(p["transP"][q0["transOff"][t]:q0["transOff"][t] + N * N]), lin(rq["transP"][rq["transOff"][rt]:rq["transOff"][rt] + N * N]), rtol=1e-4, atol=1e-7)


CHAIN_BAR_MEAN, CHAIN_BAR_VAR = 1e-3, 2e-3      # (the means never needed more than 1e-3 -- observed 8e-4; a variance 1.1e-3)


def test_htkdemo_training_chain_from_prototypes(native, oracle, tmp_path):
    """The whole training side of HTKDemo's monPlainM1S1 on the device, each stage fed by the previous one's OUTPUT FILES:
    prototypes -> HInit (Viterbi training per model) -> HRest (Baum-Welch per model) -> one embedded HERest pass; the result
    against the models the reference's own chain wrote (hmm2_expected).  Every stage ON ITS OWN -- fed with the reference's files -- is held to
    1e-4 above; here stage k reads OUR stage k-1's files, which differ from the reference's by up to that much, and HRest then iterates on
    them up to 20 times over seven files (a few hundred frames per state): observed 8e-4 on a mean, 1.1e-3 on a variance at the end of
    the chain (printed below).  The bars are 1e-3 on means and 2e-3 on variances: a stage that went wrong shows as percents, not as parts in ten thousand."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from examples.hinit_model import hinit
    from examples.hrest_model import hrest
    files = sorted(f for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    stat = [native.parm_read(os.path.join(DEMO, "train", f))[0] for f in files]
    tables = [oracle.parm_qualify(X, hasD=True) for X in stat]
    labels = [native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab"))) for f in files]
    d0, d1 = tmp_path / "hmm0", tmp_path / "hmm1"
    d0.mkdir(); d1.mkdir()
    for src, dst, step in ((os.path.join(DEMO, "proto"), d0, "hinit"), (str(d0), d1, "hrest")):
        for name in "SCVNL":
            mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=src)
            if step == "hinit":
                pk, model, hist, _ = hinit(native, mmf, name, tables, labels, max_iter=10)
            else:
                model, hist = hrest(native, mmf, name, tables, labels, max_iter=10, min_var=0.05, mix_weight_floor=3 * 1.0e-5)
            one = native.Mmf(files=[os.path.join(src, name)])                    # the model's own file, rewritten with the new values
            q, p = one.packed(), model.get_params()
            full = mmf.packed()
            h = mmf.logical[name]
            st = full["hmmState"][full["hmmStateOff"][h]:full["hmmStateOff"][h + 1]]
            g = [int(full["compGauss"][full["stateCompOff"][s]]) for s in st]
            t = int(full["hmmTrans"][h]); N = int(full["transN"][t]); o = int(full["transOff"][t])
            one.write(dict(mean=p["mean"][g], var=p["var"][g], gconst=p["gconst"][g], compWeight=q["compWeight"],
                           transP=p["transP"][o:o + N * N]), out_dir=str(dst))
    # embedded pass from the files just written
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=str(d1))
    model = native.Model(mmf.packed())
    seqs = [np.array([mmf.logical[n] for n, _, _, _ in labs], np.int32) for labs in labels]
    dX, frameOff, cols = native.parm_add_qualifiers(stat, hasD=True)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    fb = native.ForwardBackward(model); acc = native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, np.concatenate(seqs))
    fb.execute(native.fb_config(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0), acc)
    pr, st = fb.results()
    a = acc.download()
    assert abs(a["totalPr"] / a["totalT"] - (-5.900196e+01)) < 2e-3             # the reference's first-pass figure
    model.update(acc, a["vec"], minEgs=3, minVar=0.05, mixWeightFloor=3 * 1.0e-5)
    rmmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm2_expected"))
    rq, pk, p = rmmf.packed(), mmf.packed(), model.get_params()
    worst_m = worst_v = 0.0
    for name in "SCVNL":
        h, rh = mmf.logical[name], rmmf.logical[name]
        for k, (s, rs) in enumerate(zip(pk["hmmState"][pk["hmmStateOff"][h]:pk["hmmStateOff"][h + 1]], rq["hmmState"][rq["hmmStateOff"][rh]:rq["hmmStateOff"][rh + 1]])):
            g, rg = int(pk["compGauss"][pk["stateCompOff"][s]]), int(rq["compGauss"][rq["stateCompOff"][rs]])
            sigma = np.sqrt(rq["var"][rg])
            dm = float(np.max(np.abs(p["mean"][g] - rq["mean"][rg]) / np.maximum(np.abs(rq["mean"][rg]), sigma)))
            dv = float(np.max(np.abs(p["var"][g] - rq["var"][rg]) / rq["var"][rg]))
            assert dm <= CHAIN_BAR_MEAN and dv <= CHAIN_BAR_VAR, "model %s state %d: mean deviation %.3g, variance deviation %.3g" % (name, k + 2, dm, dv)
            worst_m, worst_v = max(worst_m, dm), max(worst_v, dv)
    print("chain: worst mean deviation %.3g (of max(|mean|, sigma)), worst variance deviation %.3g" % (worst_m, worst_v))


def test_htkdemo_recognition_matches_reference_label_files(native):
    """The demo's test step on the device: HVite -w monLattice -t 300.0 -p 5.0 -s 0.0 bcpvocab bcplist over the three test files
    and the seven training files with the demo's final models; every line of every .rec file the reference wrote (HResults on
    them gives the known answer of HTKDemo/results/monPlainM1S1.res: %Corr=63.91, Acc=59.40 on the test set)."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from decode_util import format_words
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm_final"))
    model = native.Model(mmf.packed())
    net = native.Net(os.path.join(DEMO, "monLattice"), os.path.join(DEMO, "bcpvocab"), mmf)
    dec = native.Decoder(model, net, lmScale=0.0)
    expected = json.load(open(os.path.join(DEMO, "hvite_expected.json")))
    n = 0
    for part in ("test", "train"):
        names = sorted(expected[part])
        stat = [native.parm_read(os.path.join(DEMO, part, u + ".mfc"))[0] for u in names]
        dX, frameOff, cols = native.parm_add_qualifiers(stat, hasD=True)
        feats = dX.to_host(np.float32, (int(frameOff[-1]), cols))
        res = dec.run([feats[frameOff[u]:frameOff[u + 1]] for u in range(len(names))], genBeam=300.0, lmScale=0.0, wordPen=5.0)
        for u, (words, total) in zip(names, res):
            got = format_words(words, net.out_syms)
            assert got == expected[part][u], (part, u)
            n += len(got)
    assert n == sum(len(v) for per in expected.values() for v in per.values()) == 292


def test_herest_pass_on_mixture_system_from_mixup(native, tmp_path):
    """Single-Gaussian demo models split to 3 (5 for one state) components with htkamd_mmf_mixup (= HHEd MU, byte-identical files:
    tests/test_mmf_labels.py), then one embedded pass: log-likelihood, floored variances and re-estimated mixture parameters
    against the reference's HERest run from the reference HHEd's output (tests/golden/demo/hmm_mixup, make_mixup_golden.py)."""
    mmf = native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm_final"))
    mmf.mixup(3)
    q1 = mmf.packed()
    h = mmf.logical["S"]
    mmf.mixup(-2, states=[int(q1["hmmState"][q1["hmmStateOff"][h]])])
    pk = mmf.packed()
    model = native.Model(pk)
    files = sorted(f for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    stat = [native.parm_read(os.path.join(DEMO, "train", f))[0] for f in files]
    seqs = [np.array([mmf.logical[n] for n, _, _, _ in native.labels_read(os.path.join(DEMO, "labels", f.replace(".mfc", ".lab")))], np.int32) for f in files]
    dX, frameOff, cols = native.parm_add_qualifiers(stat, hasD=True)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    fb = native.ForwardBackward(model); acc = native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, np.concatenate(seqs))
    fb.execute(native.fb_config(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0), acc)
    pr, st = fb.results()
    a = acc.download()
    log = open(os.path.join(DEMO, "hmm_mixup", "herest.log")).read()
    ref_avg = float(re.search(r"average log prob per frame = (\S+)", log).group(1))
    assert (st == 1).all() and "%e" % (a["totalPr"] / a["totalT"]) == "%e" % ref_avg
    stats = model.update(acc, a["vec"], minEgs=3, minVar=0.05, mixWeightFloor=3 * 1.0e-5)
    mm = re.search(r"Total (\d+) floored variance elements in (\d+) different mixes", log)
    assert (stats["nFloorVar"], stats["nFloorVarMix"]) == (int(mm.group(1)), int(mm.group(2)))
    rmmf = native.Mmf(files=[os.path.join(DEMO, "hmm_mixup", "after_herest")], hmm_list=os.path.join(DEMO, "bcplist"))
    rq, p = rmmf.packed(), model.get_params()
    for name in "SCVNL":
        hh, rh = mmf.logical[name], rmmf.logical[name]
        for s, rs in zip(pk["hmmState"][pk["hmmStateOff"][hh]:pk["hmmStateOff"][hh + 1]], rq["hmmState"][rq["hmmStateOff"][rh]:rq["hmmStateOff"][rh + 1]]):
            c0, c1, r0, r1 = int(pk["stateCompOff"][s]), int(pk["stateCompOff"][s + 1]), int(rq["stateCompOff"][rs]), int(rq["stateCompOff"][rs + 1])
            assert c1 - c0 == r1 - r0
            assert np.allclose(p["compWeight"][c0:c1], rq["compWeight"][r0:r1], rtol=1e-4, atol=2e-6), name
            for c, rc in zip(range(c0, c1), range(r0, r1)):
                g, rg = int(pk["compGauss"][c]), int(rq["compGauss"][rc])
                sigma = np.sqrt(rq["var"][rg])
                assert (np.abs(p["mean"][g] - rq["mean"][rg]) <= 1e-4 * np.maximum(np.abs(rq["mean"][rg]), sigma) + 1e-6).all(), name
                assert np.allclose(p["var"][g], rq["var"][rg], rtol=1e-4, atol=1e-6), name


def test_c_driver_runs_the_demo_pass(tmp_path):
    """examples/herest_pass.c -- a host written against include/htk_amd.h alone, compiled with gcc -- runs HTKDemo's first embedded
    pass from the model / parameter / label FILES and prints the reference's summary lines; its output models equal the
    reference's to the printed digits' tolerance."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = tmp_path / "herest_pass"
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "herest_pass.c"), "-o", str(exe),
                           "-L" + os.path.join(root, "htk_amd"), "-lhtk_amd", "-Wl,-rpath," + os.path.abspath(os.path.join(root, "htk_amd"))])
    out = tmp_path / "hmm2"; out.mkdir()
    files = sorted(os.path.join(DEMO, "train", f) for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    r = subprocess.run([str(exe), os.path.join(DEMO, "bcplist"), os.path.join(DEMO, "hmm1"), os.path.join(DEMO, "labels"), str(out), "2000.0", "0.05", "3"] + files,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    log = open(os.path.join(DEMO, "herest_pass1.log")).read()
    assert "average log prob per frame = -5.900196e+01" in r.stdout and "-5.900196e+01" in log
    assert "Total 27 floored variance elements in 15 different mixes" in r.stdout
    for name in "SCVNL":
        ours = (out / name).read_text().split()
        theirs = open(os.path.join(DEMO, "hmm2_expected", name)).read().split()
        assert len(ours) == len(theirs)
        for x, y in zip(ours, theirs):
            if x != y:
                assert abs(float(x) - float(y)) <= 1e-4 * max(abs(float(y)), 1e-3), (name, x, y)


def test_c_driver_recognises_the_demo_test_set(tmp_path):
    """examples/hvite_decode.c (C ABI only, gcc): the .rec files it writes for HTKDemo's test files are the reference HVite's."""
    import json
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = tmp_path / "hvite_decode"
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "hvite_decode.c"), "-o", str(exe),
                           "-L" + os.path.join(root, "htk_amd"), "-lhtk_amd", "-Wl,-rpath," + os.path.abspath(os.path.join(root, "htk_amd"))])
    out = tmp_path / "rec"; out.mkdir()
    expected = json.load(open(os.path.join(DEMO, "hvite_expected.json")))["test"]
    files = [os.path.join(DEMO, "test", u + ".mfc") for u in sorted(expected)]
    r = subprocess.run([str(exe), os.path.join(DEMO, "bcplist"), os.path.join(DEMO, "hmm_final"), os.path.join(DEMO, "monLattice"), os.path.join(DEMO, "bcpvocab"),
                        str(out), "300.0", "0.0", "5.0"] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    for u in expected:
        assert (out / (u + ".rec")).read_text().splitlines() == expected[u], u
