"""HCompV (flat start): global mean / variance and the variance floor macro.  Oracle (float sums in file order, HCompV.c:392,261)
vs the model and vFloors the reference's HCompV wrote for HTKDemo's training files (tests/golden/compv, generator
make_compv_golden.py); HIP path (fp64 sums) vs both."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _table(oracle):
    from htk_amd import capi
    xs = []
    for f in sorted(glob.glob(os.path.join(GOLD, "demo", "train", "tr*.mfc"))):
        X, _, _ = capi.parm_read(f)
        xs.append(oracle.parm_qualify(X, hasD=True))            # TARGETKIND = MFCC_E_D
    return xs


def _reference(native):
    m = native.Mmf(files=[os.path.join(GOLD, "compv", "S")])
    q = m.packed()
    toks = open(os.path.join(GOLD, "compv", "vFloors")).read().split()
    assert toks[:4] == ["~v", "varFloor1", "<Variance>", "26"]
    return q, np.array([float(x) for x in toks[4:]], np.float32)


def test_oracle_matches_hcompv(native, oracle):
    """Every state of the reference's output carries the global mean and variance; the oracle reproduces the printed values."""
    q, vfl = _reference(native)
    mean, var = oracle.compv(np.concatenate(_table(oracle)))
    assert q["mean"].shape == (3, 26)
    for g in range(3):
        assert np.allclose(q["mean"][g], mean, rtol=6e-7, atol=1e-10)      # "%e": 7 significant digits
        assert np.allclose(q["var"][g], var, rtol=6e-7)
    assert np.allclose(vfl, var * np.float32(0.01), rtol=6e-7)


def test_vfloors_writer_matches_hcompv(native, oracle, tmp_path):
    mean, var = oracle.compv(np.concatenate(_table(oracle)))
    out = tmp_path / "vFloors"
    native.write_vfloors(str(out), var, 0.01)
    assert out.read_text() == open(os.path.join(GOLD, "compv", "vFloors")).read()


@pytest.mark.gpu
def test_device_compv(native, oracle):
    """fp64 sums on the device vs the reference's float sums: equal to the float accumulators' own rounding (1811 frames)."""
    xs = _table(oracle)
    X = np.concatenate(xs)
    d = native.DevArray(X)
    mean, var = native.compv(d.ptr, X.shape[0], X.shape[1])
    omean, ovar = oracle.compv(X)
    q, vfl = _reference(native)
    assert np.allclose(mean, omean, rtol=1e-5, atol=1e-6) and np.allclose(var, ovar, rtol=1e-5)
    assert np.allclose(mean, q["mean"][0], rtol=1e-5, atol=1e-6) and np.allclose(var, q["var"][0], rtol=1e-5)
    exact_m = X.astype(np.float64).mean(0); exact_v = (X.astype(np.float64) ** 2).mean(0) - exact_m ** 2
    assert np.allclose(mean, exact_m, rtol=1e-6, atol=1e-7) and np.allclose(var, exact_v, rtol=1e-6)
    # variance floor and argument checks
    _, vf = native.compv(d.ptr, X.shape[0], X.shape[1], minVar=1.0)
    assert (vf >= 1.0).all() and (vf[ovar > 1.0] == var[ovar > 1.0]).all()
    with pytest.raises(native.HtkAmdError):
        native.compv(d.ptr, 1, X.shape[1])


@pytest.mark.gpu
def test_flat_start_pass_matches_reference(native, oracle):
    """The flat-start recipe end to end on the device: global statistics (htkamd_compv) put into every state of the five
    monophones, the variance floor macro 0.01*var, then one embedded re-estimation pass -- against the reference's
    HCompV -f 0.01 -m + HERest -t 2000 run from the same files (flat_hmm0.mmf -> flat_hmm1_expected.mmf, log line)."""
    import re
    demo = os.path.join(GOLD, "demo")
    stat, seqs = [], []
    mmf = native.Mmf(files=[os.path.join(GOLD, "compv", "flat_hmm0.mmf")], hmm_list=os.path.join(demo, "bcplist"))
    pk = mmf.packed()
    for f in sorted(glob.glob(os.path.join(demo, "train", "tr*.mfc"))):
        X, _, _ = native.parm_read(f)
        stat.append(X)
        labs = native.labels_read(os.path.join(demo, "labels", os.path.basename(f).replace(".mfc", ".lab")))
        seqs.append(np.array([mmf.logical[n] for n, _, _, _ in labs], np.int32))
    dX, frameOff, cols = native.parm_qualify(stat, native.parm_quals_from_kind("MFCC_E_D", 13))
    # the flat-start model from the device statistics = the one the reference's HCompV wrote (to its float rounding)
    mean, var = native.compv(dX.ptr, int(frameOff[-1]), cols)
    assert np.allclose(pk["mean"], mean[None, :], rtol=1e-5, atol=1e-6) and np.allclose(pk["var"], var[None, :], rtol=1e-5)
    assert np.allclose(mmf.var_floor, 0.01 * var, rtol=1e-5)
    model = native.Model(pk)
    fb = native.ForwardBackward(model); acc = native.Accs(model)
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    fb.prepare(dX.ptr.value, frameOff, labOff, np.concatenate(seqs))
    fb.execute(native.fb_config(pruneInit=2000.0, pruneInc=0.0, pruneLim=2000.0), acc)
    pr, st = fb.results()
    a = acc.download()
    log = open(os.path.join(GOLD, "compv", "flat_herest.log")).read()
    ref_avg = float(re.search(r"average log prob per frame = (\S+)", log).group(1))
    assert (st == 1).all() and "%e" % (a["totalPr"] / a["totalT"]) == "%e" % ref_avg
    # HERest -s: the occupation statistics file (index, name, examples, occupation of each emitting state)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        native.stats_write_file(pk, a["vec"], mmf.phys_names, os.path.join(d, "stats"))
        ours = [l.split() for l in open(os.path.join(d, "stats"))]
        raw = open(os.path.join(d, "stats")).read()
    theirs = [l.split() for l in open(os.path.join(GOLD, "compv", "flat_stats"))]
    assert len(ours) == len(theirs) == 5 and raw.startswith('   1 ')
    for o, t in zip(ours, theirs):
        assert o[:3] == t[:3] and len(o) == len(t) == 6
        assert np.allclose([float(x) for x in o[3:]], [float(x) for x in t[3:]], rtol=1e-4)
    model.update(acc, a["vec"], minEgs=3, varFloor=mmf.var_floor)
    rmmf = native.Mmf(files=[os.path.join(GOLD, "compv", "flat_hmm1_expected.mmf")], hmm_list=os.path.join(demo, "bcplist"))
    ref = rmmf.packed()
    p = model.get_params()

    def gaussians(q, h):                                        # Gaussians of physical model h, state by state (one per state here)
        states = q["hmmState"][q["hmmStateOff"][h]:q["hmmStateOff"][h + 1]]
        return [int(q["compGauss"][q["stateCompOff"][s]]) for s in states]

    for name in mmf.logical:                                    # the reference writes the models in its hash-table order
        for g, rg in zip(gaussians(pk, mmf.logical[name]), gaussians(ref, rmmf.logical[name])):
            sigma = np.sqrt(ref["var"][rg])
            assert (np.abs(p["mean"][g] - ref["mean"][rg]) <= 1e-4 * np.maximum(np.abs(ref["mean"][rg]), sigma) + 1e-6).all(), name
            assert np.allclose(p["var"][g], ref["var"][rg], rtol=1e-4, atol=1e-7), name
    assert (p["var"] >= mmf.var_floor[None, :] * (1 - 1e-6)).all()
