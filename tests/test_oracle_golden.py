"""The CPU oracle (oracle/htk_oracle.c) against golden vectors produced by the reference itself
(tests/golden/make_golden.py drives oracle/_ref: the reference's own HFB.c / HERest).  Bit-exact."""
import numpy as np
import pytest

from util import eq_nan, load_case, mmf_fmt, trans_as_saved

CASES = ["fb_small", "fb_small_prune", "fb_topo", "fb_topo_prune"]


@pytest.mark.parametrize("name", CASES)
def test_forward_backward_bit_exact(oracle, name):
    po = oracle
    case = load_case(name)
    m = po.Model(case["pk"])
    acc = po.Accs(m)
    cfg = po.fb_cfg(**case["prune"])
    for u in case["utts"]:
        rc, pr, d = po.fb_utt(m, cfg, u["feat"], u["seq"], acc, dump=True)
        assert rc == u["ok"] == 1
        assert pr == float(u["pr"])                     # LogDouble, bit for bit
        for k in ("qLo", "qHi", "aLo", "aHi"):
            assert np.array_equal(d[k], u[k]), k
        if "beta" in u:
            b = d["beta"].copy(); b[np.isnan(u["beta"])] = np.nan     # the dump of the reference covers its final beam only
            assert eq_nan(b, u["beta"])
            assert eq_nan(d["alpha"], u["alpha"])
            o = d["outp"].copy(); o[np.isnan(u["outp"])] = np.nan
            assert eq_nan(o, u["outp"])
            assert eq_nan(d["occ"], u["occ"])
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs"):
        assert np.array_equal(np.asarray(getattr(acc, k)).reshape(-1), case["acc"][k].reshape(-1)), k


@pytest.mark.parametrize("name", CASES)
def test_update_matches_reference_mmf(oracle, name):
    """HERest -m 1 (single process) wrote hmm1/MMF; the oracle's update must print the same 7 digits."""
    po = oracle
    case = load_case(name)
    m = po.Model(case["pk"])
    acc = po.Accs(m)
    cfg = po.fb_cfg(**case["prune"])
    for u in case["utts"]:
        po.fb_utt(m, cfg, u["feat"], u["seq"], acc)
    po.update(m, acc, minEgs=1, singleProcess=True)
    upd = case["upd"]
    for k, mine in (("mean", m.mean), ("var", m.var), ("compWeight", m.compWeight), ("gconst", m.gconst)):
        ref = np.asarray(upd[k], np.float32).reshape(-1)
        ok = ~np.isnan(ref)                               # components dropped from the file (weight 0) and absent <GCONST>
        assert np.array_equal(mmf_fmt(mine)[ok], ref[ok]), k
    saved = trans_as_saved(m.transP, m.transN, m.transOff)
    assert np.array_equal(mmf_fmt(saved), np.asarray(upd["transLin"], np.float32))


def test_known_answers_1k_x_8(oracle):
    """per-frame log probabilities the reference prints for the 1k x 8 set, seed 1."""
    from htk_amd import synth
    po = oracle
    known = np.load(__import__("os").path.join(__import__("util").GOLDEN, "c2_known.npz"))["per_frame"]
    assert np.allclose(known, [-61.24135, -61.13501, -60.78280], atol=1e-5)
    s = synth.generate(1000, 8, 2000, 3, 500, 1)
    m = po.Model(s.packed()); acc = po.Accs(m); cfg = po.fb_cfg()
    for u in range(3):
        rc, pr, _ = po.fb_utt(m, cfg, s.feats[u], s.seqs[u], acc)
        assert rc == 1
        assert float("%e" % (pr / 500)) == float("%e" % known[u])


def test_ladd_matches_libm(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        x, y = rng.uniform(-400, 0, 2)
        hi, lo = max(x, y), min(x, y)
        want = hi if lo - hi < -np.log(1e10) else hi + np.log(1.0 + np.exp(lo - hi))
        assert L.orc_ladd(x, y) == want
    assert L.orc_ladd(-1e10, -400.0) == -400.0          # adding log(0) changes nothing
    assert L.orc_ladd(-0.6e10, -2e10) == -1e10          # sums below LSMALL are floored to LZERO (HMath.c:1585)


def test_state_outp_variants_differ_only_in_rounding(oracle):
    """SOutP (double accumulation) and ShStrP/cSOutP (float after every component) are different roundings of the
    same quantity: equal to ~1 float ulp, not always bit-equal."""
    po = oracle
    case = load_case("fb_small")
    m = po.Model(case["pk"])
    X = case["utts"][0]["feat"]
    a = np.array([[m.state_outp(s, X[t]) for s in range(20)] for t in range(40)])
    b = np.array([[m.soutp(s, X[t]) for s in range(20)] for t in range(40)])
    assert np.allclose(a, b, rtol=3e-7)


def _rec_lines(z, key):
    return [l for l in str(z[key]).split("\n") if l.strip()]


def test_viterbi_alignment_identical_to_hvite(oracle):
    """Label files of the reference's `HVite -a -f -m` (tests/golden/hvite_rec.npz) reproduced line for line:
    times, state/model labels and the 6-decimal scores (HRec.c token passing restated in oracle/orc_viterbi.c)."""
    import os
    import util
    from htk_amd import synth
    po = oracle
    z = np.load(os.path.join(util.GOLDEN, "hvite_rec.npz"))
    s = synth.generate(60, 4, 40, 4, 120, 5)
    m = po.Model(s.packed()); names = ["p%d" % i for i in range(40)]
    for tag, beam in (("small", 1.0e10), ("small_t40", 40.0)):
        for u in range(4):
            r = po.viterbi_align(m, s.feats[u], s.seqs[u], genBeam=beam)
            assert po.format_rec(r, s.seqs[u], names) == _rec_lines(z, "%s_%d" % (tag, u))
    pk, names, seqs, feats = synth.make_topo_set()
    m = po.Model(pk)
    for u in range(len(seqs)):
        q = np.array([h for h in seqs[u] if h != 2], np.int32)
        r = po.viterbi_align(m, feats[u], q)
        assert po.format_rec(r, q, names) == _rec_lines(z, "topo_%d" % u)
