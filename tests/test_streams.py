"""Multi-stream sets (the S > 1 branches of HFB.c:1026-1066,1499-1602) -- what runs without a GPU:
  * the split of the observation into streams (SetStreamWidths / ExtractObservation, HParm.c:3094,2843) as a map dimension -> stream
  * MMF text in / out: the reference's HHEd output read and written back byte for byte (GetStateInfo / PutStateInfo HModel.c:1924,3053)
  * the oracle's restatement of Setotprob / UpMixParms with S > 1 against the reference's own `HERest -p 1` accumulators, EVERY float:
    S = 3, and S = 2 where the reference's second-visit branch (HFB.c:1044,1059) changes the numbers -- restated as it is."""
import os

import numpy as np
import pytest

import streams_util as su

DEMO = su.DEMO


def _stream_dims(native, kind, D, widths):
    import ctypes as C
    out = np.zeros(D, np.int32); why = C.create_string_buffer(200)
    w = np.array(widths, np.int32)
    rc = native.lib().htkamd_host_stream_dims(kind.encode(), C.c_int(D), C.c_int(len(widths)), w.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), why, C.c_size_t(200))
    return rc, out, why.value.decode()


def test_stream_split_follows_the_parameter_kind(native):
    # MFCC_E_D, 26 = (12 + E) x 2: the standard 3-way split takes the energies out into the last stream; HHEd's 13 | 13 is not the
    # standard 2-way split (24 | 2), so the streams are consecutive pieces
    rc, d, _ = _stream_dims(native, "MFCC_E_D", 26, [12, 12, 2])
    assert rc == 0 and list(d) == [0] * 12 + [2] + [1] * 12 + [2]
    rc, d, _ = _stream_dims(native, "MFCC_E_D", 26, [13, 13])
    assert rc == 0 and list(d) == [0] * 13 + [1] * 13
    rc, d, _ = _stream_dims(native, "MFCC_E_D", 26, [24, 2])
    assert rc == 0 and list(d) == [0] * 12 + [1] + [0] * 12 + [1]
    rc, d, _ = _stream_dims(native, "MFCC_E_D_A", 39, [12, 12, 12, 3])
    assert rc == 0 and list(d) == ([0] * 12 + [3]) + ([1] * 12 + [3]) + ([2] * 12 + [3])
    rc, d, _ = _stream_dims(native, "MFCC_E_D_A", 39, [13, 13, 13])
    assert rc == 0 and list(d) == [0] * 13 + [1] * 13 + [2] * 13
    rc, d, _ = _stream_dims(native, "USER", 10, [4, 6])
    assert rc == 0 and list(d) == [0] * 4 + [1] * 6
    rc, _, why = _stream_dims(native, "MFCC_E_D", 26, [12, 12, 3])
    assert rc != 0 and "27" in why


@pytest.mark.parametrize("S", [3, 2])
def test_mmf_with_streams_round_trips(native, tmp_path, S):
    src = os.path.join(DEMO, "hmm_streams%d" % S, "newMacros")
    mmf = native.Mmf(files=[src], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    assert pk["numStreams"] == S and pk["numStates"] == 15 and len(pk["stateCompOff"]) == 15 * S + 1
    gs = np.zeros(pk["numGauss"], np.int32)
    for e in range(15 * S):
        gs[pk["compGauss"][pk["stateCompOff"][e]:pk["stateCompOff"][e + 1]]] = e % S
    inside = pk["dimStream"][None, :] == gs[:, None]
    assert np.isinf(pk["var"][~inside]).all() and (pk["mean"][~inside] == 0).all() and np.isfinite(pk["var"][inside]).all()
    out = str(tmp_path / "newMacros")
    mmf.write(pk, one_file=out)
    assert open(out).read() == open(src).read()
    outb = str(tmp_path / "bin")
    mmf.write(pk, one_file=outb, binary=True)
    p2 = native.Mmf(files=[outb], hmm_list=os.path.join(DEMO, "bcplist")).packed()
    for k in ("mean", "var", "compWeight", "stateCompOff", "compGauss", "dimStream"):
        assert np.array_equal(pk[k], p2[k]), k
    # a Gaussian shared between streams, or a set that is multi-stream in the options only, is refused with a message
    bad = open(src).read().replace("<STREAMINFO> %d" % S, "<STREAMINFO> %d" % (S + 1), 1)
    (tmp_path / "bad").write_text(bad)
    with pytest.raises(native.HtkAmdError):
        native.Mmf(files=[str(tmp_path / "bad")], hmm_list=os.path.join(DEMO, "bcplist"))


@pytest.mark.parametrize("S", [3, 2])
def test_oracle_streams_equal_the_reference_accumulators(native, oracle, S):
    d = os.path.join(DEMO, "hmm_streams%d" % S)
    mmf = native.Mmf(files=[os.path.join(d, "newMacros")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    om = oracle.Model(pk)                                         # the reference's arithmetic, second-visit branch included
    acc = oracle.Accs(om)
    tot, T = 0.0, 0
    for u in su.demo_utterances(native, oracle, mmf):
        rc, pr, _ = oracle.fb_utt(om, oracle.fb_cfg(pruneInit=2000.0), u["feat"], u["seq"], acc)
        assert rc == 1
        tot += pr; T += len(u["feat"])
    log = open(os.path.join(d, "herest.log")).read()
    assert "average log prob per frame = %e" % (tot / T) in log
    lay = native.accs_layout(pk)
    v = np.zeros(lay.total, np.float64)
    native.accs_load_file(pk, v, list(mmf.phys_names), os.path.join(d, "HER1.acc"))
    for k, a in (("mu", acc.mu.reshape(-1)), ("muOcc", acc.muOcc), ("va", acc.va.reshape(-1)), ("vaOcc", acc.vaOcc), ("wt", acc.wt), ("wtOcc", acc.wtOcc),
                 ("tr", acc.tr), ("trOcc", acc.trOcc)):
        o = getattr(lay, k)
        assert np.array_equal(a.astype(np.float32), v[o:o + a.size].astype(np.float32)), k
    # our writer reproduces the file
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        native.accs_dump_file(pk, v, list(mmf.phys_names), os.path.join(td, "HER1.acc"))
        assert open(os.path.join(td, "HER1.acc"), "rb").read() == open(os.path.join(d, "HER1.acc"), "rb").read()
    if S == 2:
        # the intended arithmetic (every visit like the first) is another number altogether: S = 2 is not usable in the reference
        om2 = oracle.Model(pk, ms_intended=True)
        acc2 = oracle.Accs(om2)
        tot2 = sum(oracle.fb_utt(om2, oracle.fb_cfg(pruneInit=2000.0), u["feat"], u["seq"], acc2)[1] for u in su.demo_utterances(native, oracle, mmf))
        assert abs(tot2 / T - (-59.08)) < 0.01 and abs(tot / T - (-33.645)) < 0.01


# ------------------------------------------------------------------------------------------------ tied mixtures (hsKind TIEDHS)
TMIX = os.path.join(DEMO, "hmm_tmix")


@pytest.mark.parametrize("kind", ["tiedhs", "tiedhs3"])
def test_tmix_mmf_round_trips(native, tmp_path, kind):
    src = os.path.join(TMIX, kind + "_newMacros")
    mmf = native.Mmf(files=[src], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    S = 3 if kind == "tiedhs3" else 1
    assert pk["hsKind"] == 1 and pk["numStreams"] == S and pk["numGauss"] == (10 if S == 3 else 8)
    # every (state, stream) lists its stream's pool
    sco, cg = pk["stateCompOff"], pk["compGauss"]
    for e in range(15 * S):
        assert np.array_equal(cg[sco[e]:sco[e + 1]], cg[sco[e % S]:sco[e % S + 1]])
    out = str(tmp_path / "newMacros")
    mmf.write(pk, one_file=out)
    assert open(out).read() == open(src).read()                      # <TMIX> "name" and the run-length weights as PutTiedWeights writes them
    outb = str(tmp_path / "bin")
    mmf.write(pk, one_file=outb, binary=True)
    p2 = native.Mmf(files=[outb], hmm_list=os.path.join(DEMO, "bcplist")).packed()
    assert np.array_equal(pk["mean"], p2["mean"]) and np.array_equal(pk["compGauss"], p2["compGauss"])
    assert np.allclose(pk["compWeight"], p2["compWeight"], rtol=0, atol=2e-7)     # binary runs are stored as weight - 2 (HModel.c:2602): lossy by design


@pytest.mark.parametrize("kind", ["tiedhs", "tiedhs3"])
def test_oracle_tied_mixtures_equal_the_reference_accumulators(native, oracle, kind):
    """PrecomputeTMix / SOutP / UpMixParms' TIEDHS branches as restated in oracle/htk_oracle.c against the reference's `HERest -p 1`
    accumulator file: every float, the summary line, and our reader / writer of the file (pool records after the last model)."""
    mmf = native.Mmf(files=[os.path.join(TMIX, kind + "_newMacros")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = mmf.packed()
    om = oracle.Model(pk)
    acc = oracle.Accs(om)
    tot, T = 0.0, 0
    for u in su.demo_utterances(native, oracle, mmf):
        rc, pr, _ = oracle.fb_utt(om, oracle.fb_cfg(pruneInit=2000.0), u["feat"], u["seq"], acc)
        assert rc == 1
        tot += pr; T += len(u["feat"])
    assert "average log prob per frame = %e" % (tot / T) in open(os.path.join(TMIX, kind + ".log")).read()
    lay = native.accs_layout(pk)
    v = np.zeros(lay.total, np.float64)
    ref = os.path.join(TMIX, kind + "_HER1.acc")
    native.accs_load_file(pk, v, list(mmf.phys_names), ref)
    for k, a in (("mu", acc.mu.reshape(-1)), ("muOcc", acc.muOcc), ("va", acc.va.reshape(-1)), ("vaOcc", acc.vaOcc), ("wt", acc.wt), ("wtOcc", acc.wtOcc),
                 ("tr", acc.tr), ("trOcc", acc.trOcc)):
        o = getattr(lay, k)
        assert np.array_equal(a.astype(np.float32), v[o:o + a.size].astype(np.float32)), k
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        native.accs_dump_file(pk, v, list(mmf.phys_names), os.path.join(td, "HER1.acc"))
        assert open(os.path.join(td, "HER1.acc"), "rb").read() == open(ref, "rb").read()
