"""Live cross-check of the oracle against the reference build (oracle/_ref) on fresh seeds.  Runs only where the
reference was built (this container); the committed golden vectors cover the same ground elsewhere."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from util import REFDIR, eq_nan

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "ref_fbdump")),
                                reason="oracle/_ref not built (needs /root/reference)")


def run_ref(d, tflag=""):
    from oracle import refio
    scp = open(os.path.join(d, "train.scp")).read().split()
    r = subprocess.run("%s/ref_fbdump -C config %s -H hmm0/MMF -L lab -p 1 -M hmm1 hmmlist dump.bin %s"
                       % (REFDIR, tflag, " ".join(scp)), shell=True, cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return refio.read_fbdump(os.path.join(d, "dump.bin"))


@pytest.mark.parametrize("seed,tflag,prune", [
    (101, "", {}),
    (102, "-t 40.0", dict(pruneInit=40.0, pruneInc=0.0, pruneLim=40.0)),
    (103, "-t 0.5 2.0 12.0", dict(pruneInit=0.5, pruneInc=2.0, pruneLim=12.0)),     # forces retries of the beta pass
    (104, "-t 0.01", dict(pruneInit=0.01, pruneInc=0.0, pruneLim=0.01)),             # over-pruning: utterances skipped
])
def test_random_set(oracle, seed, tflag, prune):
    from htk_amd import synth
    from oracle import refio
    po = oracle
    with tempfile.TemporaryDirectory() as d:
        s = synth.generate(30, 3, 20, 5, 90, seed, outdir=d)
        ref = run_ref(d, tflag)
        pk = s.packed()
        m = po.Model(pk); acc = po.Accs(m); cfg = po.fb_cfg(**prune)
        for u in range(5):
            rc, pr, dd = po.fb_utt(m, cfg, s.feats[u], s.seqs[u], acc, dump=True)
            r = ref[u]
            assert rc == r["ok"]
            if not rc:
                continue
            assert pr == r["pr"]
            for k in ("qLo", "qHi", "aLo", "aHi"):
                assert np.array_equal(dd[k], r[k])
            b = dd["beta"].copy(); b[np.isnan(r["beta"])] = np.nan
            assert eq_nan(b, r["beta"]) and eq_nan(dd["alpha"], r["alpha"]) and eq_nan(dd["occ"], r["occ"])
        ra = refio.read_acc(os.path.join(d, "hmm1", "HER1.acc"), pk, ["p%d" % i for i in range(20)])
        for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs"):
            assert np.array_equal(np.asarray(getattr(acc, k)).reshape(-1), ra[k].reshape(-1)), k


def test_too_short_utterance_is_skipped(oracle):
    """qt > T: 'Unable to traverse' (HFB.c:1339-1344) -> FBFile returns FALSE, nothing accumulated."""
    from htk_amd import synth
    po = oracle
    with tempfile.TemporaryDirectory() as d:
        s = synth.generate(30, 2, 20, 2, 36, 7, outdir=d)          # Q = 3 models, min 9 frames
        # truncate the second utterance to 5 frames
        X = s.feats[1][:5]
        synth.write_htk_param(os.path.join(d, "data", "u00001.mfc"), X)
        ref = run_ref(d)
        assert ref[0]["ok"] == 1 and ref[1]["ok"] == 0
        m = po.Model(s.packed()); acc = po.Accs(m)
        rc, _, _ = po.fb_utt(m, po.fb_cfg(), X, s.seqs[1], acc)
        assert rc == 0 and acc.nEgs.sum() == 0


@pytest.mark.parametrize("kind", ["MFCC_0_D_A", "MFCC_E_D_A", "MFCC_E_D_A_Z", "MFCC_0"])
def test_mfcc_oracle_equals_hcopy(oracle, kind):
    """oracle/orc_mfcc.c vs the reference's HCopy on the SURVEY config-5 waveform: every float identical."""
    import wave
    from htk_amd import synth
    rng = np.random.default_rng(7); n = 48000; t = np.arange(n) / 16000
    x = (3000 * np.sin(2 * np.pi * 440 * t) * np.sin(2 * np.pi * 3 * t) + rng.normal(0, 800, n)).clip(-32768, 32767).astype("<i2")
    with tempfile.TemporaryDirectory() as d:
        w = wave.open(os.path.join(d, "t.wav"), "wb"); w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(x.tobytes()); w.close()
        open(os.path.join(d, "mfcc.conf"), "w").write(
            "SOURCEFORMAT = WAV\nSOURCERATE = 625\nWINDOWSIZE = 250000.0\nTARGETRATE = 100000.0\nNUMCHANS = 26\nNUMCEPS = 12\n"
            "CEPLIFTER = 22\nPREEMCOEF = 0.97\nUSEHAMMING = T\nTARGETKIND = %s\n" % kind)
        r = subprocess.run("%s/HCopy -C mfcc.conf t.wav t.mfc" % REFDIR, shell=True, cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        ref, period, k = synth.read_htk_param(os.path.join(d, "t.mfc"))
    mine = oracle.mfcc(x, oracle.mfcc_cfg(kind))
    assert period == 100000 and ref.shape == (298, mine.shape[1])
    assert np.array_equal(ref, mine)


def test_randomised_sweep_against_the_reference_binaries():
    """A short run of tests/fuzz_oracle_vs_ref.py: random model sets, dictionaries (pronunciation variants, probabilities, tee
    model), lattices, switches and pruning -- the reference's HVite label files and HERest traces vs the oracle."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "HVite")):
        pytest.skip("reference binaries not built (oracle/_ref)")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_oracle_vs_ref.py"), "12", "2024"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "ref_outp")), reason="oracle/_ref/ref_outp not built")
@pytest.mark.parametrize("seed,M,D", [(301, 4, 13), (302, 1, 26), (303, 16, 39)])
def test_soutp_and_doutp_forms_equal_the_reference_outp(oracle, seed, M, D):
    """What a direct caller of OutP gets (HModel.c:5503-5600), from the reference itself through oracle/ref_outp.c: the set as loaded
    (DIAGC variances -> DOutP's division, linear weights -> MixLogWeight on the fly, SOutP's double log-sum) and after
    ConvDiagC + ConvLogWt (IDOutP).  The oracle's orc_soutp_block must give the same floats, bit for bit, in both forms."""
    from htk_amd import synth
    with tempfile.TemporaryDirectory() as d:
        s = synth.generate(24, M, 12, 1, 60, seed, D=D, outdir=d)
        pk = s.packed()
        om = oracle.Model(pk)
        scp = open(os.path.join(d, "train.scp")).read().split()
        H = int(pk["numPhys"])
        states = np.asarray(pk["hmmState"], np.int32)                      # models in list order, 3 emitting states each
        assert len(states) == 3 * H
        X = s.feats[0]
        for flag, diagc in (("", True), ("-c", False)):
            out = os.path.join(d, "outp%s.bin" % flag)
            r = subprocess.run("%s/ref_outp %s -C config -H hmm0/MMF hmmlist %s %s" % (REFDIR, flag, scp[0], out), shell=True, cwd=d, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            ref = np.fromfile(out, np.float32).reshape(X.shape[0], 3 * H)
            got = om.soutp_block(X, states, diagc=diagc)
            assert np.array_equal(got, ref), (flag, float(np.abs(got - ref).max()))
