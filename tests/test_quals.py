"""Qualifier step of HParm on parameterised tables (_D _A _T _Z _N): oracle vs files written by the reference's HCopy / printed by
its HList (tests/golden/quals, generator make_quals_golden.py), and the HIP path vs the oracle."""
import os
import re

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _read(path):
    from htk_amd import capi
    X, period, kind = capi.parm_read(path)                     # host C reader: checks the _K CRC of HCopy's files
    return X, kind, period


@pytest.fixture(scope="module")
def stat():
    X, kind, period = _read(os.path.join(GOLD, "demo", "train", "tr1.mfc"))
    assert X.shape[1] == 13                                    # MFCC_E: 12 cepstra + log energy
    return X


CASES = [("MFCC_E_D_A_T", dict(hasD=True, hasA=True, hasT=True, thirdWin=3)),
         ("MFCC_E_D_A_T_Z", dict(hasD=True, hasA=True, hasT=True, thirdWin=3, nZeroMean=12)),
         ("MFCC_E_D_Z", dict(hasD=True, nZeroMean=12)),
         ("MFCC_E_D_A_V1", dict(hasD=True, hasA=True, delWin=3, v1Compat=True)),           # V1COMPAT = T, DELTAWINDOW = 3
         ("MFCC_E_D_A_SD", dict(hasD=True, hasA=True, delWin=3, simpleDiffs=True))]        # SIMPLEDIFFS = T


@pytest.mark.parametrize("kind,kw", CASES)
def test_oracle_matches_hcopy_bit_exact(oracle, stat, kind, kw):
    ref, _, _ = _read(os.path.join(GOLD, "quals", "tr1_%s.mfc" % kind))
    got = oracle.parm_qualify(stat, **kw)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def _hlist_obs(path):
    rows, cur = [], None
    for line in open(path):
        m = re.match(r"^(\d+):(.*)$", line)
        if m:
            cur = []; rows.append(cur); line = m.group(2)
        elif cur is None or line.startswith("-"):
            if line.startswith("-") and rows:
                cur = None
            continue
        cur.extend(float(x) for x in line.split())
    return np.array(rows, np.float32)


def test_oracle_null_energy_matches_hlist(oracle, stat):
    """_N: the observation is the row without the absolute energy (38 of 39 components); HList prints 3 decimals."""
    ref = _hlist_obs(os.path.join(GOLD, "quals", "tr1_MFCC_E_D_A_N.hlist"))
    got = oracle.parm_qualify(stat, hasD=True, hasA=True, nullECol=12)
    assert ref.shape == (40, 38) and got.shape == (stat.shape[0], 38)
    assert np.abs(got[:40] - ref).max() <= 5.1e-4
    full = oracle.parm_qualify(stat, hasD=True, hasA=True)
    assert np.array_equal(got, np.delete(full, 12, axis=1))


def test_quals_from_kind(native):
    from htk_amd import capi
    q = capi.parm_quals_from_kind("MFCC_E_D_A_N", 13)
    assert (q.nullECol, q.nZeroMean, q.hasD, q.hasA, q.hasT) == (12, 0, 1, 1, 0)
    q = capi.parm_quals_from_kind("MFCC_0_D_A_T_Z", 13)
    assert (q.nullECol, q.nZeroMean, q.hasT) == (-1, 13, 1)               # C0 is zero-meaned with the cepstra (HParm.c:1714)
    q = capi.parm_quals_from_kind("MFCC_E_D_Z", 13)
    assert q.nZeroMean == 12
    assert capi.lib().htkamd_parm_quals_cols(capi.C.byref(capi.parm_quals_from_kind("MFCC_E_D_A_N", 13))) == 38


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["MFCC_E_D_A_T", "MFCC_E_D_A_T_Z", "MFCC_E_D_Z", "MFCC_E_D_A_N", "MFCC_E_D_N_Z", "MFCC_E"])
def test_device_qualify_bit_exact(native, oracle, stat, kind):
    """htkamd_parm_qualify over a ragged batch (whole file, 7 frames, 1 frame, 3 frames) == oracle, bit for bit."""
    from htk_amd import capi
    utts = [stat, stat[5:12], stat[40:41], stat[100:103]]
    q = capi.parm_quals_from_kind(kind, 13, thirdWin=3)
    d, frameOff, cols = capi.parm_qualify(utts, q)
    got = d.to_host(np.float32, (int(frameOff[-1]), cols))
    for u, x in enumerate(utts):
        ref = oracle.parm_qualify(x, nZeroMean=q.nZeroMean, hasD=bool(q.hasD), hasA=bool(q.hasA), hasT=bool(q.hasT), thirdWin=3, nullECol=q.nullECol)
        assert np.array_equal(got[frameOff[u]:frameOff[u + 1]], ref), (kind, u)
    if kind in ("MFCC_E_D_A_T", "MFCC_E_D_A_T_Z", "MFCC_E_D_Z"):
        ref, _, _ = _read(os.path.join(GOLD, "quals", "tr1_%s.mfc" % kind))
        assert np.array_equal(got[:stat.shape[0]], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", [("V1", dict(v1Compat=True)), ("SD", dict(simpleDiffs=True)), ("V1SD", dict(v1Compat=True, simpleDiffs=True))])
def test_device_difference_variants(native, oracle, stat, tag, kw):
    """V1COMPAT / SIMPLEDIFFS on the device == oracle bit for bit (ragged batch incl. tables shorter than 2w+1 rows), and == the
    file HCopy wrote where there is one."""
    from htk_amd import capi
    utts = [stat, stat[5:12], stat[40:41], stat[100:106], stat[7:14]]
    q = capi.parm_quals_from_kind("MFCC_E_D_A", 13, delWin=3, **kw)
    d, frameOff, cols = capi.parm_qualify(utts, q)
    got = d.to_host(np.float32, (int(frameOff[-1]), cols))
    for u, x in enumerate(utts):
        ref = oracle.parm_qualify(x, hasD=True, hasA=True, delWin=3, **kw)
        assert np.array_equal(got[frameOff[u]:frameOff[u + 1]], ref), (tag, u)
    if tag in ("V1", "SD"):
        ref, _, _ = _read(os.path.join(GOLD, "quals", "tr1_MFCC_E_D_A_%s.mfc" % tag))
        assert np.array_equal(got[:stat.shape[0]], ref)


@pytest.mark.gpu
def test_device_qualify_rejects(native):
    from htk_amd import capi
    x = [np.zeros((4, 13), np.float32)]
    for bad in (capi.ParmQuals(13, 0, 0, 1, 0, 2, 2, 2, -1, 0, 0),      # _A without _D
                capi.ParmQuals(13, 0, 1, 0, 1, 2, 2, 2, -1, 0, 0),      # _T without _A
                capi.ParmQuals(13, 0, 0, 0, 0, 2, 2, 2, 12, 0, 0),      # _N without _D
                capi.ParmQuals(13, 14, 1, 0, 0, 2, 2, 2, -1, 0, 0)):    # more zero-mean columns than statics
        with pytest.raises(capi.HtkAmdError):
            capi.parm_qualify(x, bad)
@pytest.mark.gpu
@pytest.mark.parametrize("kind,kw", [("MFCC_E_D_A", {}), ("MFCC_E_D_A_T", dict(delWin=3, accWin=2, thirdWin=1)), ("MFCC_0_D_N", dict(delWin=2)), ("MFCC_E_D", dict(simpleDiffs=True))])
def test_parm_stream_equals_table_mode(native, oracle, kind, kw):
    """HParm's buffer mode (FillBufFromChannel HParm.c:4000-4116): rows pushed in ragged chunks come out, qwin rows late, with exactly the
    values of the table-mode qualifier step over the whole utterance (and of the oracle), incl. utterances shorter than the windows."""
    rng = np.random.default_rng(11)
    for T in (1, 2, 5, 37, 211):
        X = rng.normal(size=(T, 13)).astype(np.float32)
        q = native.parm_quals_from_kind(kind, 13, **kw)
        d, frameOff, cols = native.parm_qualify([X], q)
        table = d.to_host(np.float32, (T, cols))
        st = native.ParmStream(q, 32)
        out, pos = [], 0
        while pos < T:
            n = int(rng.integers(0, 33))
            n = min(n, T - pos)
            out.append(st.push(X[pos:pos + n], last=(pos + n == T)))
            pos += n
            if pos < T:
                assert sum(len(o) for o in out) == max(pos - st.lookahead, 0)
        got = np.concatenate(out)
        assert got.shape == table.shape and np.array_equal(got, table), (kind, T)
        st.close()
