"""Waveform files -> MFCC: the host readers (RIFF/WAVE and HTK WAVEFORM) and the whole front end against the file the reference's
HCopy coded from the same sources (tests/golden/wave, generator make_wave_golden.py)."""
import os
import struct

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wave")


def _expected():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_wave_golden import test_wave
    return test_wave()


def test_wave_readers(native, tmp_path):
    x = _expected()
    a, per = native.wave_read(os.path.join(GOLD, "test.wav"), native.WAVE_WAV)
    b, perb = native.wave_read(os.path.join(GOLD, "test.htk"), native.WAVE_HTK)
    assert np.array_equal(a, x) and np.array_equal(b, x) and per == 625.0 and perb == 625.0
    # extra chunks before "data" are skipped (GetWAVHeaderInfo walks the chunks); stereo / 8-bit are refused
    raw = open(os.path.join(GOLD, "test.wav"), "rb").read()
    i = raw.index(b"data")
    odd = raw[:i] + b"LIST" + struct.pack("<I", 6) + b"abcdef" + raw[i:]
    p = tmp_path / "list.wav"; p.write_bytes(odd)
    c, _ = native.wave_read(str(p), native.WAVE_WAV)
    assert np.array_equal(c, x)
    stereo = bytearray(raw); stereo[22:24] = struct.pack("<H", 2)
    q = tmp_path / "stereo.wav"; q.write_bytes(bytes(stereo))
    for bad, fmt in ((str(q), native.WAVE_WAV), (os.path.join(GOLD, "test.htk"), native.WAVE_WAV), (os.path.join(GOLD, "test.wav"), native.WAVE_HTK),
                     (str(tmp_path / "missing.wav"), native.WAVE_WAV)):
        with pytest.raises(native.HtkAmdError):
            native.wave_read(bad, fmt)


def test_oracle_front_end_matches_hcopy_file(native, oracle):
    ref, period, kind = native.parm_read(os.path.join(GOLD, "test_MFCC_0_D_A.mfc"))
    x, _ = native.wave_read(os.path.join(GOLD, "test.wav"))
    got = oracle.mfcc(x, oracle.mfcc_cfg("MFCC_0_D_A"))
    assert ref.shape == (98, 39) and period == 100000
    assert np.array_equal(got, ref)                              # every float of the reference's output


@pytest.mark.gpu
def test_device_front_end_matches_hcopy_file(native):
    ref, _, _ = native.parm_read(os.path.join(GOLD, "test_MFCC_0_D_A.mfc"))
    x, _ = native.wave_read(os.path.join(GOLD, "test.wav"))
    got, frameOff = native.Mfcc(native.mfcc_config("MFCC_0_D_A")).compute_host([x])
    assert got.shape == ref.shape
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-3) and (got == ref).mean() > 0.999
