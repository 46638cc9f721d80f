"""HTK parameter files through the host C reader/writer (htk_amd/host/parmfile.c) against files written by the
reference's HCopy (tests/golden/parm/: SAVECOMPRESSED/SAVEWITHCRC variants of the SURVEY config-5 waveform)."""
import os

import numpy as np
import pytest

from util import GOLDEN

P = os.path.join(GOLDEN, "parm")
MFCC_E_D_A = 6 | 0o100 | 0o400 | 0o1000


def test_reads_crc_and_compressed_files(native):
    k, per, kind = native.parm_read(os.path.join(P, "mfcc_e_d_a_K.mfc"))
    c, per2, kind2 = native.parm_read(os.path.join(P, "mfcc_e_d_a_C_K.mfc"))
    d, per3, kind3 = native.parm_read(os.path.join(P, "mfcc_e_d_a_decompressed_by_ref.mfc"))
    assert k.shape == c.shape == d.shape == (298, 39) and per == per2 == per3 == 100000
    assert kind == kind2 == kind3 == MFCC_E_D_A                       # _C and _K are stripped
    assert np.array_equal(c, d)                                       # (short + B)/A exactly as the reference decompresses
    assert np.abs(c - k).max() < 1e-3                                 # 16-bit quantisation
    assert abs(k[0, 0] - (-20.591204)) < 1e-5


def test_writer_is_byte_identical_to_hcopy(native, tmp_path):
    src = os.path.join(P, "mfcc_e_d_a_K.mfc")
    k, per, kind = native.parm_read(src)
    out = str(tmp_path / "w.mfc")
    native.parm_write(out, k, per, kind, withCrc=True)
    assert open(out, "rb").read() == open(src, "rb").read()
    d, _, _ = native.parm_read(os.path.join(P, "mfcc_e_d_a_decompressed_by_ref.mfc"))
    native.parm_write(out, d, per, kind, withCrc=False)
    assert open(out, "rb").read() == open(os.path.join(P, "mfcc_e_d_a_decompressed_by_ref.mfc"), "rb").read()


def test_corruption_is_detected(native, tmp_path):
    raw = bytearray(open(os.path.join(P, "mfcc_e_d_a_K.mfc"), "rb").read())
    raw[500] ^= 0x40
    bad = str(tmp_path / "bad.mfc")
    open(bad, "wb").write(raw)
    with pytest.raises(native.HtkAmdError, match="CRC"):
        native.parm_read(bad)
    open(bad, "wb").write(raw[:200])
    with pytest.raises(native.HtkAmdError):
        native.parm_read(bad)
    with pytest.raises(native.HtkAmdError):
        native.parm_read(str(tmp_path / "missing.mfc"))
