#!/usr/bin/env python3
"""Golden vectors for the waveform end of the front end (BASELINE config[4]): a 1.0 s 16 kHz synthetic waveform as a RIFF/WAVE
file and as an HTK WAVEFORM file, and the MFCC_0_D_A file the reference's HCopy (oracle/_ref, built from /root/reference by
oracle/Makefile) codes from each with SURVEY App. F's configuration (the two outputs are identical).

    python tests/golden/make_wave_golden.py
"""
import os
import struct
import subprocess
import tempfile
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "wave")
CFG = ("SOURCERATE = 625\nWINDOWSIZE = 250000.0\nTARGETRATE = 100000.0\nNUMCHANS = 26\nNUMCEPS = 12\nCEPLIFTER = 22\n"
       "PREEMCOEF = 0.97\nUSEHAMMING = T\nTARGETKIND = MFCC_0_D_A\nENORMALISE = F\n")


def test_wave(n=16000, seed=7):
    rng = np.random.default_rng(seed); t = np.arange(n) / 16000
    return (3000 * np.sin(2 * np.pi * 440 * t) * np.sin(2 * np.pi * 3 * t) + rng.normal(0, 800, n)).clip(-32768, 32767).astype("<i2")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    x = test_wave()
    with wave.open(os.path.join(OUT, "test.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
    with open(os.path.join(OUT, "test.htk"), "wb") as f:
        f.write(struct.pack(">iihh", len(x), 625, 2, 0)); f.write(x.astype(">i2").tobytes())
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for fmt, src in (("WAV", "test.wav"), ("HTK", "test.htk")):
            cfg = os.path.join(d, "cfg" + fmt)
            open(cfg, "w").write("SOURCEFORMAT = %s\n" % fmt + CFG)
            out = os.path.join(d, fmt + ".mfc")
            subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "HCopy"), "-C", cfg, os.path.join(OUT, src), out])
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1]
        open(os.path.join(OUT, "test_MFCC_0_D_A.mfc"), "wb").write(outs[0])
    print(sorted(os.listdir(OUT)), len(outs[0]))
