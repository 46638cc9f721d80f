"""Throughput of the device front end at BASELINE config[4]: 16 kHz mono PCM, 25 ms / 10 ms, 26 channels, 12 cepstra,
MFCC_0_D_A, 3-second utterances (298 frames).  Waveforms resident on the device; time = htkamd_mfcc_compute only.
(The CPU side of the comparison -- the reference's HCopy on one core -- is measured separately; tools never touch oracle/.)
Run on the GPU box: python tools/mfcc_bench.py [nUtt]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from htk_amd import capi  # noqa: E402


def main():
    nU = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(7)
    n = 48000
    t = np.arange(n) / 16000.0
    base = (3000 * np.sin(2 * np.pi * 440 * t) + 2000 * np.sin(2 * np.pi * 1800 * t)).astype(np.float32)
    waves = [(base + rng.normal(0, 500, n)).astype(np.int16) for _ in range(8)]
    waves = [waves[i % 8] for i in range(nU)]
    cfg = capi.mfcc_config("MFCC_0_D_A")
    fe = capi.Mfcc(cfg)
    sampOff = np.concatenate([[0], np.cumsum([len(w) for w in waves])]).astype(np.int32)
    allw = np.concatenate(waves)
    frames = capi.lib().htkamd_mfcc_num_frames(C.byref(cfg), C.c_int(n)) * nU
    dW = capi.DevArray(allw)
    dO = capi.DevArray(nbytes=4 * frames * fe.cols)
    frameOff = np.zeros(nU + 1, np.int32)
    for rep in range(3):
        t0 = time.perf_counter()
        capi.check(capi.lib().htkamd_mfcc_compute(fe.h, dW.ptr, allw.ctypes.data_as(C.c_void_p) and sampOff.ctypes.data_as(C.c_void_p), C.c_int(nU),
                                                  frameOff.ctypes.data_as(C.c_void_p), dO.ptr, None), "mfcc_compute")
        dt = time.perf_counter() - t0
        print("GPU run %d: %d utterances, %d frames in %.2f ms = %.2f M frames/s, %.2f GB/s of PCM, %.0f x real time"
              % (rep, nU, frames, dt * 1e3, frames / dt / 1e6, allw.nbytes / dt / 1e9, nU * 3.0 / dt))


if __name__ == "__main__":
    main()
