"""Throughput of the batch forced-alignment kernels (HVite -a) over chain length.   python tools/align_bench.py [utterances]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from htk_amd import capi, synth  # noqa: E402


def main():
    nu = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for frames in (500, 756, 804, 1560):
        s = synth.generate_fast(1000, 8, 2000, nu, frames, seed=5, model_seed=3)
        model = capi.Model(s.packed())
        X = np.concatenate(s.feats)
        frameOff = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int32)
        labOff = np.concatenate([[0], np.cumsum([len(q) for q in s.seqs])]).astype(np.int32)
        labs = np.concatenate(s.seqs).astype(np.int32)
        dX = capi.DevArray(X)
        vit = capi.Viterbi(model)
        vit.align(dX.ptr.value, frameOff, labOff, labs)                 # warm-up
        t0 = time.perf_counter()
        for _ in range(3):
            got = vit.align(dX.ptr.value, frameOff, labOff, labs)
        dt = (time.perf_counter() - t0) / 3
        assert all(g["status"] == 1 for g in got)
        print("frames %5d  models/utt %4d  %7.1f ms per %d utterances  %9.0f utt/s  %6.2f M frames/s"
              % (frames, len(s.seqs[0]), dt * 1e3, nu, nu / dt, nu * frames / dt / 1e6))


if __name__ == "__main__":
    main()
