"""Where (if anywhere) the device front end's floats differ from the oracle's (bit-equal to the reference's HCopy): mismatch rate per
configuration and column, over the configurations of tests/test_gpu_parity.py::test_mfcc_matches_reference_front_end and longer waveforms.
    python tools/mfcc_diag.py        (GPU box)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from htk_amd import capi
import pyoracle as po


def wave(n=48000, seed=7):
    rng = np.random.default_rng(seed); t = np.arange(n) / 16000
    return (3000 * np.sin(2 * np.pi * 440 * t) * np.sin(2 * np.pi * 3 * t) + rng.normal(0, 800, n)).clip(-32768, 32767).astype("<i2")


tot = bad = 0
for kind, kw in [("MFCC_0_D_A", {}), ("MFCC_E_D_A", {}), ("MFCC_E_D_A_Z", {}), ("MFCC_0", {}), ("MFCC_E_D", dict(rawEnergy=False, zMeanSource=True)),
                 ("MFCC_0_D_A", dict(loFreq=300.0, hiFreq=3400.0, numChans=20, numCeps=10, usePower=True)), ("MFCC_0_D_A_T", {}) if False else ("MFCC_E", dict(eNormalise=False))]:
    waves = [wave(48000, 7), wave(12345, 8), wave(400, 9), wave(399, 10)] + [wave(160000, s) for s in range(20, 26)]
    got, fo = capi.Mfcc(capi.mfcc_config(kind, **kw)).compute_host(waves)
    ref = np.concatenate([po.mfcc(w, po.mfcc_cfg(kind, **kw)) for w in waves])
    ne = got != ref
    tot += ne.size; bad += int(ne.sum())
    print(kind, kw, got.shape, "mismatches %d of %d" % (ne.sum(), ne.size), "columns with any:", np.nonzero(ne.any(0))[0].tolist())
    if ne.any():
        gi = got.view(np.int32).astype(np.int64); ri = ref.view(np.int32).astype(np.int64)
        print("   ulp differences (capped at 10):", np.bincount(np.minimum(np.abs(gi - ri)[ne], 10)).tolist(), "max abs diff", float(np.abs(got - ref).max()))
print("total: %d of %d floats differ" % (bad, tot))
