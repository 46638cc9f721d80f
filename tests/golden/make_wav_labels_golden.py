"""Label files the reference's HVite writes from a WAVEFORM source (its own HParm / HSigP front end inside the tool): forced alignment
(-a -m) and recognition over a word loop (-w), on tests/golden/wave/test.wav with a 5-phone set fitted to that file (tests/golden/wave/fitted.mmf, written here).  The expected lines go to tests/golden/wave/expected_wav_labels.json;
tests/test_cli_tools.py::test_wav_sources_align_and_decode_match_reference_hvite asks the device path for the same bytes.
    python tests/golden/make_wav_labels_golden.py        (needs oracle/_ref)"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from htk_amd import synth  # noqa: E402

FRONT = "SOURCERATE = 625\nWINDOWSIZE = 250000.0\nTARGETRATE = 100000.0\nNUMCHANS = 26\nNUMCEPS = 12\nCEPLIFTER = 22\nPREEMCOEF = 0.97\nUSEHAMMING = T\nENORMALISE = F\n"
LABELS = ["p0", "p1", "p2", "p3", "p4"]


def fitted_set():
    """a 5-phone set FITTED to test.wav: the file's 98 MFCC_0_D_A frames (the oracle's front end = the reference's) cut into 5 x 3 runs,
    a state per run -- two components at the run's mean -/+ 0.3 sigma -- so that alignments and recognition are decided by the data"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    raw = open(os.path.join(HERE, "wave", "test.wav"), "rb").read()
    pcm = np.frombuffer(raw[44:], "<i2")
    X = po.mfcc(pcm, po.mfcc_cfg("MFCC_0_D_A", eNormalise=False))
    T, D = X.shape
    cuts = np.linspace(0, T, 16).astype(int)
    gvar = X.var(0) + 1e-3
    means = np.zeros((15, 2, D), np.float32); var = np.zeros((15, 2, D), np.float32)
    for k in range(15):
        seg = X[cuts[k]:cuts[k + 1]]
        mu = seg.mean(0); v = np.maximum(seg.var(0), 0.1 * gvar)
        means[k, 0] = mu - 0.3 * np.sqrt(v); means[k, 1] = mu + 0.3 * np.sqrt(v); var[k, :] = v
    w = np.tile(np.array([0.6, 0.4], np.float32), (15, 1))
    st = np.arange(15, dtype=np.int32).reshape(5, 3)
    return synth.SynthSet(D=D, NS=15, M=2, NP=5, means=means, var=var, w=w, st=st)


def write_case(d, fmt="WAV", mmf=None):
    """the files both tools read: model set (the committed tests/golden/wave/fitted.mmf unless `mmf` names another), lists, configuration,
    word loop; returns the model names"""
    import shutil
    names = ["p%d" % i for i in range(5)]
    shutil.copy(mmf or os.path.join(HERE, "wave", "fitted.mmf"), os.path.join(d, "MMF"))
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    open(os.path.join(d, "dict"), "w").write("".join("%s %s\n" % (n, n) for n in names))
    open(os.path.join(d, "wav.conf"), "w").write("SOURCEFORMAT = %s\n%sTARGETKIND = MFCC_0_D_A\n" % (fmt, FRONT) + ("SOURCEKIND = WAVEFORM\n" if fmt == "HTK" else ""))
    V = len(names)
    with open(os.path.join(d, "loop.slf"), "w") as f:             # the word loop HBuild writes
        f.write("VERSION=1.0\nN=%d L=%d\nI=0 W=!NULL\nI=1 W=!NULL\n" % (V + 4, 2 * V + 3))
        for i, n in enumerate(names):
            f.write("I=%d W=%s\n" % (2 + i, n))
        f.write("I=%d W=!NULL\nI=%d W=!NULL\n" % (V + 2, V + 3))
        j = 0
        f.write("J=%d S=0 E=1 l=0.00\n" % j); j += 1
        f.write("J=%d S=%d E=1 l=0.00\n" % (j, V + 2)); j += 1
        for i in range(V):
            f.write("J=%d S=1 E=%d l=-1.61\n" % (j, 2 + i)); j += 1
            f.write("J=%d S=%d E=%d l=0.00\n" % (j, 2 + i, V + 2)); j += 1
        f.write("J=%d S=%d E=%d l=0.00\n" % (j, V + 2, V + 3))
    return names


def run_tool(exe, d, wav, mode):
    out = os.path.join(d, "out_" + mode + "_" + os.path.basename(exe))
    os.makedirs(out, exist_ok=True)
    base = os.path.splitext(os.path.basename(wav))[0]
    common = ["-C", os.path.join(d, "wav.conf"), "-H", os.path.join(d, "MMF"), "-l", out, "-y", "rec"]
    if mode == "align":
        open(os.path.join(out, base + ".lab"), "w").write("\n".join(LABELS) + "\n")
        cmd = [exe] + common + ["-a", "-m", "-f", "-L", out, os.path.join(d, "dict"), os.path.join(d, "hmmlist"), wav]
    else:
        cmd = [exe] + common + ["-w", os.path.join(d, "loop.slf"), "-t", "250.0", "-p", "0.0", os.path.join(d, "dict"), os.path.join(d, "hmmlist"), wav]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return open(os.path.join(out, base + ".rec")).read().splitlines()


def main():
    ref = os.path.join(ROOT, "oracle", "_ref", "HVite")
    fs = fitted_set()
    synth.write_mmf_packed(os.path.join(HERE, "wave", "fitted.mmf"), fs.packed(), ["p%d" % i for i in range(5)], kind="MFCC_0_D_A")
    exp = {"generator": "tests/golden/make_wav_labels_golden.py: oracle/_ref/HVite on tests/golden/wave/test.wav / test.htk with tests/golden/wave/fitted.mmf"}
    for fmt, src in (("WAV", "test.wav"), ("HTK", "test.htk")):
        with tempfile.TemporaryDirectory() as d:
            write_case(d, fmt)
            for mode in ("align", "loop"):
                exp["%s/%s" % (fmt, mode)] = run_tool(ref, d, os.path.join(HERE, "wave", src), mode)
    json.dump(exp, open(os.path.join(HERE, "wave", "expected_wav_labels.json"), "w"), indent=1)
    print(json.dumps(exp, indent=1)[:2500])


if __name__ == "__main__":
    main()
