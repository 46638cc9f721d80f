"""Fixtures for the RECOGNITION side of BASELINE config[0] (HTKDemo monPlainM1S1): the final models of the demo's training chain
(hmms/hmm.2 after its embedded passes), the phone-loop lattice and vocabulary it recognises with, the three test files, and the
label files the reference's HVite writes for the test and training sets with the demo's switches (-t 300.0 -p 5.0 -s 0.0).
HResults on those label files gives the figures of HTKDemo/results/monPlainM1S1.res (WORD: %Corr=63.91, Acc=59.40 on the test set),
which the script checks before it keeps anything.
    python tests/golden/make_demo_test_golden.py      (needs /root/reference/HTKDemo and oracle/_ref)"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "demo")


def main():
    src = "/root/reference/HTKDemo"
    tmp = tempfile.mkdtemp(prefix="htkdemo_")
    demo = os.path.join(tmp, "HTKDemo")
    shutil.copytree(src, demo)
    subprocess.check_call(["chmod", "-R", "u+w", demo])
    for d in ("hmms/hmm.0", "hmms/hmm.1", "hmms/hmm.2", "hmms/hmm.3", "hmms/tmp", "proto", "test", "accs"):
        os.makedirs(os.path.join(demo, d), exist_ok=True)
    env = dict(os.environ, PATH=REF + os.pathsep + os.environ["PATH"], PWD=demo)
    log = subprocess.run(["perl", "runDemo", "configs/monPlainM1S1.dcf"], cwd=demo, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True).stdout
    if "%Corr=63.91, Acc=59.40 [H=85, D=35, S=13, I=6, N=133]" not in log:
        sys.exit("demo run did not reproduce the known test-set result:\n" + log[-2000:])
    for d in ("hmm_final", "test"):
        os.makedirs(os.path.join(OUT, d), exist_ok=True)
    for m in "SCVNL":
        shutil.copy(os.path.join(demo, "hmms/hmm.2", m), os.path.join(OUT, "hmm_final", m))
    shutil.copy(os.path.join(demo, "networks/monLattice"), os.path.join(OUT, "monLattice"))
    shutil.copy(os.path.join(demo, "lists/bcpvocab"), os.path.join(OUT, "bcpvocab"))
    expected = {}
    for name, files in (("test", sorted(os.listdir(os.path.join(demo, "data/test")))), ("train", sorted(os.listdir(os.path.join(demo, "data/train"))))):
        rec = os.path.join(demo, "rec_" + name)
        os.makedirs(rec)
        paths = [os.path.join("data", name, f) for f in files if f.endswith(".mfc")]
        subprocess.check_call(["HVite", "-C", "toolconfs/hvite.conf", "-d", "hmms/hmm.2", "-l", rec, "-w", "networks/monLattice", "-t", "300.0",
                               "-p", "5.0", "-s", "0.0", "lists/bcpvocab", "lists/bcplist"] + paths, cwd=demo, env=env,
                              stdout=subprocess.DEVNULL)
        expected[name] = {f.replace(".mfc", ""): open(os.path.join(rec, f.replace(".mfc", ".rec"))).read().splitlines() for f in files if f.endswith(".mfc")}
        if name == "test":
            for f in files:
                shutil.copy(os.path.join(demo, "data/test", f), os.path.join(OUT, "test", f))
    json.dump(expected, open(os.path.join(OUT, "hvite_expected.json"), "w"), indent=1)
    print({k: {u: len(v) for u, v in per.items()} for k, per in expected.items()})
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
