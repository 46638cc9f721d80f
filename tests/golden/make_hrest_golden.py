"""Fixtures for the isolated-unit trainers of BASELINE config[0] (HTKDemo monPlainM1S1): the models HInit wrote (hmms/hmm.0, the
INPUT of HRest), the prototype HInit started from, and the trace lines of both tools ("Ave LogProb at iter ...").  The models
HRest wrote (hmms/hmm.1) are tests/golden/demo/hmm1 already; this script checks that its run reproduces them byte for byte.
    python tests/golden/make_hrest_golden.py      (needs /root/reference/HTKDemo and oracle/_ref)"""
import filecmp
import os
import re
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
    if not os.path.isdir(src) or not os.path.exists(os.path.join(REF, "HRest")):
        sys.exit("needs /root/reference/HTKDemo and oracle/_ref (make -C oracle)")
    tmp = tempfile.mkdtemp(prefix="htkdemo_")
    demo = os.path.join(tmp, "HTKDemo")
    shutil.copytree(src, demo)
    subprocess.check_call(["chmod", "-R", "u+w", demo])
    for d in ("hmms/hmm.0", "hmms/hmm.1", "hmms/hmm.2", "hmms/hmm.3", "hmms/tmp", "proto", "test", "accs"):
        os.makedirs(os.path.join(demo, d), exist_ok=True)
    env = dict(os.environ, PATH=REF + os.pathsep + os.environ["PATH"], PWD=demo)
    log = subprocess.run(["perl", "runDemo", "configs/monPlainM1S1.dcf"], cwd=demo, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True).stdout
    for m in "SCVNL":
        if not filecmp.cmp(os.path.join(demo, "hmms/hmm.1", m), os.path.join(OUT, "hmm1", m), shallow=False):
            sys.exit("this run's hmm.1/%s differs from the committed fixture" % m)
    os.makedirs(os.path.join(OUT, "hmm0"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "proto"), exist_ok=True)
    for m in "SCVNL":
        shutil.copy(os.path.join(demo, "hmms/hmm.0", m), os.path.join(OUT, "hmm0", m))
        shutil.copy(os.path.join(demo, "proto", m), os.path.join(OUT, "proto", m))
    keep, tool, model = [], None, None
    for l in log.splitlines():
        m = re.match(r"Calling (HInit|HRest) for HMM (\S+)", l)
        if m:
            tool, model = m.group(1), m.group(2)
            continue
        if tool and re.search(r"Ave LogProb|Estimation (converged|aborted)|Iteration \d+: Average LogP|examples", l):
            keep.append("%s %s: %s" % (tool, model, l.strip()))
        if l.startswith("Calling HERest") or "HERest" in l:
            tool = None
    open(os.path.join(OUT, "hinit_hrest.log"), "w").write("\n".join(keep) + "\n")
    print("\n".join(keep[:12]))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
