"""Fixtures for BASELINE config[0] (HTKDemo monophone system, the reference's own CPU-runnable case).

Run in the build container (needs /root/reference and the reference tools built by `make -C oracle`):
    python tests/golden/make_demo_golden.py

It copies HTKDemo to a scratch directory, runs the demo's own driver (`runDemo configs/monPlainM1S1.dcf`: HInit + HRest
+ 3 x HERest + HVite/HResults) with the reference tools, and keeps
  * inputs  : the 5 single-Gaussian monophone models after HRest (hmms/hmm.1; plus the same set re-saved by the reference's
              HHEd with an empty script, the writer's known answer), the HMM list, the 7 training parameter
              files (MFCC_E, 13 columns; DATA of the reference) and their label files,
  * expected: the models the reference's FIRST embedded re-estimation writes from them (HERest -w 3 -v 0.05 -u tmvw
              -t 2000.0, TARGETKIND = MFCC_E_D) and the lines of its log the survey quotes
              ("average log prob per frame = -5.900196e+01", "27 floored variance elements in 15 different mixes").
Everything lands in tests/golden/demo/ (about 130 kB)."""
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
    if not os.path.isdir(src) or not os.path.exists(os.path.join(REF, "HERest")):
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
    if "-5.900196e+01" not in log:
        sys.exit("demo run did not reproduce the known first-pass likelihood:\n" + log[-3000:])
    # first embedded pass on its own, from hmm.1
    os.makedirs(os.path.join(demo, "hmms/pass1"), exist_ok=True)
    train = sorted(os.path.join("data/train", f) for f in os.listdir(os.path.join(demo, "data/train")) if f.endswith(".mfc"))
    cmd = ["HERest", "-A", "-w", "3", "-v", "0.05", "-C", "toolconfs/herest.conf", "-u", "tmvw", "-d", "hmms/hmm.1", "-D",
           "-M", "hmms/pass1", "-L", "labels/bcplabs/mon", "-t", "2000.0", "-T", "1", "lists/bcplist"] + train
    log1 = subprocess.run(cmd, cwd=demo, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    # hmm.1 as the reference's own LoadHMMSet + SaveHMMSet round trip writes it (transition rows pass through log/exp)
    os.makedirs(os.path.join(demo, "hmms/resaved"), exist_ok=True)
    open(os.path.join(demo, "empty.hed"), "w").close()
    subprocess.check_call(["HHEd", "-C", "toolconfs/herest.conf", "-d", "hmms/hmm.1", "-M", "hmms/resaved", "empty.hed", "lists/bcplist"], cwd=demo, env=env)
    shutil.rmtree(OUT, ignore_errors=True)
    for d in ("hmm1", "hmm1_resaved", "hmm2_expected", "train", "labels"):
        os.makedirs(os.path.join(OUT, d))
    for m in "SCVNL":
        shutil.copy(os.path.join(demo, "hmms/hmm.1", m), os.path.join(OUT, "hmm1", m))
        shutil.copy(os.path.join(demo, "hmms/pass1", m), os.path.join(OUT, "hmm2_expected", m))
        shutil.copy(os.path.join(demo, "hmms/resaved", m), os.path.join(OUT, "hmm1_resaved", m))
    shutil.copy(os.path.join(demo, "lists/bcplist"), os.path.join(OUT, "bcplist"))
    for f in train:
        shutil.copy(os.path.join(demo, f), os.path.join(OUT, "train", os.path.basename(f)))
        lab = os.path.basename(f).replace(".mfc", ".lab")
        shutil.copy(os.path.join(demo, "labels/bcplabs/mon", lab), os.path.join(OUT, "labels", lab))
    keep = [l for l in log1.splitlines() if re.search(r"average log prob|floored variance|Pruning|Updating|frames", l)]
    with open(os.path.join(OUT, "herest_pass1.log"), "w") as f:
        f.write("\n".join(keep) + "\n")
    print("\n".join(keep))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
