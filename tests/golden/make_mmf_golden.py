"""Fixture for the MMF reader/writer: a small synthetic set with ~s/~t macros, written by htk_amd.synth, and the same set
as the reference re-saves it (HHEd with an empty edit script: LoadHMMSet + SaveHMMSet), in text and (-B) in binary form.  Needs oracle/_ref.
    python tests/golden/make_mmf_golden.py"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from htk_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "mmf")


def main():
    os.makedirs(OUT, exist_ok=True)
    s = synth.generate(10, 3, 6, 1, 20, 77, D=5)
    synth.write_mmf(os.path.join(OUT, "syn_in.mmf"), s, kind="USER")
    with open(os.path.join(OUT, "syn_list"), "w") as f:
        f.write("".join("p%d\n" % i for i in range(6)))
        f.write("alias p3\n")                       # a logical name sharing a physical model
    open(os.path.join(OUT, "empty.hed"), "w").close()
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "HHEd"), "-H", "syn_in.mmf", "-w", "syn_resaved.mmf", "empty.hed", "syn_list"], cwd=OUT)
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "HHEd"), "-B", "-H", "syn_in.mmf", "-w", "syn_resaved_bin.mmf", "empty.hed", "syn_list"], cwd=OUT)
    # shared mixture pdfs: the reference's HHEd ties some components into ~m macros
    with open(os.path.join(OUT, "tie.hed"), "w") as f:
        f.write("TI mA {(p0,p1,p2).state[2].mix[1]}\nTI mB {(p3,p4).state[3-4].mix[2]}\n")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "HHEd"), "-H", "syn_in.mmf", "-w", "syn_tied.mmf", "tie.hed", "syn_list"], cwd=OUT)
    os.remove(os.path.join(OUT, "tie.hed"))
    os.remove(os.path.join(OUT, "empty.hed"))
    print(os.listdir(OUT))


if __name__ == "__main__":
    main()
