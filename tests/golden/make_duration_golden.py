"""Fixture for duration vectors in model files (GetDuration HModel.c:1580, PutDuration :2840): a small hand-written set with a <GAMMAD> duration kind,
a ~d macro referenced by a state and by a model, an inline <DURATION> in a state and in a model -- and the same set as the reference re-saves it
(HHEd with an empty edit script: LoadHMMSet + SaveHMMSet), in text and (-B) in binary form.  Needs oracle/_ref.
    python tests/golden/make_duration_golden.py"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "mmf")

SET = """~o
<STREAMINFO> 1 3
<VECSIZE> 3<GAMMAD><USER><DIAGC>
~d "durA"
<DURATION> 2
 4.500000e+00 1.250000e+00
~t "T1"
<TRANSP> 4
 0 1 0 0
 0 0.6 0.4 0
 0 0 0.7 0.3
 0 0 0 0
~s "S1"
<MEAN> 3
 1.0 2.0 3.0
<VARIANCE> 3
 1.0 1.5 2.0
~d "durA"
~h "a"
<BEGINHMM>
<NUMSTATES> 4
<STATE> 2
~s "S1"
<STATE> 3
<NUMMIXES> 2
<MIXTURE> 1 0.25
<MEAN> 3
 0.5 0.25 0.125
<VARIANCE> 3
 2.0 2.0 2.0
<MIXTURE> 2 0.75
<MEAN> 3
 -1.0 -2.0 -3.0
<VARIANCE> 3
 0.5 0.5 0.5
<DURATION> 3
 1.0 2.0 3.0
~t "T1"
~d "durA"
<ENDHMM>
~h "b"
<BEGINHMM>
<NUMSTATES> 4
<STATE> 2
~s "S1"
<STATE> 3
<MEAN> 3
 9.0 8.0 7.0
<VARIANCE> 3
 1.0 1.0 1.0
~t "T1"
<DURATION> 1
 7.0
<ENDHMM>
"""


def main():
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "dur_in.mmf"), "w").write(SET)
    open(os.path.join(OUT, "dur_list"), "w").write("a\nb\n")
    open(os.path.join(OUT, "empty.hed"), "w").close()
    hhed = os.path.join(ROOT, "oracle", "_ref", "HHEd")
    subprocess.check_call([hhed, "-H", "dur_in.mmf", "-w", "dur_resaved.mmf", "empty.hed", "dur_list"], cwd=OUT)
    subprocess.check_call([hhed, "-B", "-H", "dur_in.mmf", "-w", "dur_resaved_bin.mmf", "empty.hed", "dur_list"], cwd=OUT)
    os.remove(os.path.join(OUT, "empty.hed"))
    print(open(os.path.join(OUT, "dur_resaved.mmf")).read())


if __name__ == "__main__":
    main()
