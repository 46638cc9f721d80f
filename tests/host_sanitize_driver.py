"""Exercises the host-side C code (htk_amd/host/*.c: MMF reader/writer/mix-up, labels/MLF, networks, parameter / waveform /
accumulator / statistics files, update) from a build with -fsanitize=address,undefined.  Run by tests/test_host_sanitizers.py in a
child process with the sanitizer runtime preloaded; prints OK at the end."""
import ctypes as C
import glob
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from htk_amd import capi  # noqa: E402

capi.LIBPATH = sys.argv[1]                     # the sanitizer build (host code only)
GOLD = os.path.join(ROOT, "tests", "golden")
demo = os.path.join(GOLD, "demo")


def main():
    tmp = tempfile.mkdtemp()
    # MMF: text + binary, macros, directory search, writers, mix-up
    for files, lst in (([os.path.join(GOLD, "mmf", "syn_in.mmf")], os.path.join(GOLD, "mmf", "syn_list")),
                       ([os.path.join(GOLD, "mmf", "syn_resaved_bin.mmf")], os.path.join(GOLD, "mmf", "syn_list")),
                       ([os.path.join(GOLD, "mmf", "syn_tied.mmf")], os.path.join(GOLD, "mmf", "syn_list")),
                       ([os.path.join(GOLD, "compv", "flat_hmm0.mmf")], os.path.join(demo, "bcplist"))):
        m = capi.Mmf(files=files, hmm_list=lst)
        q = m.packed()
        par = dict(mean=q["mean"], var=q["var"], gconst=q["gconst"], compWeight=q["compWeight"], transP=q["transP"])
        m.write(par, one_file=os.path.join(tmp, "o.mmf"))
        m.write(par, one_file=os.path.join(tmp, "o.bin"), binary=True)
        capi.Mmf(files=[os.path.join(tmp, "o.bin")], hmm_list=lst).packed()
    m = capi.Mmf(hmm_list=os.path.join(demo, "bcplist"), hmm_dir=os.path.join(demo, "hmm_final"))
    m.mixup(3); m.mixup(-2, states=[0, 4]); m.mixup(8)
    q = m.packed()
    m.write(dict(mean=q["mean"], var=q["var"], gconst=q["gconst"], compWeight=q["compWeight"], transP=q["transP"]), out_dir=tmp)
    for bad in ("<BEGINHMM> <NUMSTATES> 3", "~h \"x\" <BEGINHMM> <NUMSTATES> 3 <STATE> 2 <MEAN> 2 1 2", "~o <VECSIZE> 2 ~s \"a\" <NUMMIXES> 2 <MIXTURE> 1"):
        p = os.path.join(tmp, "bad.mmf"); open(p, "w").write(bad)
        try:
            capi.Mmf(files=[p])
        except capi.HtkAmdError:
            pass
    # labels, MLF
    for f in glob.glob(os.path.join(demo, "labels", "*.lab")):
        capi.labels_read(f)
    mlf = os.path.join(tmp, "x.mlf")
    open(mlf, "w").write('#!MLF!#\n"*/a.lab"\n0 100 x\n100 200 y 1.5\n///\n0 1 z\n.\n"*/b?.lab"\nw\n')
    ml = capi.Mlf(mlf)
    assert ml.find("dir/a.lab") is not None and ml.find("q/b1.lab") is not None and ml.find("nothing") is None
    t = capi.Trans(2)
    t.add(0, 300000, "a-b+c[2]", -10.0, aux1="a-b+c", aux1_score=-30.0, aux2="WORD", aux2_score=-2.5)
    t.add(300000, 700000, "a-b+c[3]", -20.0)
    t.add(700000, 900000, "'quoted", 0.0, aux1="x-y", aux1_score=0.0)
    t.format(100000.0, states=True, models=True, flags="NXC")
    t.write(os.path.join(tmp, "t.lab"))
    o = capi.MlfOut(os.path.join(tmp, "o.mlf")); o.add("*/t.rec", t); o.close()
    capi.Mlf(os.path.join(tmp, "o.mlf"))
    t2 = capi.Trans(2); t2.add(0, 100, "s[2]", 1.0, aux1="m", aux1_score=2.0, aux2="w", aux2_score=3.0); t2.format(100000.0, True, True, "SWTM")
    t2.write(os.path.join(tmp, "t2.lab"))
    scp = os.path.join(tmp, "x.scp")
    open(scp, "w").write('a.mfc "b c.mfc" u=p.mfc[3,9]\n')
    assert len(capi.scp_read(scp)) == 3
    # networks: SLF + dictionary, word-internal contexts, alignment networks
    for case in ("loop", "bigram", "tee", "wint"):
        d = os.path.join(GOLD, "decode", case)
        mm = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
        capi.Net(os.path.join(d, "net.slf"), os.path.join(d, "dict"), mm).arrays()
    d = os.path.join(GOLD, "decode", "bigram")
    mm = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    capi.Net(None, os.path.join(d, "dict"), mm, words=["AB", "E", "F"], boundary="H").arrays()
    try:
        capi.Net(None, os.path.join(d, "dict"), mm, words=["nope"])
    except capi.HtkAmdError:
        pass
    # parameter files (_C, _K), waveform files
    for f in glob.glob(os.path.join(GOLD, "parm", "*.mfc")) + glob.glob(os.path.join(GOLD, "quals", "*.mfc")):
        X, per, kind = capi.parm_read(f)
        capi.parm_write(os.path.join(tmp, "p.mfc"), X, per, kind, withCrc=True)
        Y, _, _ = capi.parm_read(os.path.join(tmp, "p.mfc"))
        assert np.array_equal(X, Y)
    capi.wave_read(os.path.join(GOLD, "wave", "test.wav"), capi.WAVE_WAV)
    capi.wave_read(os.path.join(GOLD, "wave", "test.htk"), capi.WAVE_HTK)
    open(os.path.join(tmp, "trunc.wav"), "wb").write(open(os.path.join(GOLD, "wave", "test.wav"), "rb").read()[:1000])
    for bad in ("trunc.wav",):
        try:
            capi.wave_read(os.path.join(tmp, bad), capi.WAVE_WAV)
        except capi.HtkAmdError:
            pass
    # accumulator / statistics files on a synthetic vector
    m = capi.Mmf(files=[os.path.join(GOLD, "mmf", "syn_in.mmf")], hmm_list=os.path.join(GOLD, "mmf", "syn_list"))
    q = m.packed()
    lay = capi.AccsLayout()
    d0, keep = capi._desc_from_packed(q)
    capi.check(capi.lib().htkamd_accs_layout_from_desc(C.byref(d0), C.byref(lay)), "layout")
    vec = np.random.default_rng(1).random(lay.total)
    vec[lay.nEgs:lay.nEgs + q["numPhys"]] = 3
    capi.accs_dump_file(q, vec, m.phys_names, os.path.join(tmp, "HER1.acc"))
    back = np.zeros_like(vec)
    capi.accs_load_file(q, back, m.phys_names, os.path.join(tmp, "HER1.acc"))
    capi.stats_write_file(q, vec, m.phys_names, os.path.join(tmp, "stats"))
    capi.write_vfloors(os.path.join(tmp, "vFloors"), q["var"][0], 0.01)
    print("OK")


if __name__ == "__main__":
    main()
