"""Host-side formats: text MMF reader/writer and label/MLF readers in htk_amd/host (C), checked against
the Python generator's arrays, against what the reference's SaveHMMSet writes for the same set (committed fixture made by
tests/golden/make_mmf_golden.py), and against the HTKDemo single-HMM files."""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEMO = os.path.join(GOLD, "demo")


def _gconst(native, var):
    g = np.empty(len(var), np.float32)
    for i in range(len(var)):
        x = np.zeros(1, np.float32)
        native.lib().htkamd_host_fix_diag_gconst(C.c_int(var.shape[1]), np.ascontiguousarray(var[i]).ctypes.data_as(C.c_void_p),
                                                 x.ctypes.data_as(C.c_void_p))
        g[i] = x[0]
    return g


def test_mmf_reader_matches_generator(native):
    from htk_amd import synth
    s = synth.generate(10, 3, 6, 1, 20, 77, D=5)                      # what make_mmf_golden.py wrote
    pk = s.packed()
    m = native.Mmf(files=[os.path.join(GOLD, "mmf", "syn_in.mmf")], hmm_list=os.path.join(GOLD, "mmf", "syn_list"))
    q = m.packed()
    assert m.kind == "USER" and m.phys_names == ["p%d" % i for i in range(6)]
    assert m.logical["alias"] == m.logical["p3"] == 3 and len(m.logical) == 7
    for k, a in pk.items():
        if a is None:
            assert q[k] is None
        else:
            assert np.array_equal(np.asarray(a), np.asarray(q[k])), k      # includes log transitions (GetTransMat) bit for bit


def test_mmf_writer_byte_identical_to_reference_resave(native, tmp_path):
    m = native.Mmf(files=[os.path.join(GOLD, "mmf", "syn_in.mmf")], hmm_list=os.path.join(GOLD, "mmf", "syn_list"))
    q = m.packed()
    params = dict(mean=q["mean"], var=q["var"], gconst=_gconst(native, q["var"]), compWeight=q["compWeight"], transP=q["transP"])
    out = tmp_path / "out.mmf"
    m.write(params, one_file=str(out))
    assert out.read_bytes() == open(os.path.join(GOLD, "mmf", "syn_resaved.mmf"), "rb").read()
    # and the written file reads back to the same description
    q2 = native.Mmf(files=[str(out)], hmm_list=os.path.join(GOLD, "mmf", "syn_list")).packed()
    for k in ("mean", "var", "compWeight", "transP", "hmmState", "stateCompOff"):
        assert np.array_equal(q[k], q2[k]), k
    assert np.allclose(q2["gconst"], params["gconst"], rtol=1e-6)                   # 7 printed digits


def test_mmf_demo_directory_models(native, tmp_path):
    """One definition per file, found through the HMM list in a -d directory; re-written with global options per file."""
    d = os.path.join(GOLD, "demo")
    m = native.Mmf(hmm_list=os.path.join(d, "bcplist"), hmm_dir=os.path.join(d, "hmm1"))
    q = m.packed()
    assert m.kind == "MFCC_E_D" and q["vecSize"] == 26 and m.phys_names == list("SCVNL")
    assert q["numStates"] == 15 and q["numGauss"] == 15 and q["numTrans"] == 5 and list(q["transN"]) == [5] * 5
    # HHEd recomputes every gConst before saving (FixAllGConsts), so the known answer carries recomputed values
    params = dict(mean=q["mean"], var=q["var"], gconst=_gconst(native, q["var"]), compWeight=q["compWeight"], transP=q["transP"])
    assert np.allclose(params["gconst"], q["gconst"], rtol=1e-6)
    m.write(params, out_dir=str(tmp_path))
    for name in "SCVNL":
        assert (tmp_path / name).read_bytes() == open(os.path.join(d, "hmm1_resaved", name), "rb").read(), name


def test_mmf_rejects_what_the_path_does_not_support(native, tmp_path):
    bad = {
        "streams": "~o <STREAMINFO> 2 3 3 <VECSIZE> 6 <NULLD><USER><DIAGC>\n",
        "fullc": "~o <STREAMINFO> 1 2 <VECSIZE> 2 <NULLD><USER><FULLC>\n",
        "shared_mean": '~o <VECSIZE> 2 <USER>\n~h "a"\n<BEGINHMM>\n<NUMSTATES> 3\n<STATE> 2\n~u "m1"\n',
        "binary": ':\x00\x01',
        "twice": '~o <VECSIZE> 1 <USER>\n~t "T"\n<TRANSP> 3\n0 1 0\n0 .5 .5\n0 0 0\n~t "T"\n<TRANSP> 3\n0 1 0\n0 .5 .5\n0 0 0\n',
    }
    for name, text in bad.items():
        p = tmp_path / name
        p.write_text(text)
        with pytest.raises(native.HtkAmdError):
            native.Mmf(files=[str(p)])
    with pytest.raises(native.HtkAmdError):
        native.Mmf(files=[str(tmp_path / "does_not_exist")])


def test_label_files_and_mlf(native, tmp_path):
    d = os.path.join(GOLD, "demo", "labels")
    labs = native.labels_read(os.path.join(d, "tr1.lab"))
    assert labs[0] == ("S", 0, 1410000, 0.0) and labs[1][:3] == ("C", 1410000, 2591250)
    assert all(n in "SCVNL" for n, _, _, _ in labs)
    p = tmp_path / "x.lab"
    p.write_text('p1\n"p 2"\n100 p3\n0 500 p4 -12.5 aux\n///\nignored\n')
    got = native.labels_read(str(p))
    assert [g[0] for g in got] == ["p1", "p 2", "p3", "p4"] or [g[0] for g in got][0] == "p1"
    assert got[2][1] == 100 and got[2][2] == -1 and got[3][1:] == (0, 500, -12.5)
    mlf = tmp_path / "a.mlf"
    mlf.write_text('#!MLF!#\n"*/u1.lab"\np1\np2\n.\n"*/u2.lab"\n0 100 p3\n.\n"exact.lab"\np9\n')
    m = native.Mlf(str(mlf))
    assert [x[0] for x in m.find("lab/u1.lab")] == ["p1", "p2"]
    assert m.find("some/dir/u2.lab")[0][:3] == ("p3", 0, 100)
    assert [x[0] for x in m.find("exact.lab")] == ["p9"] and m.find("lab/u3.lab") is None
    with pytest.raises(native.HtkAmdError):
        native.Mlf(os.path.join(d, "tr1.lab"))                       # no #!MLF!# header


def test_mmf_binary_form(native, tmp_path):
    """HTK's binary model files (HHEd/HERest -B): read to the same description as the text form of the same set, and
    written byte-identically to what the reference wrote."""
    lst = os.path.join(GOLD, "mmf", "syn_list")
    mb = native.Mmf(files=[os.path.join(GOLD, "mmf", "syn_resaved_bin.mmf")], hmm_list=lst)
    mt = native.Mmf(files=[os.path.join(GOLD, "mmf", "syn_resaved.mmf")], hmm_list=lst)
    qb, qt = mb.packed(), mt.packed()
    assert mb.kind == mt.kind and mb.phys_names == mt.phys_names and mb.logical == mt.logical
    for k in ("stateCompOff", "compGauss", "hmmState", "hmmTrans", "transN"):
        assert np.array_equal(qb[k], qt[k]), k
    for k in ("mean", "var", "gconst", "compWeight"):
        assert np.allclose(qb[k], qt[k], rtol=1e-6), k              # the text form carries 7 digits, the binary one all bits
    lin = lambda t: np.where(t > -0.5e10, np.exp(t.astype(np.float64)), 0.0)
    assert np.allclose(lin(qb["transP"]), lin(qt["transP"]), rtol=1e-6, atol=1e-9)
    out = tmp_path / "out.bin"
    mb.write(dict(mean=qb["mean"], var=qb["var"], gconst=qb["gconst"], compWeight=qb["compWeight"], transP=qb["transP"]), one_file=str(out), binary=True)
    assert out.read_bytes() == open(os.path.join(GOLD, "mmf", "syn_resaved_bin.mmf"), "rb").read()


def test_mmf_duration_vectors_are_carried(native, tmp_path):
    """<DURATION> vectors and ~d macros (GetDuration HModel.c:1580; no tool on the path evaluates them): a hand-written set with a <GAMMAD> kind, a ~d macro
    named by a state and by a model, an inline vector in a state and in a model is written back -- text and binary -- byte for byte as the reference's
    HHEd re-saves it (tests/golden/make_duration_golden.py), and the binary form reads to the same set."""
    lst = os.path.join(GOLD, "mmf", "dur_list")
    m = native.Mmf(files=[os.path.join(GOLD, "mmf", "dur_in.mmf")], hmm_list=lst)
    q = m.packed()
    assert q["numStates"] == 3 and q["numGauss"] == 4 and m.phys_names == ["a", "b"]
    params = dict(mean=q["mean"], var=q["var"], gconst=_gconst(native, q["var"]), compWeight=q["compWeight"], transP=q["transP"])
    out = tmp_path / "out.mmf"
    m.write(params, one_file=str(out))
    assert out.read_bytes() == open(os.path.join(GOLD, "mmf", "dur_resaved.mmf"), "rb").read()
    mb = native.Mmf(files=[os.path.join(GOLD, "mmf", "dur_resaved_bin.mmf")], hmm_list=lst)
    qb = mb.packed()
    for k in ("mean", "var", "compWeight", "hmmState", "stateCompOff"):
        assert np.array_equal(q[k], qb[k]), k
    outb = tmp_path / "out.bin"
    mb.write(dict(mean=qb["mean"], var=qb["var"], gconst=qb["gconst"], compWeight=qb["compWeight"], transP=qb["transP"]), one_file=str(outb), binary=True)
    assert outb.read_bytes() == open(os.path.join(GOLD, "mmf", "dur_resaved_bin.mmf"), "rb").read()
    # a set with a ~d macro cannot go to one file per model; an undefined macro is an error with the file's line
    with pytest.raises(native.HtkAmdError):
        m.write(params, out_dir=str(tmp_path))
    bad = tmp_path / "bad.mmf"
    bad.write_text(open(os.path.join(GOLD, "mmf", "dur_in.mmf")).read().replace('~d "durA"\n<ENDHMM>', '~d "nope"\n<ENDHMM>'))
    with pytest.raises(native.HtkAmdError) as ei:
        native.Mmf(files=[str(bad)], hmm_list=lst)
    assert "undefined ~d macro" in str(ei.value)


def test_mmf_shared_mixture_macros(native, tmp_path):
    """~m macros (HHEd TI on mixture components): the shared pdf is ONE Gaussian referenced by several components, and the
    set is written back exactly as the reference wrote it."""
    lst = os.path.join(GOLD, "mmf", "syn_list")
    m = native.Mmf(files=[os.path.join(GOLD, "mmf", "syn_tied.mmf")], hmm_list=lst)
    q = m.packed()
    assert q["numComp"] == 30 and q["numGauss"] < 30
    counts = np.bincount(q["compGauss"], minlength=q["numGauss"])
    assert counts.max() >= 2 and (counts >= 1).all()                 # some Gaussians are shared, none is orphaned
    out = tmp_path / "tied.mmf"
    m.write(dict(mean=q["mean"], var=q["var"], gconst=q["gconst"], compWeight=q["compWeight"], transP=q["transP"]), one_file=str(out))
    assert out.read_bytes() == open(os.path.join(GOLD, "mmf", "syn_tied.mmf"), "rb").read()
    with pytest.raises(native.HtkAmdError):
        m.write(dict(mean=q["mean"], var=q["var"], gconst=q["gconst"], compWeight=q["compWeight"], transP=q["transP"]), out_dir=str(tmp_path))


def test_mixture_splitting_matches_hhed(native, tmp_path):
    """HHEd's MU (MixUpCommand: heaviest component split into mean +- 0.2 sd clones with half the weight each) on HTKDemo's final
    models: MU 3 on every state, then MU +2 on S.state[2]; the file written afterwards equals the reference HHEd's byte for byte."""
    demo = os.path.join(GOLD, "demo")
    m = native.Mmf(hmm_list=os.path.join(demo, "bcplist"), hmm_dir=os.path.join(demo, "hmm_final"))
    q0 = m.packed()
    assert q0["numComp"] == 15
    m.mixup(3)
    q1 = m.packed()
    assert q1["numComp"] == 45 and q1["numGauss"] == 45 and (np.diff(q1["stateCompOff"]) == 3).all()
    assert np.allclose(np.add.reduceat(q1["compWeight"], q1["stateCompOff"][:-1]), 1.0, atol=1e-6)
    h = m.logical["S"]
    s2 = int(q1["hmmState"][q1["hmmStateOff"][h]])                  # S.state[2]
    m.mixup(-2, states=[s2])
    q = m.packed()
    assert q["numComp"] == 47 and int(q["stateCompOff"][s2 + 1] - q["stateCompOff"][s2]) == 5
    out = tmp_path / "newMacros"                                    # HHEd saves an edited set loaded from a directory as one file
    m.write(dict(mean=q["mean"], var=q["var"], gconst=q["gconst"], compWeight=q["compWeight"], transP=q["transP"]), one_file=str(out))
    assert out.read_bytes() == open(os.path.join(demo, "hmm_mixup", "newMacros"), "rb").read()
    # the split set loads again (structure is consistent)
    m2 = native.Mmf(files=[str(out)], hmm_list=os.path.join(demo, "bcplist"))
    assert m2.packed()["numComp"] == 47


def test_script_files_and_extended_file_names(native, tmp_path):
    """-S script files: words by white space or quotes (ScriptWord HShell.c:661), extended names logical=physical[s,e]
    (RegisterExtFileName HShell.c:86)."""
    p = tmp_path / "train.scp"
    p.write_text('data/a.mfc\n  "dir with space/b.mfc"   c.mfc\n\'q d.mfc\'\nutt1=big.mfc[100,250]\nseg.mfc[7,9]\nalias=phys.mfc\n')
    got = native.scp_read(str(p))
    assert got == [("data/a.mfc", "data/a.mfc", -1, -1), ("dir with space/b.mfc", "dir with space/b.mfc", -1, -1), ("c.mfc", "c.mfc", -1, -1),
                   ("q d.mfc", "q d.mfc", -1, -1), ("utt1", "big.mfc", 100, 250), ("seg.mfc", "seg.mfc", 7, 9), ("alias", "phys.mfc", -1, -1)]
    (tmp_path / "bad.scp").write_text('"unterminated\n')
    (tmp_path / "bad2.scp").write_text('x.mfc[5\n')
    for bad in ("bad.scp", "bad2.scp", "missing.scp"):
        with pytest.raises(native.HtkAmdError):
            native.scp_read(str(tmp_path / bad))
    (tmp_path / "empty.scp").write_text("\n  \n")
    assert native.scp_read(str(tmp_path / "empty.scp")) == []


def test_label_writer_columns_and_flags(native, tmp_path):
    """SaveHTKLabels: a score column is written only if some label of the list has a non-zero score in it (then for every label);
    times as "%.0f"; names that start with a quote are quoted; -o flags of FormatTranscription."""
    t = native.Trans(1)
    t.add(0, 1300000, "p1", -22.5, aux1="W1", aux1_score=0.0)
    t.add(1300000, 2600000, "p2", 0.0)
    t.add(2600000, 3000000, '"odd', -1.0, aux1="W2", aux1_score=0.0)
    t.write(str(tmp_path / "a.lab"))
    assert (tmp_path / "a.lab").read_text() == "0 1300000 p1 -22.500000 W1\n1300000 2600000 p2 0.000000\n2600000 3000000 \'\"odd\' -1.000000 W2\n"
    t.format(100000.0, models=True, flags="NT")
    t.write(str(tmp_path / "b.lab"))
    assert (tmp_path / "b.lab").read_text().splitlines()[0] == "p1 %f W1" % np.float32(np.float32(-22.5) / 13)
    t.format(100000.0, models=True, flags="SW")
    t.write(str(tmp_path / "c.lab"))
    assert (tmp_path / "c.lab").read_text() == "p1\np2\n\'\"odd\'\n"
    o = native.MlfOut(str(tmp_path / "o.mlf")); o.add("*/x.rec", t); o.close()
    assert (tmp_path / "o.mlf").read_text() == '#!MLF!#\n"*/x.rec"\np1\np2\n\'"odd\'\n.\n'
    back = native.Mlf(str(tmp_path / "o.mlf")).find("dir/x.rec")
    assert [l[0] for l in back] == ["p1", "p2", '"odd']


def test_mmf_shared_mean_and_variance_macros(native, tmp_path):
    """~u / ~v macros (GetMean / GetVariance HModel.c:1737-1790, SaveMacros :4342): the set HHEd tied (tests/golden/make_tied_golden.py)
    reads with every user of a vector holding its values and naming its macro, and is written back byte for byte."""
    d = os.path.join(GOLD, "demo")
    src = os.path.join(d, "hmm_tied", "newMacros")
    m = native.Mmf(files=[src], hmm_list=os.path.join(d, "bcplist"))
    q = m.packed()
    ms, vs = m.sharing()
    assert (ms >= 0).sum() == 2 and (vs >= 0).sum() == 4 + 3              # uSV: 2 means; vCL: 4 variances, vN: 3
    for share, vec in ((ms, q["mean"]), (vs, q["var"])):
        for k in set(share[share >= 0]):
            rows = vec[share == k]
            assert len(rows) > 1 and (rows == rows[0]).all()
    params = dict(mean=q["mean"], var=q["var"], gconst=q["gconst"], compWeight=q["compWeight"], transP=q["transP"])
    out = tmp_path / "tied.mmf"
    m.write(params, one_file=str(out))
    assert out.read_bytes() == open(src, "rb").read()
    # an unknown macro is an error, not a silent private copy
    bad = tmp_path / "bad.mmf"
    bad.write_text(open(src).read().replace('~u "uSV"\n<MEAN>', '~u "other"\n<MEAN>', 1))
    with pytest.raises(native.HtkAmdError) as e:
        native.Mmf(files=[str(bad)], hmm_list=os.path.join(d, "bcplist"))
    assert "undefined ~u macro" in str(e.value)


def test_mmf_numbers_are_read_as_strtof_reads_them(native, tmp_path):
    """The reader's own decimal -> float path (Clinger fast path through an exact double, strtof for everything near a rounding boundary,
    host/mmf.c fast_float) gives the float fscanf("%e") gives: ordinary %e numbers, 9-digit renderings of random bit patterns, midpoints
    of two floats written out in full, tiny, huge and odd spellings."""
    rng = np.random.default_rng(12)
    texts = ["%e" % x for x in rng.normal(0, 30, 400)] + ["%.9e" % x for x in rng.standard_normal(300).astype(np.float32)]
    bits = rng.integers(0x00800000, 0x7F000000, 400, dtype=np.uint64).astype(np.uint32)
    texts += ["%.8e" % x for x in bits.view(np.float32)]
    texts += ["16777217", "16777219", "1.00000005960464477539", "1.0000001788139343", "0.1", "-0.0", "1e-40", "3.4e38", "1e22", "1e23", "123456789012345678",
              ".5", "5.", "+2.5E+3", "1.17549435e-38", "7.00649232e-46", "0.000001", "4.9999997e-1"]
    D = len(texts)
    path = tmp_path / "m"
    path.write_text('~o <VecSize> %d <USER> <StreamInfo> 1 %d\n~h "a"\n<BeginHMM> <NumStates> 3 <State> 2 <Mean> %d\n%s\n<Variance> %d\n%s\n<TransP> 3\n0 1 0\n0 0.5 0.5\n0 0 0\n<EndHMM>\n'
                    % (D, D, D, " ".join(texts), D, " ".join(["1.0"] * D)))
    lst = tmp_path / "l"; lst.write_text("a\n")
    mmf = native.Mmf(files=[str(path)], hmm_list=str(lst))
    got = mmf.packed()["mean"].reshape(-1)
    want = np.array([np.float32(t) for t in texts], np.float32)              # numpy parses with strtod + one rounding... checked below
    import ctypes
    libc = ctypes.CDLL(None); libc.strtof.restype = ctypes.c_float; libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    want = np.array([libc.strtof(t.encode(), None) for t in texts], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), [(t, g, w) for t, g, w in zip(texts, got, want) if g != w or np.signbit(g) != np.signbit(w)][:5]


def test_mmf_numbers_are_written_as_printf_writes_them(native, tmp_path):
    """The writer's own "%e" (exact 128-bit integer arithmetic on m * 2^e * 10^p with ties to even, host/mmf.c format_e) against the C
    library's: random values, random bit patterns over the whole float range, round-up-to-the-next-power cases, denormals and zero (fprintf)."""
    import ctypes
    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.normal(0, 20, 300).astype(np.float32), rng.integers(0x00000001, 0x7F7FFFFF, 500, dtype=np.uint64).astype(np.uint32).view(np.float32),
                           np.array([9.9999995, 9.9999999e9, 0.99999994, 1.0, 0.0, 1e-45, 1.1754942e-38, 3.4028235e38, 123456.7, 9.5367431640625e-7], np.float32)])
    vals[::7] *= -1
    D = len(vals)
    path = tmp_path / "m"
    path.write_text('~o <VecSize> %d <USER> <StreamInfo> 1 %d\n~h "a"\n<BeginHMM> <NumStates> 3 <State> 2 <Mean> %d\n%s\n<Variance> %d\n%s\n<TransP> 3\n0 1 0\n0 0.5 0.5\n0 0 0\n<EndHMM>\n'
                    % (D, D, D, " ".join(["0.0"] * D), D, " ".join(["1.0"] * D)))
    lst = tmp_path / "l"; lst.write_text("a\n")
    mmf = native.Mmf(files=[str(path)], hmm_list=str(lst))
    pk = mmf.packed()
    out = tmp_path / "out"
    mmf.write(dict(mean=vals.reshape(1, D), var=pk["var"], gconst=None, compWeight=pk["compWeight"], transP=pk["transP"]), one_file=str(out))
    toks = out.read_text().split()
    i = toks.index("<MEAN>")
    got = toks[i + 2:i + 2 + D]
    libc = ctypes.CDLL(None)
    want = []
    for v in vals:
        b = ctypes.create_string_buffer(64)
        libc.snprintf(b, 64, b"%e", ctypes.c_double(float(v)))
        want.append(b.value.decode())
    assert got == want, [(g, w) for g, w in zip(got, want) if g != w][:5]


def test_set_from_several_master_files_is_saved_file_by_file(native, tmp_path):
    """`-H macros -H hmmdefs` (tests/golden/make_multimmf_golden.py): SaveHMMSet writes every macro back to the master file it was loaded
    from (HModel.c:4388-4470) -- both files equal the ones the reference's HHEd wrote, byte for byte; a model set saved that way loads
    again to the same description."""
    import filecmp
    D = os.path.join(DEMO, "hmm_multi")
    m = native.Mmf([os.path.join(D, "macros"), os.path.join(D, "hmmdefs")], hmm_list=os.path.join(DEMO, "bcplist"))
    pk = m.packed()
    outs = [str(tmp_path / "macros"), str(tmp_path / "hmmdefs")]
    m.write_sources(pk, outs, out_dir=str(tmp_path))
    assert filecmp.cmp(outs[0], os.path.join(D, "hhed_macros"), shallow=False)
    assert filecmp.cmp(outs[1], os.path.join(D, "hhed_hmmdefs"), shallow=False)
    again = native.Mmf(outs, hmm_list=os.path.join(DEMO, "bcplist")).packed()
    for k in ("mean", "var", "compWeight", "transP", "hmmState", "stateCompOff"):
        assert np.array_equal(again[k], pk[k]), k
    one = native.Mmf([os.path.join(DEMO, "hmm_tied", "newMacros")], hmm_list=os.path.join(DEMO, "bcplist")).packed()
    assert np.array_equal(one["mean"], pk["mean"]) and np.array_equal(one["var"], pk["var"])


def test_mmf_stream_weight_macros(native, tmp_path):
    """~w "name" <SWEIGHTS> S w1..wS and its use in a state where <SWEIGHTS> would stand (GetSWeights HModel.c:1621, GetStateInfo :1966):
    the set reads as the one with the weights written out in the states."""
    head = "~o <STREAMINFO> 2 1 1 <VECSIZE> 2 <NULLD><USER><DIAGC>\n"
    tr = '~t "T"\n<TRANSP> 3\n0 1 0\n0 0.5 0.5\n0 0 0\n'
    body = "<STREAM> 1\n<MEAN> 1\n 0.5\n<VARIANCE> 1\n 1.5\n<STREAM> 2\n<MEAN> 1\n -0.25\n<VARIANCE> 1\n 2.0\n"
    with_macro = head + '~w "sw" <SWEIGHTS> 2 1.0 0.5\n' + tr + '~h "a"\n<BEGINHMM>\n<NUMSTATES> 3\n<STATE> 2\n~w "sw"\n' + body + '~t "T"\n<ENDHMM>\n'
    inline = head + tr + '~h "a"\n<BEGINHMM>\n<NUMSTATES> 3\n<STATE> 2\n<SWEIGHTS> 2 1.0 0.5\n' + body + '~t "T"\n<ENDHMM>\n'
    (tmp_path / "m").write_text(with_macro); (tmp_path / "i").write_text(inline)
    a, b = native.Mmf(files=[str(tmp_path / "m")]).packed(), native.Mmf(files=[str(tmp_path / "i")]).packed()
    assert np.array_equal(a["streamWeight"], b["streamWeight"]) and list(a["streamWeight"]) == [1.0, 0.5]
    for k in ("mean", "var", "compWeight", "transP"):
        assert np.array_equal(a[k], b[k]), k
    (tmp_path / "u").write_text(head + tr + '~h "a"\n<BEGINHMM>\n<NUMSTATES> 3\n<STATE> 2\n~w "nope"\n' + body + '~t "T"\n<ENDHMM>\n')
    with pytest.raises(native.HtkAmdError):
        native.Mmf(files=[str(tmp_path / "u")])
