"""The drop-in boundary as a LINKED program: the reference's unchanged HERest.c object + HTKLib without its HFB module + this
repository's HFB (shim/htklib_hfb_shim.c) + libhtk_amd.so = oracle/_ref/HERest_amd (recipe: oracle/Makefile; built only where
/root/reference exists, shipped to the GPU box with the other oracle/_ref products).
CPU: the link resolved every symbol, the HFB entry points come from the shim, and without a device the run stops with the library's
ENODEV message.  GPU: HTKDemo's first embedded re-estimation pass run by that program reproduces the reference's own run -- the
log lines HERest prints and the models it writes."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "HERest_amd")
DEMO = os.path.join(os.path.dirname(__file__), "golden", "demo")
HFB_SYMBOLS = ["InitFB", "SetTraceFB", "InitialiseForBack", "UseAlignHMMSet", "InitUttInfo", "GetInputObs", "LoadLabs", "LoadData",
               "InitUttObservations", "FBFile", "PrLog", "SetMinDurs", "FindStateOrder"]

needs_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/HERest_amd not built (needs /root/reference: make -C oracle)")


def _demo_cmd(out_dir, conf):
    files = sorted(os.path.join(DEMO, "train", f) for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    return [EXE, "-T", "1", "-w", "3", "-v", "0.05", "-C", conf, "-u", "tmvw", "-d", os.path.join(DEMO, "hmm1"), "-M", out_dir,
            "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + files


@needs_exe
def test_link_resolves_and_hfb_comes_from_the_shim():
    nm = subprocess.run(["nm", EXE], capture_output=True, text=True, check=True).stdout
    defined = {l.split()[-1] for l in nm.splitlines() if len(l.split()) == 3 and l.split()[1] in "Tt"}
    undefined = {l.split()[-1] for l in nm.splitlines() if len(l.split()) == 2 and l.split()[0] == "U"}
    for s in HFB_SYMBOLS + ["__wrap_NewHMMScan", "NewHMMScan", "DumpAccs", "LoadAccs", "LoadHMMSet"]:
        assert s in defined, s
    # what stays undefined is resolved at load time: libc/libm and the C ABI of the HIP library
    ours = sorted(u for u in undefined if u.startswith("htkamd_"))
    assert "htkamd_fb_execute" in ours and "htkamd_model_create" in ours and "htkamd_accs_download" in ours
    assert all("@" in u or u.startswith("htkamd_") or u.startswith("_") for u in undefined), sorted(undefined)[:20]
    # the shim object itself imports only HTKLib's public API and ours
    obj = os.path.join(ROOT, "oracle", "_ref", "obj", "htklib_hfb_shim.o")
    und = subprocess.run(["nm", "-u", obj], capture_output=True, text=True, check=True).stdout.split()
    assert "FBFile" not in und and "htkamd_fb_prepare" in und and "ReadAsTable" in und and "__real_NewHMMScan" in und
    header = open(os.path.join(ROOT, "include", "htk_amd.h")).read()
    for u in und:
        if u.startswith("htkamd_"):
            assert re.search(r"\b%s\s*\(" % u, header), u            # every entry point the shim binds is declared in include/htk_amd.h


@needs_exe
def test_without_a_device_the_front_end_stops_with_enodev(native, tmp_path):
    if native.lib().htkamd_device_count() > 0:              # (asked of the library itself: torch, asked after it, has answered "none" on a GPU box)
        pytest.skip("a GPU is visible")
    conf = tmp_path / "herest.conf"
    conf.write_text("TARGETKIND = MFCC_E_D\n")
    r = subprocess.run(_demo_cmd(str(tmp_path), str(conf)), capture_output=True, text=True)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "7399" in out and "no HIP device" in out and "HTKAMD_ENODEV" in out, out[-600:]
    assert not (tmp_path / "S").exists()                               # nothing was re-estimated on the way


@pytest.mark.gpu
@needs_exe
def test_reference_herest_front_end_runs_its_e_step_on_the_gpu(native, tmp_path):
    conf = tmp_path / "herest.conf"
    conf.write_text("TARGETKIND = MFCC_E_D\n")
    r = subprocess.run(_demo_cmd(str(tmp_path), str(conf)), capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    ref_log = open(os.path.join(DEMO, "herest_pass1.log")).read()
    for pat in (r"Pruning-On\[2000\.0\]", r"Total 27 floored variance elements in 15 different mixes",
                r"average log prob per frame = -5\.900196e\+01", r"total frames seen\s+= 1\.811000e\+03"):
        assert re.search(pat, ref_log) and re.search(pat, r.stdout), (pat, r.stdout[-800:])
    assert r.stdout.count("Utterance prob per frame") == 7             # one FBFile per training file, served by the library
    def by_model(mmf):
        """name -> (means, variances, gconsts, transition matrix) of a model, whatever order the file defined things in"""
        pk, out = mmf.packed(), {}
        for name, h in mmf.logical.items():
            st = pk["hmmState"][pk["hmmStateOff"][h]:pk["hmmStateOff"][h + 1]]
            g = np.concatenate([pk["compGauss"][pk["stateCompOff"][x]:pk["stateCompOff"][x + 1]] for x in st])
            t = pk["hmmTrans"][h]
            out[name] = (pk["mean"][g], pk["var"][g], pk["gconst"][g], pk["transP"][pk["transOff"][t]:pk["transOff"][t + 1]])
        return out

    ref = by_model(native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=os.path.join(DEMO, "hmm2_expected")))
    # the front-end saves its set the way it loaded it: one file per model, or everything in `newMacros` (SaveHMMSet HModel.c:4979)
    if (tmp_path / "newMacros").exists():
        got = by_model(native.Mmf(files=[str(tmp_path / "newMacros")], hmm_list=os.path.join(DEMO, "bcplist")))
    else:
        got = by_model(native.Mmf(hmm_list=os.path.join(DEMO, "bcplist"), hmm_dir=str(tmp_path)))
    lin = lambda t: np.where(t > -0.5e10, np.exp(t.astype(np.float64)), 0.0)
    assert sorted(ref) == sorted(got) and len(ref) == 5
    for name, (rm, rv, rg, rt) in ref.items():
        gm, gv, gg, gt = got[name]
        assert (np.abs(gm - rm) <= 1e-4 * np.maximum(np.abs(rm), np.sqrt(rv)) + 1e-6).all(), name
        assert np.allclose(gv, rv, rtol=1e-4, atol=1e-7), name
        assert np.allclose(gg, rg, rtol=1e-5), name
        assert np.allclose(lin(gt), lin(rt), rtol=1e-4, atol=1e-7), name


@pytest.mark.gpu
@needs_exe
def test_shim_exact_switch_gives_the_references_lines(native, tmp_path):
    """Under the reference's own HERest.o the shim scores exactly and, by default, takes the recursions' log-adds from the fp32 transcendental unit
    (tolerance class): every "Utterance prob per frame" line HERest -T 1 prints is within 1e-6 relative of the reference binary's.  With
    HTKAMD_SHIM_EXACT=1 (table-driven log-add) the lines (%e: seven digits of a double that is then the reference's bit for bit) are EQUAL."""
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "HERest")
    if not os.path.exists(ref_exe):
        pytest.skip("oracle/_ref/HERest is not on this box")
    conf = tmp_path / "herest.conf"
    conf.write_text("TARGETKIND = MFCC_E_D\n")
    lines = {}
    for tag, exe, env in (("ref", ref_exe, {}), ("shim", EXE, {"HTKAMD_SHIM_EXACT": "1"}), ("fast", EXE, {})):
        out = tmp_path / tag
        out.mkdir()
        cmd = _demo_cmd(str(out), str(conf))
        cmd[0] = exe
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
        lines[tag] = [l.strip() for l in r.stdout.splitlines() if "Utterance prob per frame" in l]
        assert len(lines[tag]) == 7, r.stdout[-800:]
    assert lines["shim"] == lines["ref"]
    for a, b in zip(lines["fast"], lines["ref"]):
        x, y = float(a.split("=")[1]), float(b.split("=")[1])
        assert abs(x - y) <= 1e-6 * abs(y), (a, b)


@pytest.mark.gpu
@needs_exe
def test_reference_herest_front_end_on_a_three_stream_set(native, tmp_path):
    """The reference's HERest.o over the shim on a multi-stream set (tests/golden/demo/hmm_streams3): the shim packs every StreamElem,
    hands the library undivided rows, and adds the statistics back per stream; the front-end's own UpdateModels then writes the model the
    reference's HERest writes."""
    import test_cli_tools as cli
    d3 = os.path.join(DEMO, "hmm_streams3")
    conf = tmp_path / "herest.conf"
    conf.write_text("TARGETKIND = MFCC_E_D\n")
    files = sorted(os.path.join(DEMO, "train", f) for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    r = subprocess.run([EXE, "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(d3, "newMacros"), "-M", str(tmp_path),
                        "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + files, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    for line in open(os.path.join(d3, "herest.log")).read().splitlines():
        assert line in r.stdout, (line, r.stdout[-600:])
    cli._mmf_close(cli._mmf_numbers(str(tmp_path / "newMacros")), cli._mmf_numbers(os.path.join(d3, "after_herest")))


@pytest.mark.gpu
@needs_exe
def test_reference_herest_front_end_on_a_two_stream_set(native, tmp_path):
    """... and on the set split into TWO streams, where the reference's Setotprob gives a tied state met again half its log probability
    (HFB.c:1059): under the reference's own HERest.o the shim asks the library for that arithmetic (htkamd_model_set_compat), so the
    drop-in writes the reference's summary lines and model there too."""
    import test_cli_tools as cli
    d2 = os.path.join(DEMO, "hmm_streams2")
    conf = tmp_path / "herest.conf"
    conf.write_text("TARGETKIND = MFCC_E_D\n")
    files = sorted(os.path.join(DEMO, "train", f) for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))
    r = subprocess.run([EXE, "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(d2, "newMacros"), "-M", str(tmp_path),
                        "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + files, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    for line in open(os.path.join(d2, "herest.log")).read().splitlines():
        assert line in r.stdout, (line, r.stdout[-600:])
    cli._mmf_close(cli._mmf_numbers(str(tmp_path / "newMacros")), cli._mmf_numbers(os.path.join(d2, "after_herest")))


# ---------------------------------------------------------------------------------------------------------------- HVite
HVITE = os.path.join(ROOT, "oracle", "_ref", "HVite_amd")
needs_hvite = pytest.mark.skipif(not os.path.exists(HVITE), reason="oracle/_ref/HVite_amd not built (needs /root/reference: make -C oracle)")


def _hvite_cmd(out_dir, conf, part, names):
    return [HVITE, "-C", conf, "-d", os.path.join(DEMO, "hmm_final"), "-w", os.path.join(DEMO, "monLattice"), "-l", out_dir, "-t", "300.0", "-p", "5.0",
            "-s", "0.0", os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + [os.path.join(DEMO, part, u + ".mfc") for u in names]


@needs_hvite
def test_hvite_link_routes_the_recogniser_entry_points_through_the_shim(tmp_path):
    nm = subprocess.run(["nm", HVITE], capture_output=True, text=True, check=True).stdout
    defined = {l.split()[-1] for l in nm.splitlines() if len(l.split()) == 3 and l.split()[1] in "Tt"}
    for s in ["__wrap_StartRecognition", "__wrap_ProcessObservation", "__wrap_CompleteRecognition", "__wrap_InitPSetInfo", "__wrap_InitVRecInfo",
              "StartRecognition", "TranscriptionFromLattice", "FormatTranscription", "ExpandWordNet", "ReadLattice", "OpenBuffer", "FBFile"]:
        assert s in defined, s
    obj = os.path.join(ROOT, "oracle", "_ref", "obj", "htklib_hrec_shim.o")
    und = subprocess.run(["nm", "-u", obj], capture_output=True, text=True, check=True).stdout.split()
    assert "htkamd_decoder_run_out" in und and "NewLattice" in und and "__real_InitPSetInfo" in und
    import torch
    if not torch.cuda.is_available():
        import json
        conf = tmp_path / "c"; conf.write_text("TARGETKIND = MFCC_E_D\n")
        names = sorted(json.load(open(os.path.join(DEMO, "hvite_expected.json")))["test"])
        r = subprocess.run(_hvite_cmd(str(tmp_path), str(conf), "test", names), capture_output=True, text=True)
        out = r.stdout + r.stderr
        assert r.returncode != 0 and "7399" in out and "no HIP device" in out, out[-600:]


@pytest.mark.gpu
@needs_hvite
def test_reference_hvite_front_end_recognises_on_the_gpu(tmp_path):
    """The reference's HVite.c, its HNet network expansion, HParm buffers and label writer unchanged; StartRecognition /
    ProcessObservation / CompleteRecognition served by the library: every .rec file of HTKDemo's test and training sets equals the
    one the reference's own HVite wrote."""
    import json
    expected = json.load(open(os.path.join(DEMO, "hvite_expected.json")))
    conf = tmp_path / "c"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    n = 0
    for part in ("test", "train"):
        names = sorted(expected[part])
        out = tmp_path / part; out.mkdir()
        r = subprocess.run(_hvite_cmd(str(out), str(conf), part, names), capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
        for u in names:
            got = (out / (u + ".rec")).read_text().splitlines()
            assert got == expected[part][u], (part, u, got[:3], expected[part][u][:3])
            n += len(got)
    assert n == 292


@pytest.mark.gpu
@needs_hvite
@pytest.mark.parametrize("name", ["after_herest", "sw_after_herest"])
def test_reference_hvite_front_end_on_a_three_stream_set(tmp_path, name):
    """The reference's HVite.o over the recogniser shim on a multi-stream set, stream weights 1 1 1 and 1 0.5 2: the label files of the
    reference's own HVite (tests/golden/make_streams_hvite_golden.py), word-level recognition."""
    import json
    d3 = os.path.join(DEMO, "hmm_streams3")
    exp = json.load(open(os.path.join(d3, "hvite_expected.json")))[name]
    conf = tmp_path / "hvite.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    path = lambda u: os.path.join(DEMO, "test" if u.startswith("te") else "train", u + ".mfc")
    for what, opts in (("recw", ["-w", os.path.join(DEMO, "monLattice"), "-t", "300.0", "-p", "5.0", "-s", "0.0"]),):       # word level: what this shim serves
        out = tmp_path / what; out.mkdir()
        names = sorted(exp[what])
        r = subprocess.run([HVITE, "-C", str(conf), "-H", os.path.join(d3, name), "-l", str(out)] + opts + [os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] +
                           [path(u) for u in names], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
        for u in names:
            assert (out / (u + ".rec")).read_text().splitlines() == exp[what][u], (what, u)


@pytest.mark.gpu
@needs_hvite
@pytest.mark.parametrize("tag", ["bigram_n4_t2500", "loop_n3_t2500", "tee_n3_t2500"])
def test_reference_hvite_front_end_nbest_and_lattices_on_the_gpu(tmp_path, tag):
    """HVite -n i [N] -z lat through the shim: CompleteRecognition hands the reference's HVite.c the Lattice built from the token-set
    kernel's nodes and arcs; its own WriteLattice / TranscriptionFromLattice then write the files the all-reference HVite wrote
    (tests/golden/decode/nbest)."""
    import json
    import numpy as np
    from htk_amd import synth
    gold = os.path.join(os.path.dirname(__file__), "golden", "decode")
    meta = json.load(open(os.path.join(gold, "nbest", tag, "nbest.json")))
    src = os.path.join(gold, meta["case"])
    for fn in ("MMF", "dict", "hmmlist", meta["slf"] + ".slf"):
        os.symlink(os.path.join(src, fn), str(tmp_path / fn))
    (tmp_path / "nbtmp").mkdir()
    (tmp_path / "nbtmp" / "config").write_text("")
    z = np.load(os.path.join(src, meta["feats"] + ".npz"))
    files = []
    for u in range(len(z.files)):
        synth.write_htk_param(str(tmp_path / "nbtmp" / ("u%d.mfc" % u)), z["u%d" % u], kind=9)
        files.append("nbtmp/u%d.mfc" % u)
    base = [HVITE, "-C", "nbtmp/config", "-H", "MMF", "-w", meta["slf"] + ".slf"] + meta["opts"].split()
    r = subprocess.run(base + ["-l", "nbtmp", "-n", str(meta["nToks"]), "1", "-z", "lat", "dict", "hmmlist"] + files, cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    def same(a, b):
        # Here the network is the one the reference's HNet built, walked in ITS node order: tokens reach a node in another order than
        # in tools/hvite's network, and a relative token's likelihood -- a float re-based at every TokSetMerge (HRec.c:361) -- may end
        # up one bit away.  Visible only where an alternative's acoustic score is within that bit of a printing boundary
        # (a=-0.00 / a=0.00): everything but the second decimal of a= must agree.
        la, lb = a.splitlines(), b.splitlines()
        assert len(la) == len(lb)
        for x, y in zip(la, lb):
            if x != y:
                fx, fy = x.split(), y.split()
                assert x.startswith("J=") and len(fx) == len(fy), (x, y)
                for p_, q_ in zip(fx, fy):
                    assert p_ == q_ or (p_.startswith("a=") and abs(float(p_[2:]) - float(q_[2:])) <= 0.0101), (x, y)
    for u in range(len(files)):
        same((tmp_path / "nbtmp" / ("u%d.lat" % u)).read_text(), open(os.path.join(gold, "nbest", tag, "u%d.lat" % u)).read())
    r = subprocess.run(base + ["-i", "nb.mlf", "-n", str(meta["nToks"]), str(meta["nTrans"]), "dict", "hmmlist"] + files, cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    want = ["#!MLF!#"]
    for u in range(len(files)):
        want.append('"nbtmp/u%d.rec"' % u)
        want += "\n///\n".join("\n".join(a) for a in meta["nbest"]["u%d" % u]).split("\n") + ["."]
    got = (tmp_path / "nb.mlf").read_text().splitlines()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if g != w:                                           # scores of alternatives: the same float noise, 6 printed decimals
            fg, fw = g.split(), w.split()
            assert fg[:3] == fw[:3] and abs(float(fg[3]) - float(fw[3])) < 1e-3, (g, w)


# ---- HDecode's block scorer (HTKLVRec/HLVModel.h:OutPBlock, HLVRec-outP.c:OutPBlock_HMod) served by the library ------------------
OUTPBLOCK = os.path.join(ROOT, "oracle", "_ref", "ref_outpblock")
needs_outpblock = pytest.mark.skipif(not os.path.exists(OUTPBLOCK), reason="oracle/_ref/ref_outpblock not built (needs /root/reference)")


def _outpblock_cmd(conv, block, ac, conf=None):
    data = sorted(os.path.join(DEMO, "train", f) for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))[0]
    return [OUTPBLOCK] + (["-C", conf] if conf else []) + (["-c"] if conv else []) + ["-d", os.path.join(DEMO, "hmm1"), os.path.join(DEMO, "bcplist"), data, str(block), str(ac)]


@needs_outpblock
def test_block_scorer_shim_exports_hdecodes_two_entry_points():
    obj = os.path.join(ROOT, "oracle", "_ref", "obj", "hlvmodel_outp_shim.o")
    nm = subprocess.run(["nm", obj], capture_output=True, text=True, check=True).stdout
    defined = {l.split()[-1] for l in nm.splitlines() if len(l.split()) == 3 and l.split()[1] == "T"}
    assert {"OutPBlock", "OutPBlock_HMod"} <= defined
    und = subprocess.run(["nm", "-u", obj], capture_output=True, text=True, check=True).stdout.split()
    assert "htkamd_outp_block_mode" in und and "htkamd_model_create" in und
    header = open(os.path.join(ROOT, "include", "htk_amd.h")).read()
    for u in und:
        if u.startswith("htkamd_"):
            assert re.search(r"\b%s\s*\(" % u, header), u


@needs_outpblock
def test_block_scorer_shim_without_a_device_stops_with_enodev(native):
    if native.lib().htkamd_device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run(_outpblock_cmd(False, 4, 1.0), capture_output=True, text=True)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and "no HIP device" in out and "HTKAMD_ENODEV" in out, out[-600:]


@pytest.mark.gpu
@needs_outpblock
@pytest.mark.parametrize("conv,block,ac", [(False, 4, 1.0), (True, 4, 1.0), (True, 7, 0.5), (False, 1, 1.0 / 13)])
def test_block_scorer_shim_equals_the_references_outp_bit_for_bit(conv, block, ac, tmp_path):
    conf = tmp_path / "c.conf"
    conf.write_text("TARGETKIND = MFCC_E_D\n")
    r = subprocess.run(_outpblock_cmd(conv, block, ac, str(conf)), capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-800:]
    m = re.search(r"scores (\d+) mismatches (\d+) maxdiff (\S+)", r.stdout)
    assert m and int(m.group(1)) > 1000 and int(m.group(2)) == 0, r.stdout
