"""The command-line drivers tools/herest.c and tools/hvite.c (plain C over include/htk_amd.h, built by htk_amd.build into tools/bin/):
the programs a user of the reference's HERest / HVite switches to, with the reference's switches.
CPU: they build, refuse to run without a device (no fallback) and reject unknown switches.  GPU: HTKDemo's re-estimation pass and its
recognition step through the CLIs against the reference's logs / models / label files; HERest's parallel mode (-p N dumps, -p 0 merge);
word-level alignment and HVite's output formats against the committed HVite files."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")
DEMO = os.path.join(os.path.dirname(__file__), "golden", "demo")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "decode")


@pytest.fixture(scope="module")
def tools(native):
    from htk_amd import build as nbuild
    nbuild.build_tools()
    for t in ("herest", "hvite"):
        assert os.path.exists(os.path.join(BIN, t)), t
    return BIN


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, **kw)


def demo_train_files():
    return sorted(os.path.join(DEMO, "train", f) for f in os.listdir(os.path.join(DEMO, "train")) if f.endswith(".mfc"))


def herest_demo_cmd(tools, conf, out, extra=()):
    return [os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", conf, "-u", "tmvw", "-d", os.path.join(DEMO, "hmm1"), "-M", out,
            "-L", os.path.join(DEMO, "labels"), "-t", "2000.0"] + list(extra) + [os.path.join(DEMO, "bcplist")]


def test_tools_refuse_to_run_without_a_device_and_check_their_switches(tools, tmp_path):
    import torch
    conf = tmp_path / "c"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    r = run([os.path.join(tools, "herest"), "-Q", "x", os.path.join(DEMO, "bcplist"), "f.mfc"])
    assert r.returncode != 0 and "unknown switch -Q" in r.stderr
    r = run([os.path.join(tools, "hvite"), os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist"), "f.mfc"])
    assert r.returncode != 0 and "either -w net or -a" in r.stderr
    if not torch.cuda.is_available():
        r = run(herest_demo_cmd(tools, str(conf), str(tmp_path)) + demo_train_files())
        assert r.returncode != 0 and "no HIP device" in r.stderr
        r = run([os.path.join(tools, "hvite"), "-w", os.path.join(DEMO, "monLattice"), os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist"), "f.mfc"])
        assert r.returncode != 0 and "no HIP device" in r.stderr


def _same_models(out_dir, ref_dir, tol=1e-4):
    for name in "SCVNL":
        ours = open(os.path.join(out_dir, name)).read().split()
        theirs = open(os.path.join(ref_dir, name)).read().split()
        assert len(ours) == len(theirs), name
        for x, y in zip(ours, theirs):
            if x != y:
                assert abs(float(x) - float(y)) <= tol * max(abs(float(y)), 1e-3), (name, x, y)


@pytest.mark.gpu
def test_herest_cli_runs_the_demo_pass(tools, tmp_path):
    conf = tmp_path / "herest.conf"; conf.write_text("# HTKDemo/toolconfs/herest.conf\nTARGETKIND = MFCC_E_D\n")
    out = tmp_path / "hmm2"; out.mkdir()
    scp = tmp_path / "train.scp"; scp.write_text("\n".join(demo_train_files()[2:]) + "\n")
    r = run(herest_demo_cmd(tools, str(conf), str(out), ["-S", str(scp)]) + demo_train_files()[:2])
    assert r.returncode == 0, r.stderr
    log = open(os.path.join(DEMO, "herest_pass1.log")).read()
    for line in ("Pruning-On[2000.0]", "Total 27 floored variance elements in 15 different mixes",
                 "Reestimation complete - average log prob per frame = -5.900196e+01", "     - total frames seen          = 1.811000e+03"):
        assert line in log and line in r.stdout, (line, r.stdout[-600:])
    assert r.stdout.count("Utterance prob per frame") == 7
    _same_models(str(out), os.path.join(DEMO, "hmm2_expected"))


@pytest.mark.gpu
def test_herest_cli_parallel_mode_dump_and_merge(tools, tmp_path):
    """HERest -p 1 / -p 2 over two halves of the data, then -p 0 over the two HERn.acc files: the merged re-estimation equals the
    single-process one (HERest.c:502-557), and the dump files are the reference's byte layout (tests/test_accio.py reads them)."""
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    files = demo_train_files()
    accdir = tmp_path / "acc"; accdir.mkdir()
    for k, part in ((1, files[:4]), (2, files[4:])):
        r = run(herest_demo_cmd(tools, str(conf), str(accdir), ["-p", str(k)]) + part)
        assert r.returncode == 0 and (accdir / ("HER%d.acc" % k)).exists(), r.stderr
    out = tmp_path / "hmm2"; out.mkdir()
    r = run(herest_demo_cmd(tools, str(conf), str(out), ["-p", "0"]) + [str(accdir / "HER1.acc"), str(accdir / "HER2.acc")])
    assert r.returncode == 0, r.stderr
    assert "average log prob per frame = -5.900196e+01" in r.stdout and "Total 27 floored variance elements in 15 different mixes" in r.stdout
    _same_models(str(out), os.path.join(DEMO, "hmm2_expected"))


def _mmf_numbers(path):
    """All tokens of an MMF; numbers as floats, everything else as text."""
    out = []
    for t in open(path).read().split():
        try:
            out.append(float(t))
        except ValueError:
            out.append(t)
    return out


def _mmf_close(ours, theirs, tol=1e-4):
    """Token-by-token comparison of two text MMFs (tokens from _mmf_numbers): names and keywords equal, numbers within `tol` of the
    reference's -- relative for variances, weights, transition probabilities and gConsts, and relative to max(|mean|, sigma_i) for the
    elements of a mean vector (a mean near zero has no scale of its own).  sigma_i comes from the variance vector that
    follows the mean in the file (the same mixture component); a mean without one (a ~u macro, a component with a ~v reference) takes
    the median over the file's variance vectors at element i."""
    assert len(ours) == len(theirs)
    cols, i = {}, 0
    while i < len(theirs):
        if theirs[i] == "<VARIANCE>":
            n = int(theirs[i + 1])
            for k in range(n):
                cols.setdefault(k, []).append(theirs[i + 2 + k])
            i += 2 + n
        else:
            i += 1
    vmed = {k: float(np.median(v)) for k, v in cols.items()}
    in_mean, left, pos, own = False, 0, 0, None
    for idx, (x, y) in enumerate(zip(ours, theirs)):
        if not isinstance(y, float):
            assert x == y, (idx, x, y)
            in_mean, left = (y == "<MEAN>"), (-1 if y in ("<MEAN>", "<VARIANCE>") else 0)
            continue
        assert isinstance(x, float), (idx, x, y)
        if left == -1:                                        # the vector's length
            assert x == y
            left = int(y); pos = 0
            j = idx + 1 + left                                # the variance vector of the same component, if it follows
            own = theirs[j + 2:j + 2 + left] if in_mean and j + 1 < len(theirs) and theirs[j] == "<VARIANCE>" else None
            continue
        scale = max(abs(y), 1e-3)
        if in_mean and left > 0:
            scale = max(abs(y), float(np.sqrt(own[pos] if own is not None else vmed.get(pos, 0.0))))
        if left > 0:
            left -= 1; pos += 1
        assert abs(x - y) <= tol * scale, (idx, x, y, scale)


@pytest.mark.gpu
def test_herest_cli_tied_mean_and_variance_vectors(tools, tmp_path):
    """A set with ~u / ~v macros (HHEd TI on means and variances, tests/golden/make_tied_golden.py) through one pass: the statistics of
    the users of a vector are pooled, a tied variance gets no mean-shift term, every user keeps the shared value (UpdateVars /
    UpdateMeans HERest.c:974-1122) -- the MMF equals the one the reference's HERest wrote, macros included."""
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    out = tmp_path / "next"; out.mkdir()
    tied = os.path.join(DEMO, "hmm_tied")
    r = run([os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(tied, "newMacros"), "-M", str(out),
             "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0, r.stderr
    for line in open(os.path.join(tied, "herest.log")).read().splitlines():
        assert line in r.stdout, (line, r.stdout[-400:])
    ours, theirs = _mmf_numbers(str(out / "newMacros")), _mmf_numbers(os.path.join(tied, "after_herest"))
    _mmf_close(ours, theirs)
    assert open(str(out / "newMacros")).read().count('~v "vCL"') == 5 and open(str(out / "newMacros")).read().count('~u "uSV"') == 3


@pytest.mark.gpu
def test_herest_cli_mean_tied_across_models_follows_the_scan_order(tools, tmp_path):
    """Means tied across two models with private variances (tests/golden/make_tied2_golden.py), in a set whose HMM scan order (C L N S V)
    differs from the order of definition (S C V N L): the variance of the first mixture to reach the shared mean IN SCAN ORDER carries the
    mean-shift term (UpdateVars, HERest.c:1045-1122) -- the MMF equals the reference's."""
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    out = tmp_path / "next"; out.mkdir()
    tied = os.path.join(DEMO, "hmm_tied2")
    r = run([os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(tied, "newMacros"), "-M", str(out),
             "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0, r.stderr
    for line in open(os.path.join(tied, "herest.log")).read().splitlines():
        assert line in r.stdout, (line, r.stdout[-400:])
    _mmf_close(_mmf_numbers(str(out / "newMacros")), _mmf_numbers(os.path.join(tied, "after_herest")))


@pytest.mark.gpu
def test_herest_cli_tied_mixture_pool_in_shared_form(tools, tmp_path):
    """The tied-mixture pool WITHOUT `HK TIEDHS` (tests/golden/make_tmix_golden.py: ~m macros inside ordinary mixtures, hsKind SHAREDHS).
    The reference's HERest mistreats such a set: ConvLogWt (HUtil.c:474-485) walks the mixtures with GoNextMix(noSkip = FALSE), which skips
    a shared pdf after its first visit, so only the FIRST state's weights become logarithms and every other state's linear weights are
    then read as log weights -- -59.47 per frame on this data.  This library scores the mixtures as written (-61.00, the value the
    reference's own TIEDHS arithmetic gives to within its pruning, tiedhs.log); the test pins that number and that the run works."""
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    out = tmp_path / "next"; out.mkdir()
    tm = os.path.join(DEMO, "hmm_tmix")
    r = run([os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(tm, "newMacros"), "-M", str(out),
             "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0, r.stderr
    assert "average log prob per frame = -6.09970" in r.stdout and "-5.946912e+01" in open(os.path.join(tm, "herest.log")).read()
    assert open(str(out / "newMacros")).read().count('~m "MIX_') == 8 + 15 * 8
    # ... and with --compat (HTKAMD_COMPAT_SHARED_LOGWT) the reference's reading of those weights: its summary lines and the model it wrote
    out2 = tmp_path / "compat"; out2.mkdir()
    r = run([os.path.join(tools, "herest"), "--compat", "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(tm, "newMacros"), "-M", str(out2),
             "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0, r.stderr
    for line in open(os.path.join(tm, "herest.log")).read().splitlines():
        assert line in r.stdout, (line, r.stdout[-400:])
    _mmf_close(_mmf_numbers(str(out2 / "newMacros")), _mmf_numbers(os.path.join(tm, "after_herest")))


@pytest.mark.gpu
def test_herest_cli_several_master_files_round_trip(tools, tmp_path):
    """The usual iteration `-H dir/macros -H dir/hmmdefs -M next`: the re-estimated macros go back to next/macros and next/hmmdefs as the
    reference's SaveHMMSet writes them (tests/golden/make_multimmf_golden.py: herest_macros / herest_hmmdefs from its HERest), and the
    next iteration starts from those two files."""
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    multi = os.path.join(DEMO, "hmm_multi")
    out = tmp_path / "next"; out.mkdir()
    base = [os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-L", os.path.join(DEMO, "labels"), "-t", "2000.0"]
    r = run(base + ["-H", os.path.join(multi, "macros"), "-H", os.path.join(multi, "hmmdefs"), "-M", str(out), os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0, r.stderr
    for name in ("macros", "hmmdefs"):
        _mmf_close(_mmf_numbers(str(out / name)), _mmf_numbers(os.path.join(multi, "herest_" + name)))
    out2 = tmp_path / "next2"; out2.mkdir()
    r = run(base + ["-H", str(out / "macros"), "-H", str(out / "hmmdefs"), "-M", str(out2), os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0 and (out2 / "macros").exists() and (out2 / "hmmdefs").exists(), r.stderr


@pytest.mark.gpu
def test_herest_cli_repeats_an_iteration_the_fp16_scores_cannot_hold(tools, tmp_path):
    """--score fastest (fp16 x 2 scores) on data with one value far outside anything the models describe (3e4 in one coefficient of one
    frame): the pass reports HTKAMD_ERANGE, the tool says so and repeats the iteration with the bf16 x 3 scores -- the models it writes
    are the ones --score bf16 writes, byte for byte."""
    import shutil
    from htk_amd import capi
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    data = tmp_path / "data"; data.mkdir()
    files = []
    for i, f in enumerate(demo_train_files()):
        dst = str(data / os.path.basename(f))
        if i == 1:
            X, per, kind = capi.parm_read(f)
            X[7, 2] = 3.0e4
            capi.parm_write(dst, X, per, kind)
        else:
            shutil.copy(f, dst)
        files.append(dst)
    outs = {}
    for mode in ("fastest", "bf16"):
        out = tmp_path / mode; out.mkdir()
        r = run(herest_demo_cmd(tools, str(conf), str(out), ["--score", mode, "--batch", "3"]) + files)
        assert r.returncode == 0, r.stderr
        assert ("repeating the iteration with the bf16 x 3 scoring path" in r.stderr) == (mode == "fastest"), r.stderr[-800:]
        outs[mode] = {n: open(os.path.join(str(out), n), "rb").read() for n in "SCVNL"}
        assert r.stdout.count("Reestimation complete") == 1
    assert outs["fastest"] == outs["bf16"]


@pytest.mark.gpu
def test_herest_cli_iterations_in_one_process_equal_a_chain_of_runs(tools, tmp_path):
    """--iterations 3 (features, transcriptions, batch tables and the model stay on the device; only the last set is written) against
    three HERest-style runs chained through binary model files: the same models (the second and third iteration of the chain start
    from exactly the floats the first wrote), the same summary figure per iteration."""
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    base = [os.path.join(tools, "herest"), "-T", "0", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", "-B"]
    src = os.path.join(DEMO, "hmm_mixup", "newMacros")
    logs = []
    for k in range(3):
        out = tmp_path / ("chain%d" % (k + 1)); out.mkdir()
        r = run(base + ["-H", src, "-M", str(out), os.path.join(DEMO, "bcplist")] + demo_train_files())
        assert r.returncode == 0, r.stderr
        logs.append(re.search(r"average log prob per frame = (\S+)", r.stdout).group(1))
        src = str(out / "newMacros")
    one = tmp_path / "one"; one.mkdir()
    r = run(base + ["--iterations", "3", "-H", os.path.join(DEMO, "hmm_mixup", "newMacros"), "-M", str(one), os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0, r.stderr
    got = re.findall(r"average log prob per frame = (\S+)", r.stdout)
    assert got == logs, (got, logs)
    assert float(logs[2]) > float(logs[1]) > float(logs[0])
    from htk_amd import capi
    a = capi.Mmf([str(one / "newMacros")], hmm_list=os.path.join(DEMO, "bcplist")).packed()
    b = capi.Mmf([src], hmm_list=os.path.join(DEMO, "bcplist")).packed()
    # (a chained run derives 1/variance and the log weights with the host's libm at load, the resident model with the device's at the
    # update: last-bit differences in a handful of derived values, hence 1e-5 rather than equality after three iterations)
    sigma = np.sqrt(b["var"])
    assert (np.abs(a["mean"] - b["mean"]) <= 1e-5 * np.maximum(np.abs(b["mean"]), sigma)).all()
    for k in ("var", "compWeight", "transP"):
        assert np.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), k


@pytest.mark.gpu
def test_herest_cli_parallel_mode_with_tied_vectors(tools, tmp_path):
    """-p 1 / -p 2 / -p 0 on the set with ~u / ~v vectors: a shared vector has one record per dump (HTrain.c:1484-1493; byte layout
    checked against the reference in tests/test_accio.py); the merged update equals the reference's single-process pass."""
    conf = tmp_path / "herest.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    tied = os.path.join(DEMO, "hmm_tied")
    files = demo_train_files()
    accdir = tmp_path / "acc"; accdir.mkdir()
    base = [os.path.join(tools, "herest"), "-T", "1", "-w", "3", "-v", "0.05", "-C", str(conf), "-u", "tmvw", "-H", os.path.join(tied, "newMacros"),
            "-L", os.path.join(DEMO, "labels"), "-t", "2000.0"]
    for k, part in ((1, files[:3]), (2, files[3:])):
        r = run(base + ["-M", str(accdir), "-p", str(k), os.path.join(DEMO, "bcplist")] + part)
        assert r.returncode == 0 and (accdir / ("HER%d.acc" % k)).exists(), r.stderr
    out = tmp_path / "next"; out.mkdir()
    r = run(base + ["-M", str(out), "-p", "0", os.path.join(DEMO, "bcplist"), str(accdir / "HER1.acc"), str(accdir / "HER2.acc")])
    assert r.returncode == 0, r.stderr
    ours, theirs = _mmf_numbers(str(out / "newMacros")), _mmf_numbers(os.path.join(tied, "after_herest"))
    _mmf_close(ours, theirs)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,conf", [("pmvw", "HMAP: MAPTAU = 6.0\nHMAP: MINVAR = 0.02\nHMAP: MIXWEIGHTFLOOR = 2.0\nHMAP: TRACE = 1\n"), ("pm", "HMAP: TRACE = 1\n"),
                                        ("tied_pmv", "HMAP: MAPTAU = 3.0\nHMAP: TRACE = 1\n")])
def test_herest_cli_map_reestimation(tools, tmp_path, flags, conf):
    """HERest -u p...: MAPUpdateModels (HMap.c:413) from the pass's accumulators -- prior-weighted means, variances with the mean-shift
    term of HMap.c:350-356, weights from max(0, w*vSize*tau - 1) counts, HMap's own configuration (MAPTAU, MINVAR, MIXWEIGHTFLOOR) --
    against the MMF and the trace lines of the reference's HERest (tests/golden/make_map_golden.py)."""
    cf = tmp_path / "herest.conf"; cf.write_text("TARGETKIND = MFCC_E_D\n" + conf)
    out = tmp_path / "next"; out.mkdir()
    gold = os.path.join(DEMO, "hmm_map")
    src = os.path.join(DEMO, "hmm_tied" if flags.startswith("tied_") else "hmm_mixup", "newMacros")       # tied_: the set with ~u / ~v vectors
    r = run([os.path.join(tools, "herest"), "-T", "1", "-C", str(cf), "-u", flags.replace("tied_", ""), "-H", src, "-M", str(out),
             "-L", os.path.join(DEMO, "labels"), "-t", "2000.0", os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode == 0, r.stderr
    for line in open(os.path.join(gold, "herest_%s.log" % flags)).read().splitlines():
        assert line.strip() in r.stdout, (line, r.stdout[-600:])
    ours, theirs = _mmf_numbers(str(out / "newMacros")), _mmf_numbers(os.path.join(gold, "after_" + flags))
    _mmf_close(ours, theirs)


@pytest.mark.gpu
def test_herest_cli_map_refuses_transitions(tools, tmp_path):
    cf = tmp_path / "herest.conf"; cf.write_text("TARGETKIND = MFCC_E_D\n")
    out = tmp_path / "next"; out.mkdir()
    r = run([os.path.join(tools, "herest"), "-C", str(cf), "-u", "ptm", "-H", os.path.join(DEMO, "hmm_mixup", "newMacros"), "-M", str(out),
             "-L", os.path.join(DEMO, "labels"), os.path.join(DEMO, "bcplist")] + demo_train_files())
    assert r.returncode != 0 and "no MAP update of transition probabilities" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_hvite_cli_recognises_the_demo_sets(tools, tmp_path):
    expected = json.load(open(os.path.join(DEMO, "hvite_expected.json")))
    for part in ("test", "train"):
        names = sorted(expected[part])
        out = tmp_path / part; out.mkdir()
        conf = tmp_path / "hvite.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
        r = run([os.path.join(tools, "hvite"), "-C", str(conf), "-d", os.path.join(DEMO, "hmm_final"), "-w", os.path.join(DEMO, "monLattice"), "-l", str(out),
                 "-t", "300.0", "-p", "5.0", "-s", "0.0", os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] +
                [os.path.join(DEMO, part, u + ".mfc") for u in names])
        assert r.returncode == 0, r.stderr
        for u in names:
            assert (out / (u + ".rec")).read_text().splitlines() == expected[part][u], (part, u)


def _write_case_files(native, case, tmp_path):
    from htk_amd import synth
    src = os.path.join(GOLD, case)
    z = np.load(os.path.join(src, "feats.npz"))
    files = []
    for u in range(len(z.files)):
        fn = str(tmp_path / ("u%d.mfc" % u))
        synth.write_htk_param(fn, z["u%d" % u], kind=9)
        files.append(fn)
    return src, files


@pytest.mark.gpu
@pytest.mark.parametrize("fn", ["loop__-f_-m.mlf", "loop__-m_-o_N.mlf", "loop__-o_ST.mlf", "wint__-m_-o_SWX.mlf", "wint__-f_-m_-o_X.mlf", "loop__-f.mlf"])
def test_hvite_cli_output_formats_equal_the_reference_mlf(native, tools, tmp_path, fn):
    """-m / -f / -o through the CLI, written with -i: whole master label files equal to the reference HVite's, byte for byte."""
    case, optstr = fn[:-4].split("__")
    opts = optstr.replace("_", " ").split()
    src, files = _write_case_files(native, case, tmp_path)
    out = tmp_path / "out.mlf"
    r = run([os.path.join(tools, "hvite"), "-H", os.path.join(src, "MMF"), "-w", os.path.join(src, "net.slf"), "-i", str(out), "-l", "*", "-t", "250.0"] + opts +
            [os.path.join(src, "dict"), os.path.join(src, "hmmlist")] + files)
    assert r.returncode == 0, r.stderr
    assert out.read_text() == open(os.path.join(GOLD, "outfmt", fn)).read()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["net", "loop"])
def test_hvite_cli_cross_word_expansion(native, tools, tmp_path, which):
    """hvite -C config with FORCECXTEXP = T, ALLOWXWRDEXP = T: the network is expanded with cross-word contexts (ExpandWordNet HNet.c:3438,
    xc > 0); word-level and model-level (-m: triphones named by the neighbouring words) label files equal the reference HVite's."""
    from htk_amd import synth
    src = os.path.join(GOLD, "xwrd")
    z = np.load(os.path.join(src, "feats_%s.npz" % which))
    files = []
    for u in range(len(z.files)):
        fn = str(tmp_path / ("u%d.mfc" % u))
        synth.write_htk_param(fn, z["u%d" % u], kind=9)
        files.append(fn)
    expected = json.load(open(os.path.join(src, "expected_%s.json" % which)))
    for opts, per in expected.items():
        out = tmp_path / ("out_" + opts.replace(" ", "_")); out.mkdir()
        r = run([os.path.join(tools, "hvite"), "-C", os.path.join(src, "config"), "-H", os.path.join(src, "MMF"), "-w", os.path.join(src, which + ".slf"), "-l", str(out)] + opts.split() +
                [os.path.join(src, "dict"), os.path.join(src, "hmmlist")] + files)
        assert r.returncode == 0, r.stderr
        for u in range(len(files)):
            assert (out / ("u%d.rec" % u)).read_text().splitlines() == per["u%d" % u], (which, opts, u)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", json.load(open(os.path.join(GOLD, "nbest", "index.json"))))
def test_hvite_cli_nbest_and_lattices(native, tools, tmp_path, tag):
    """hvite -n i [N] [-z lat]: the lattice files and the N-best label files (alternatives separated by "///") of the reference's HVite,
    byte for byte (WriteLattice HNet.c:631, TranscriptionFromLattice HRec.c:2176).  Run inside a directory with the reference run's
    relative file names, so that the header lines of the lattices agree as well."""
    from htk_amd import synth
    meta = json.load(open(os.path.join(GOLD, "nbest", tag, "nbest.json")))
    src = os.path.join(GOLD, meta["case"])
    for fn in ("MMF", "dict", "hmmlist", meta["slf"] + ".slf"):
        os.symlink(os.path.join(src, fn), str(tmp_path / fn))
    cfg = []
    if os.path.exists(os.path.join(src, "config")):
        os.symlink(os.path.join(src, "config"), str(tmp_path / "config")); cfg = ["-C", "config"]
    (tmp_path / "nbtmp").mkdir()
    z = np.load(os.path.join(src, meta["feats"] + ".npz"))
    files = []
    for u in range(len(z.files)):
        synth.write_htk_param(str(tmp_path / "nbtmp" / ("u%d.mfc" % u)), z["u%d" % u], kind=9)
        files.append("nbtmp/u%d.mfc" % u)
    base = [os.path.join(tools, "hvite")] + cfg + ["-H", "MMF", "-w", meta["slf"] + ".slf"] + meta["opts"].split()
    r = run(base + ["-l", "nbtmp", "-n", str(meta["nToks"]), "1", "-z", "lat", "dict", "hmmlist"] + files, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    for u in range(len(files)):
        assert (tmp_path / "nbtmp" / ("u%d.lat" % u)).read_text() == open(os.path.join(GOLD, "nbest", tag, "u%d.lat" % u)).read(), (tag, u)
        assert (tmp_path / "nbtmp" / ("u%d.rec" % u)).read_text().splitlines() == meta["nbest"]["u%d" % u][0]
    r = run(base + ["-i", "nb.mlf", "-n", str(meta["nToks"]), str(meta["nTrans"]), "dict", "hmmlist"] + files, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    want = ["#!MLF!#"]
    for u in range(len(files)):
        want.append('"nbtmp/u%d.rec"' % u)
        want += "\n///\n".join("\n".join(a) for a in meta["nbest"]["u%d" % u]).split("\n") + ["."]
    assert (tmp_path / "nb.mlf").read_text().splitlines() == want


@pytest.mark.gpu
def test_hvite_cli_word_level_alignment(native, tools, tmp_path):
    """hvite -a [-b w] [-m] from word-level label files (DoAlignment HVite.c:830) against the reference's label files."""
    src, files = _write_case_files(native, "bigram", tmp_path)
    d = os.path.join(GOLD, "align")
    z = np.load(os.path.join(d, "feats.npz"))
    from htk_amd import synth
    files = []
    for u in range(len(z.files)):
        fn = str(tmp_path / ("a%d.mfc" % u)); synth.write_htk_param(fn, z["u%d" % u], kind=9); files.append(fn)
    words = json.load(open(os.path.join(d, "words.json")))
    for u, ws in enumerate(words):
        (tmp_path / ("a%d.lab" % u)).write_text("\n".join(ws) + "\n")
    expected = json.load(open(os.path.join(d, "expected.json")))
    for opts, per in expected.items():
        out = tmp_path / ("o" + str(abs(hash(opts)))); out.mkdir()
        r = run([os.path.join(tools, "hvite"), "-a", "-H", os.path.join(src, "MMF"), "-l", str(out)] + opts.split() + [os.path.join(src, "dict"), os.path.join(src, "hmmlist")] + files)
        assert r.returncode == 0, (opts, r.stderr)
        for u in range(len(words)):
            assert (out / ("a%d.rec" % u)).read_text().splitlines() == per["u%d" % u], (opts, u)


@pytest.mark.gpu
def test_hvite_cli_alignment_beam_retries_and_numeric_arguments(native, tools, tmp_path):
    """`-t f i l` (HVite.c:308-322): alignment retries a file with beams f, f+i, .. while no token reaches the final node (DoAlignment
    :900-913) -- against the reference's HVite run beside it; and the optional numbers of -t / -n are taken only when the WHOLE next argument
    is a number (NextArg() == INTARG / FLOATARG), so a dictionary called `3dict` stays a dictionary."""
    import shutil
    ref = os.path.join(ROOT, "oracle", "_ref", "HVite")
    conf = tmp_path / "hvite.conf"; conf.write_text("TARGETKIND = MFCC_E_D\n")
    files = demo_train_files()
    base = ["-C", str(conf), "-d", os.path.join(DEMO, "hmm_final"), "-a", "-m", "-L", os.path.join(DEMO, "labels")]
    for beam in (["-t", "15.0", "20.0", "200.0"], ["-t", "15.0", "20.0", "40.0"], ["-t", "150.0"]):
        ours = tmp_path / ("o" + "_".join(beam[1:])); ours.mkdir()
        r = run([os.path.join(tools, "hvite")] + base + beam + ["-l", str(ours), os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + files)
        assert r.returncode == 0, r.stderr
        if not os.path.exists(ref):
            continue
        theirs = tmp_path / ("r" + "_".join(beam[1:])); theirs.mkdir()
        rr = subprocess.run([ref] + base + beam + ["-l", str(theirs), os.path.join(DEMO, "bcpvocab"), os.path.join(DEMO, "bcplist")] + files, capture_output=True, text=True)
        assert rr.returncode == 0, rr.stdout + rr.stderr
        got, exp = sorted(os.listdir(str(ours))), sorted(os.listdir(str(theirs)))
        assert got == exp and (len(exp) == len(files) or beam[-1] == "40.0"), (beam, got, exp)
        for f in exp:
            assert (ours / f).read_text() == (theirs / f).read_text(), (beam, f)
    # a dictionary whose name starts with a digit, right after -n 2 and after -t 300.0
    shutil.copy(os.path.join(DEMO, "bcpvocab"), str(tmp_path / "3dict"))
    out = tmp_path / "n"; out.mkdir()
    r = run([os.path.join(tools, "hvite"), "-C", str(conf), "-d", os.path.join(DEMO, "hmm_final"), "-w", os.path.join(DEMO, "monLattice"), "-l", str(out), "-t", "300.0", "-n", "2",
             str(tmp_path / "3dict").replace(str(tmp_path), "."), os.path.join(DEMO, "bcplist")] + [os.path.join(DEMO, "test", "te1.mfc")], cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert (out / "te1.rec").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,src", [("WAV", "test.wav"), ("HTK", "test.htk")])
def test_cli_tools_code_waveform_sources_on_the_device(native, tools, tmp_path, fmt, src):
    """BASELINE config[4] through the drivers: SOURCEFORMAT = WAV (or SOURCEKIND = WAVEFORM on an HTK waveform file) + TARGETKIND =
    MFCC_0_D_A makes herest / hvite code the waveform on the device (htkamd_mfcc_compute + the qualifier step) instead of reading a
    parameter file.  One pass and one alignment over the golden waveform give the log probability / the label file that the same tools
    give on the MFCC_0_D_A file the reference's HCopy coded from it (tests/golden/wave; >99.9 % of its values bit-equal, rest 1 ulp)."""
    from htk_amd import synth
    W = os.path.join(os.path.dirname(__file__), "golden", "wave")
    s = synth.generate(12, 2, 4, 1, 98, 31, D=39)
    pk = s.packed()
    names = ["p%d" % i for i in range(pk["numPhys"])]
    synth.write_mmf_packed(str(tmp_path / "MMF"), pk, names, kind="MFCC_0_D_A")
    (tmp_path / "hmmlist").write_text("\n".join(names) + "\n")
    front = "SOURCERATE = 625\nWINDOWSIZE = 250000.0\nTARGETRATE = 100000.0\nNUMCHANS = 26\nNUMCEPS = 12\nCEPLIFTER = 22\nPREEMCOEF = 0.97\nUSEHAMMING = T\nENORMALISE = F\n"
    (tmp_path / "wav.conf").write_text("SOURCEFORMAT = %s\n%sTARGETKIND = MFCC_0_D_A\n" % (fmt, front) + ("SOURCEKIND = WAVEFORM\n" if fmt == "HTK" else ""))
    (tmp_path / "mfc.conf").write_text("TARGETKIND = MFCC_0_D_A\n")
    lab = "\n".join(["p0", "p1", "p2", "p3", "p1"]) + "\n"
    outs = {}
    for tag, conf, data in (("wav", "wav.conf", os.path.join(W, src)), ("mfc", "mfc.conf", os.path.join(W, "test_MFCC_0_D_A.mfc"))):
        d = tmp_path / tag; d.mkdir()
        base = os.path.splitext(os.path.basename(data))[0]
        (d / (base + ".lab")).write_text(lab)
        r = run([os.path.join(tools, "herest"), "-C", str(tmp_path / conf), "-H", str(tmp_path / "MMF"), "-M", str(d), "-L", str(d), "-m", "1", "-v", "0.01", str(tmp_path / "hmmlist"), data])
        assert r.returncode == 0, r.stdout + r.stderr
        lp = float(re.search(r"average log prob per frame = (\S+)", r.stdout).group(1))
        (tmp_path / (tag + ".dict")).write_text("".join("%s %s\n" % (n, n) for n in names))
        r2 = run([os.path.join(tools, "hvite"), "-C", str(tmp_path / conf), "-H", str(tmp_path / "MMF"), "-a", "-m", "-L", str(d), "-l", str(d), "-y", "rec",
                  str(tmp_path / (tag + ".dict")), str(tmp_path / "hmmlist"), data])
        assert r2.returncode == 0, r2.stdout + r2.stderr
        outs[tag] = (lp, (d / (base + ".rec")).read_text().split())
    assert abs(outs["wav"][0] - outs["mfc"][0]) <= 2e-6 * abs(outs["mfc"][0])
    a, b = outs["wav"][1], outs["mfc"][1]
    assert len(a) == len(b) and len(a) >= 15
    for x, y in zip(a, b):
        try:
            assert abs(float(x) - float(y)) <= 1e-3 * max(1.0, abs(float(y))), (x, y)
        except ValueError:
            assert x == y


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,src", [("WAV", "test.wav"), ("HTK", "test.htk")])
def test_wav_sources_align_and_decode_match_reference_hvite(native, tools, tmp_path, fmt, src):
    """north_star's "bit-exact Viterbi alignments" FROM A WAVEFORM SOURCE: tools/bin/hvite codes the waveform on the device (k_mfcc_*, the
    qualifier kernels) and aligns (-a -m -f: model and state level) / recognises over a word loop (-w) -- the label files must be the bytes
    the reference's HVite writes from the same waveform with its own HParm / HSigP front end.  Expected lines: committed
    (tests/golden/make_wav_labels_golden.py -> tests/golden/wave/expected_wav_labels.json); where oracle/_ref/HVite is on the box it is run
    beside as well.  The set is fitted to the file (tests/golden/wave/fitted.mmf), so the boundaries are decided by the data."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_wav_labels_golden as g
    exp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wave", "expected_wav_labels.json")))
    wav = os.path.join(os.path.dirname(__file__), "golden", "wave", src)
    d = str(tmp_path)
    g.write_case(d, fmt)
    ref = os.path.join(ROOT, "oracle", "_ref", "HVite")
    for mode in ("align", "loop"):
        ours = g.run_tool(os.path.join(tools, "hvite"), d, wav, mode)
        assert ours == exp["%s/%s" % (fmt, mode)], (fmt, mode, ours)
        if os.path.exists(ref):
            assert ours == g.run_tool(ref, d, wav, mode), (fmt, mode)
