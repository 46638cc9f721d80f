"""Network decoding on the device (K7, htkamd_decoder_*) against the reference's HVite label files (committed fixtures) and
against the oracle: word sequences, frame boundaries and printed scores identical."""
import os

import numpy as np
import pytest

from decode_util import format_words, load_decode_case, parse_opts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["loop", "bigram", "tee", "wint", "ties", "xwrd:net", "xwrd:loop"])
def test_decoder_reproduces_hvite_label_files(native, oracle, case):
    mmf, net, feats, expected = load_decode_case(native, case)
    model = native.Model(mmf.packed())
    om = oracle.Model(mmf.packed())
    arrays = net.arrays()
    for opts, per in expected.items():
        if "-m" in opts.split():
            continue                                                 # model-level output: its own test below
        p = parse_opts(opts)
        dec = native.Decoder(model, net, lmScale=p["lmScale"])
        res = dec.run(feats, **p)
        for u, (words, total) in enumerate(res):
            if "u%d" % u not in per:                                 # "No tokens survived": HVite writes no entry for the file
                assert words is None and oracle.decode(om, feats[u], arrays, **p)[0] is None, (case, opts, u)
                continue
            assert words is not None, (case, opts, u)
            assert format_words(words, net.out_syms) == per["u%d" % u], (case, opts, u)
            ow, ot = oracle.decode(om, feats[u], arrays, **p)
            assert words == ow and total == ot                      # token likelihoods are the same doubles


def test_decoder_larger_loop_matches_oracle(native, oracle, tmp_path):
    """400 single-model words over 150 tied states x 4 mixtures (fan-in of the loop node > one wave), 6 ragged utterances, beam 120."""
    from htk_amd import synth
    s = synth.generate(150, 4, 400, 6, 90, 17, D=13)
    d = tmp_path
    synth.write_mmf(str(d / "MMF"), s, kind="USER")
    names = ["p%d" % i for i in range(400)]
    (d / "hmmlist").write_text("\n".join(names) + "\n")
    (d / "dict").write_text("".join("%s %s\n" % (n, n) for n in names))
    V = len(names)
    with open(d / "net.slf", "w") as f:                             # the shape HBuild gives a word loop
        f.write("VERSION=1.0\nN=%d L=%d\n" % (V + 4, 2 * V + 3))
        f.write("I=0 W=!NULL\nI=1 W=!NULL\n")
        for i, n in enumerate(names):
            f.write("I=%d W=%s\n" % (2 + i, n))
        f.write("I=%d W=!NULL\nI=%d W=!NULL\n" % (V + 2, V + 3))
        j = 0
        f.write("J=%d S=0 E=1 l=0.00\n" % j); j += 1
        f.write("J=%d S=%d E=1 l=0.00\n" % (j, V + 2)); j += 1
        for i in range(V):
            f.write("J=%d S=1 E=%d l=%.2f\n" % (j, 2 + i, np.log(1.0 / V))); j += 1
            f.write("J=%d S=%d E=%d l=0.00\n" % (j, 2 + i, V + 2)); j += 1
        f.write("J=%d S=%d E=%d l=0.00\n" % (j, V + 2, V + 3))
    mmf = native.Mmf(files=[str(d / "MMF")], hmm_list=str(d / "hmmlist"))
    net = native.Net(str(d / "net.slf"), str(d / "dict"), mmf)
    feats = [f[: 90 - 7 * u] for u, f in enumerate(s.feats)]
    model = native.Model(mmf.packed()); om = oracle.Model(mmf.packed())
    res = native.Decoder(model, net).run(feats, genBeam=120.0)
    for u, (words, total) in enumerate(res):
        ow, ot = oracle.decode(om, feats[u], net.arrays(), genBeam=120.0)
        assert words == ow and total == ot and len(words) >= 3
    # an impossible utterance (shorter than any path through the network) reports "no token survived"
    short = native.Decoder(model, net).run([s.feats[0][:2]], genBeam=120.0)
    assert short[0][0] is None
    # one decoder over batches of different sizes: its workspace is kept between calls and grown when a batch needs more
    dec = native.Decoder(model, net)
    for sel in ([0], [0, 1, 2, 3, 4, 5], [2], [5, 4, 3]):
        got = dec.run([feats[u] for u in sel], genBeam=120.0)
        assert [(w, t) for w, t in got] == [res[u] for u in sel], sel


def test_decoder_config3_size_matches_reference(native, tmp_path):
    """BASELINE config[3] at full size: 5000 tied states x 16 mixtures, 6000-word loop, 500-frame utterances, -t 250.
    The expected label lines were written by the reference's HVite for the same (seeded) set:
    tests/golden/decode/config3/expected.json; the set itself is regenerated here from its seed."""
    import json
    import os
    from htk_amd import synth
    exp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "decode", "config3", "expected.json")))["-t 250.0"]
    s = synth.generate(5000, 16, 6000, 2, 500, 3)
    V = 6000
    names = ["p%d" % i for i in range(V)]
    d = tmp_path
    synth.write_mmf(str(d / "MMF"), s)
    (d / "hmmlist").write_text("\n".join(names) + "\n")
    (d / "dict").write_text("".join("%s %s\n" % (n, n) for n in names))
    with open(d / "net.slf", "w") as f:                             # HBuild's word loop: l = log(1/V) printed with two decimals
        f.write("VERSION=1.0\nN=%d L=%d\nI=0 W=!NULL\nI=1 W=!NULL\n" % (V + 4, 2 * V + 3))
        for i, n in enumerate(names):
            f.write("I=%d W=%s\n" % (2 + i, n))
        f.write("I=%d W=!NULL\nI=%d W=!NULL\n" % (V + 2, V + 3))
        j = 0
        f.write("J=%d S=0 E=1 l=0.00\n" % j); j += 1
        f.write("J=%d S=%d E=1 l=0.00\n" % (j, V + 2)); j += 1
        for i in range(V):
            f.write("J=%d S=1 E=%d l=%.2f\n" % (j, 2 + i, np.log(1.0 / V))); j += 1
            f.write("J=%d S=%d E=%d l=0.00\n" % (j, 2 + i, V + 2)); j += 1
        f.write("J=%d S=%d E=%d l=0.00\n" % (j, V + 2, V + 3))
    mmf = native.Mmf(files=[str(d / "MMF")], hmm_list=str(d / "hmmlist"))
    net = native.Net(str(d / "net.slf"), str(d / "dict"), mmf)
    model = native.Model(mmf.packed())
    res = native.Decoder(model, net).run(s.feats, genBeam=250.0)
    for u, (words, total) in enumerate(res):
        assert format_words(words, net.out_syms) == exp["u%05d" % u], u


def test_decoder_config3_size_bigram_network_matches_reference(native, tmp_path):
    """BASELINE config[3] as it is worded -- 5000 tied states x 16 mixtures and a BIGRAM network: 6000 words, five explicit successors
    per word plus a back-off null node that reaches every word (18 000 word/model nodes, 42 000 arcs, fan-in 6000 at the back-off
    node), -t 250 -s 5 -p -10.  Expected label lines: the reference's HVite on the same seeded set
    (tests/golden/make_config3_golden.py -> tests/golden/decode/config3/expected_bigram.json)."""
    import json
    import sys
    from htk_amd import synth
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_config3_golden import V, write_bigram
    exp = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "decode", "config3", "expected_bigram.json")))["-t 250.0 -s 5.0 -p -10.0"]
    s = synth.generate(5000, 16, V, 2, 500, 3)
    names = ["p%d" % i for i in range(V)]
    d = tmp_path
    synth.write_mmf(str(d / "MMF"), s)
    (d / "hmmlist").write_text("\n".join(names) + "\n")
    (d / "dict").write_text("".join("%s %s\n" % (n, n) for n in names))
    write_bigram(str(d / "net.slf"), names)
    mmf = native.Mmf(files=[str(d / "MMF")], hmm_list=str(d / "hmmlist"))
    net = native.Net(str(d / "net.slf"), str(d / "dict"), mmf)
    model = native.Model(mmf.packed())
    res = native.Decoder(model, net, lmScale=5.0).run(s.feats, genBeam=250.0, lmScale=5.0, wordPen=-10.0)
    for u, (words, total) in enumerate(res):
        assert format_words(words, net.out_syms) == exp["u%05d" % u], u


def test_decoder_edge_cases(native, oracle):
    """Empty batch, an utterance with no frames, a word budget that is too small, and a beam so tight that the path dies:
    the statuses follow CompleteRecognition (no token in the final node -> nothing to output)."""
    mmf, net, feats, expected = load_decode_case(native, "bigram")
    model = native.Model(mmf.packed())
    dec = native.Decoder(model, net)
    assert dec.run([]) == []
    res = dec.run([feats[0], feats[0][:0], feats[1]], genBeam=250.0)
    assert res[0][0] is not None and res[2][0] is not None and res[1][0] is None
    om = oracle.Model(mmf.packed())
    assert res[0][0] == oracle.decode(om, feats[0], net.arrays(), genBeam=250.0)[0]       # neighbours of the empty one unaffected
    with pytest.raises(native.HtkAmdError):
        dec.run([feats[0]], lmScale=3.0)                             # LM scale is part of the decoder (LikeToWord look-ahead)
    lib = native.lib()
    import ctypes as C
    X = np.ascontiguousarray(feats[0], np.float32)
    dX = native.DevArray(X)
    frameOff = np.array([0, X.shape[0]], np.int32)
    nW = np.zeros(1, np.int32); tot = np.zeros(1, np.float64)
    wp = np.zeros(2, np.int32); ws = np.zeros(2, np.int32); we = np.zeros(2, np.int32); sc = np.zeros(2, np.float32)
    cfg = native.DecodeConfig(250.0, 1.0e10, 1.0, 0.0, 1.0, 0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    native.check(lib.htkamd_decoder_run(dec.h, C.byref(cfg), dX.ptr, p(frameOff), C.c_int(1), C.c_int(2), p(nW), p(wp), p(ws), p(we), p(sc), None, p(tot), None), "run")
    assert nW[0] == -3 and tot[0] > -1e9                              # five words do not fit into maxWords = 2
    tight = dec.run([feats[0]], genBeam=0.5)
    ow, _ = oracle.decode(om, feats[0], net.arrays(), genBeam=0.5)
    assert tight[0][0] == ow                                          # whatever the reference semantics give (None or a path)


@pytest.mark.parametrize("case", ["loop", "bigram", "wint", "xwrd:net", "xwrd:loop"])
def test_model_level_labels_of_recognition(native, case):
    """HVite -m with -w: model-level labels of the recognised words = decoder (words, LM scores) + forced alignment of the
    recognised pronunciations' model chain; compared line by line with the reference's output."""
    from decode_util import format_model_labels
    from util import batch_arrays
    mmf, net, feats, expected = load_decode_case(native, case)
    model = native.Model(mmf.packed())
    n = 0
    for opts, per in expected.items():
        if "-m" not in opts.split():
            continue
        p = parse_opts(opts)
        dec = native.Decoder(model, net, lmScale=p["lmScale"])
        res = dec.run(feats, **p)
        seqm = [net.seq_models([w[0] for w in words]) if net.xwrd else [net.pron_models[w[0]] for w in words] for words, _ in res]
        chains = [np.array([m for wm in sm for m in wm], np.int32) for sm in seqm]
        X, frameOff, labOff, labs = batch_arrays([dict(seq=c, feat=f) for c, f in zip(chains, feats)])
        dX = native.DevArray(X)
        al = native.Viterbi(model).align(dX.ptr.value, frameOff, labOff, labs, genBeam=p["genBeam"])
        for u, (words, total) in enumerate(res):
            got = format_model_labels(words, dec.last_lm[u], al[u], net.pron_models, mmf.phys_names, net.word_names, p["lmScale"], p["wordPen"], seq_models=seqm[u])
            assert got == per["u%d" % u], (case, opts, u)
            n += len(got)
    assert n > 10


def test_decoder_with_matrix_core_scores(native):
    """scoreMode = MFMA in the decoder: same words and boundaries as the exact mode on the golden cases, scores within 1e-2
    (sums of ~100 frame scores each good to ~1e-4)."""
    for case in ("loop", "bigram"):
        mmf, net, feats, expected = load_decode_case(native, case)
        model = native.Model(mmf.packed())
        dec = native.Decoder(model, net)
        ex = dec.run(feats, genBeam=250.0)
        for mode in (1, 4):                                  # fp32 matrix-core products; bf16 x 3 (the mode bench.py reports beside the exact decoding leg)
            if mode == 4 and int(mmf.packed()["vecSize"]) > 45:
                continue
            mf = dec.run(feats, genBeam=250.0, scoreMode=mode)
            for (we, te), (wm, tm) in zip(ex, mf):
                assert [w[:3] for w in we] == [w[:3] for w in wm], mode
                assert np.allclose([w[3] for w in we], [w[3] for w in wm], atol=1e-2) and abs(te - tm) < 5e-2, mode


def test_word_level_forced_alignment_with_pronunciation_variants(native):
    """HVite -a from word-level transcriptions (DoAlignment HVite.c:830): the transcription becomes a linear lattice
    (LatticeFromLabels, with the -b boundary word when given), every pronunciation of a word is a parallel branch, token passing
    picks the variant; word-level and model-level (-m) label files against the reference's (make_align_golden.py)."""
    import json
    from decode_util import GOLD, format_model_labels, format_words, parse_opts
    from util import batch_arrays
    src, d = os.path.join(GOLD, "bigram"), os.path.join(GOLD, "align")
    mmf = native.Mmf(files=[os.path.join(src, "MMF")], hmm_list=os.path.join(src, "hmmlist"))
    model = native.Model(mmf.packed())
    z = np.load(os.path.join(d, "feats.npz"))
    feats = [z["u%d" % u] for u in range(len(z.files))]
    trans = json.load(open(os.path.join(d, "words.json")))
    expected = json.load(open(os.path.join(d, "expected.json")))
    n = variants = 0
    for opts, per in expected.items():
        p = parse_opts(opts)
        t = opts.split()
        bnd = t[t.index("-b") + 1] if "-b" in t else None
        for u, (X, words) in enumerate(zip(feats, trans)):
            net = native.Net(None, os.path.join(src, "dict"), mmf, words=words, boundary=bnd)
            dec = native.Decoder(model, net, lmScale=p["lmScale"])
            (got_words, total), = dec.run([X], **p)
            assert [net.word_names[w[0]] for w in got_words] == ([bnd] if bnd else []) + words + ([bnd] if bnd else [])
            variants += sum(1 for w in got_words if net.pron_models[w[0]] == [0, 2])            # AB's second pronunciation
            if "-m" in t:
                chain = np.array([m for w in got_words for m in net.pron_models[w[0]]], np.int32)
                Xb, frameOff, labOff, labs = batch_arrays([dict(seq=chain, feat=X)])
                dX = native.DevArray(Xb)
                al = native.Viterbi(model).align(dX.ptr.value, frameOff, labOff, labs, genBeam=p["genBeam"])
                got = format_model_labels(got_words, dec.last_lm[0], al[0], net.pron_models, mmf.phys_names, net.word_names, p["lmScale"], p["wordPen"])
            else:
                got = format_words(got_words, net.out_syms)
            assert got == per["u%d" % u], (opts, u)
            n += len(got)
    assert n > 50 and variants > 0
    with pytest.raises(native.HtkAmdError):
        native.Net(None, os.path.join(src, "dict"), mmf, words=["AB", "NOSUCHWORD"])


def test_output_formatting_matches_hvite_mlf(native, tmp_path):
    """HVite's -o flags (FormatTranscription), the -m / -f auxiliary labels and the master label file it writes with -i, through
    the C label writer (htkamd_trans_* / htkamd_mlf_out_*): whole MLFs equal to the reference's, byte for byte."""
    from decode_util import GOLD
    from util import batch_arrays
    gold = os.path.join(GOLD, "outfmt")
    n = 0
    for fn in sorted(os.listdir(gold)):
        case, optstr = fn[:-4].split("__")
        opts = optstr.replace("_", " ").split()
        models, states = "-m" in opts, "-f" in opts
        flags = opts[opts.index("-o") + 1] if "-o" in opts else ""
        mmf, net, feats, _ = load_decode_case(native, case)
        model = native.Model(mmf.packed())
        dec = native.Decoder(model, net, lmScale=1.0)
        res = dec.run(feats, genBeam=250.0)
        al = None
        if models or states:
            chains = [np.array([m for w in words for m in net.pron_models[w[0]]], np.int32) for words, _ in res]
            X, frameOff, labOff, labs = batch_arrays([dict(seq=c, feat=f) for c, f in zip(chains, feats)])
            dX = native.DevArray(X)
            al = native.Viterbi(model).align(dX.ptr.value, frameOff, labOff, labs, genBeam=250.0)
        out = native.MlfOut(str(tmp_path / fn))
        for u, (words, total) in enumerate(res):
            t = native.Trans((1 if models else 0) + (1 if states else 0) if (models or states) else 0)
            q = k0 = 0                                            # model index / first state slot of that model in the alignment
            for (w, s, e, sc), lm in zip(words, dec.last_lm[u]):
                aux = float(np.float32(np.float64(np.float32(np.float32(lm) * np.float32(1.0))) + np.float64(np.float32(0.0))))
                if not (models or states):
                    if net.out_syms[w] != "":
                        t.add(s * 100000.0, e * 100000.0, net.out_syms[w], float(np.float32(sc)))
                    continue
                for k, m in enumerate(net.pron_models[w]):
                    word = net.word_names[w] if k == 0 else None
                    name = mmf.phys_names[m]
                    if not states:
                        t.add(al[u]["modStart"][q] * 100000.0, al[u]["modEnd"][q] * 100000.0, name, float(np.float32(al[u]["modScore"][q])),
                              aux1=word, aux1_score=aux if word else 0.0)
                    else:
                        first = True
                        for j in range(al[u]["nStates"][q]):
                            st, en = int(al[u]["segStart"][k0 + j]), int(al[u]["segEnd"][k0 + j])
                            if st < 0:
                                continue
                            ssc = float(np.float32(al[u]["segScore"][k0 + j]))
                            if models:
                                t.add(st * 100000.0, en * 100000.0, "s%d" % (j + 2), ssc, aux1=name if first else None,
                                      aux1_score=float(np.float32(al[u]["modScore"][q])) if first else 0.0,
                                      aux2=word if first else None, aux2_score=aux if (first and word) else 0.0)
                            else:
                                t.add(st * 100000.0, en * 100000.0, "%s[%d]" % (name, j + 2), ssc, aux1=word if first else None,
                                      aux1_score=aux if (first and word) else 0.0)
                            first = False
                    k0 += al[u]["nStates"][q]
                    q += 1
            t.format(100000.0, states=states, models=models, flags=flags)
            out.add("*/u%d.rec" % u, t)
        out.close()
        assert (tmp_path / fn).read_text() == open(os.path.join(gold, fn)).read(), fn
        n += 1
    assert n == 16


# ----------------------------------------------------------------------------------------- HRec's dynamic instance order
TIES2 = ["ties2/tie_122", "ties2/tie_220"]


@pytest.mark.parametrize("case", TIES2)
def test_decoder_equals_oracle_on_manufactured_exact_ties(native, oracle, case):
    mmf, net, feats, expected = load_decode_case(native, case)
    model, om, arrays = native.Model(mmf.packed()), oracle.Model(mmf.packed()), net.arrays()
    for opts, per in expected.items():
        p = parse_opts(opts)
        res = native.Decoder(model, net, lmScale=p["lmScale"]).run(feats, **p)
        for u, (words, total) in enumerate(res):
            ow, ot = oracle.decode(om, feats[u], arrays, **p)
            assert words == ow and total == ot


@pytest.mark.parametrize("case", TIES2)
def test_decoder_exact_ties_follow_hrec_instance_order(native, case):
    """Exact ties between homophones are broken by HRec's instance-list order (AttachInst / MoveToRecent / ReOrderList, HRec.c:1123-1301).
    Two manufactured files (tests/fuzz_ties_vs_ref.py: 2 of 745) on which the batch kernel's static pull order keeps the other of two
    equally scored words: the kernel notes the tie and the utterance is decoded again in the list's order (decode_ord.hip) -- HVite's
    labels.  The static order alone (ORDER_FAST) is what differs, in that one word."""
    mmf, net, feats, expected = load_decode_case(native, case)
    model = native.Model(mmf.packed())
    nfast = 0
    for opts, per in expected.items():
        p = parse_opts(opts)
        dec = native.Decoder(model, net, lmScale=p["lmScale"])
        res = dec.run(feats, **p)
        assert dec.last_tied() >= 1
        for u, (words, total) in enumerate(res):
            assert format_words(words, net.out_syms) == per["u%d" % u], (case, opts, u)
        dec.set_order(native.ORDER_EXACT)
        res2 = dec.run(feats, **p)
        assert dec.last_tied() == len(feats) and res2 == res
        dec.set_order(native.ORDER_FAST)
        res3 = dec.run(feats, **p)
        assert dec.last_tied() == 0
        for (w3, t3), (w1, t1) in zip(res3, res):
            assert t3 == t1 and len(w3) == len(w1)
            nfast += sum(1 for x, y in zip(w3, w1) if x != y)
    assert 1 <= nfast <= 2


@pytest.mark.parametrize("case", ["loop", "bigram", "tee", "wint", "ties", "xwrd:net", "xwrd:loop"])
def test_list_order_kernel_reproduces_hvite_label_files(native, case):
    """Every utterance through the list kernel (ORDER_EXACT): the label files of the reference's HVite, as from the batch kernel."""
    mmf, net, feats, expected = load_decode_case(native, case)
    model = native.Model(mmf.packed())
    for opts, per in expected.items():
        if "-m" in opts.split():
            continue
        p = parse_opts(opts)
        dec = native.Decoder(model, net, lmScale=p["lmScale"])
        dec.set_order(native.ORDER_EXACT)
        res = dec.run(feats, **p)
        assert dec.last_tied() == len(feats)
        for u, (words, total) in enumerate(res):
            want = per.get("u%d" % u)
            got = None if words is None else format_words(words, net.out_syms)
            assert got == want, (case, opts, u)


def test_list_walk_out_of_capacity_keeps_the_batch_kernels_answer(native, tmp_path):
    """in the default mode an utterance with an exact tie is decoded again by the list kernel; when that walk runs out of one
    of its fixed capacities (here: a chain of 70 !NULL nodes in front of the final node, deeper than the 64 zero-time nodes ReOrderList's
    explicit stack holds: status -6) the batch kernel's valid result stays instead of turning into a failure."""
    import json
    from decode_util import GOLD
    d = os.path.join(GOLD, "ties2", "tie_122")
    lines = open(os.path.join(d, "net.slf")).read().splitlines()
    nodes = [l for l in lines if l.startswith("I=")]
    arcs = [l for l in lines if l.startswith("J=")]
    n0 = len(nodes)
    final = n0 - 1                                           # the lattice's last node is its end
    # the old end node stays (a !NULL), 70 more follow it in a row
    extra = ["I=%d W=!NULL" % (n0 + k) for k in range(70)]
    xarcs = ["J=%d S=%d E=%d l=0.00" % (len(arcs) + k, final if k == 0 else n0 + k - 1, n0 + k) for k in range(70)]
    slf = tmp_path / "net.slf"
    slf.write_text("VERSION=1.0\nN=%d L=%d\n" % (n0 + 70, len(arcs) + 70) + "\n".join(nodes + extra + arcs + xarcs) + "\n")
    mmf = native.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    net = native.Net(str(slf), os.path.join(d, "dict"), mmf)
    z = np.load(os.path.join(d, "feats.npz"))
    feats = [z["u%d" % u] for u in range(len(z.files))]
    expected = json.load(open(os.path.join(d, "expected.json")))
    p = parse_opts(next(iter(expected)))
    model = native.Model(mmf.packed())
    dec = native.Decoder(model, net, lmScale=p["lmScale"])
    dec.set_order(native.ORDER_FAST)
    fast = dec.run(feats, **p)
    assert all(w is not None and len(w) > 0 for w, _ in fast)
    dec.set_order(native.ORDER_AUTO)
    auto = dec.run(feats, **p)
    assert dec.last_tied() >= 1                               # the tie was seen and the list kernel tried
    assert auto == fast


@pytest.mark.parametrize("case", ["loop", "bigram", "tee", "wint", "ties", "xwrd:net"])
def test_token_kernel_variants_agree(native, case, monkeypatch):
    """The register kernel with the word ends' tokens in LDS (the default where they fit), the same with every exit token in memory
    (HTKAMD_DECODE_NOEXL) and the kernel with no tokens in registers at all (HTKAMD_DECODE_NOREG, read when the decoder is created):
    the same words, boundaries, scores and total likelihoods to the last bit, with and without -u (maxActive)."""
    mmf, net, feats, expected = load_decode_case(native, case)
    model = native.Model(mmf.packed())
    for opts in list(expected)[:2]:
        if "-m" in opts.split():
            continue
        p = parse_opts(opts)
        for extra in ({}, {"maxActive": 12}):
            kw = dict(p, **extra)
            monkeypatch.delenv("HTKAMD_DECODE_NOEXL", raising=False); monkeypatch.delenv("HTKAMD_DECODE_NOREG", raising=False)
            ref = native.Decoder(model, net, lmScale=p["lmScale"]).run(feats, **kw)
            monkeypatch.setenv("HTKAMD_DECODE_NOEXL", "1")
            assert native.Decoder(model, net, lmScale=p["lmScale"]).run(feats, **kw) == ref, (case, opts, extra, "NOEXL")
            monkeypatch.setenv("HTKAMD_DECODE_NOREG", "1")
            assert native.Decoder(model, net, lmScale=p["lmScale"]).run(feats, **kw) == ref, (case, opts, extra, "NOREG")
