"""The decoding oracle (oracle/orc_decode.c) and the network builder (htk_amd/host/net.c) against the reference's HVite:
label files (times, output symbols, scores as printed) identical for word loops, a back-off bigram lattice with
pronunciation variants/probabilities/output symbols, tee models, and several -t/-v/-s/-p/-r settings.
Fixtures: tests/golden/make_decode_golden.py."""
import os

import numpy as np
import pytest

from decode_util import GOLD, format_words, load_decode_case, parse_opts


@pytest.mark.parametrize("case", ["loop", "bigram", "tee", "wint", "ties", "xwrd:net", "xwrd:loop"])
def test_oracle_decode_reproduces_hvite(native, oracle, case):
    mmf, net, feats, expected = load_decode_case(native, case)
    om = oracle.Model(mmf.packed())
    arrays = net.arrays()
    n = 0
    for opts, per in expected.items():
        if "-m" in opts.split():
            continue
        for u, X in enumerate(feats):
            words, total = oracle.decode(om, X, arrays, **parse_opts(opts))
            if "u%d" % u not in per:                                 # "No tokens survived": HVite writes no entry for the file
                assert words is None, (case, opts, u)
                continue
            assert words is not None
            assert format_words(words, net.out_syms) == per["u%d" % u], (case, opts, u)
            n += len(words)
    assert n > 30


def test_network_shapes(native):
    mmf, net, feats, expected = load_decode_case(native, "loop")
    a = net.arrays()
    # HBuild word loop over V = 12 words: 4 null lattice nodes + initial/final, V model nodes + V word-end nodes
    assert (a["kind"] == 0).sum() == 12 and (a["kind"] == 1).sum() == 12 and (a["kind"] == 2).sum() == 6
    assert a["initial"] == 0 and a["final"] == 1 and len(a["linkDest"]) == a["linkOff"][-1]
    lm = a["linkLike"][a["linkLike"] != 0]
    assert len(lm) == 12 and np.allclose(lm, np.log(1.0 / 12), atol=5e-3)            # HBuild prints l= with two decimals
    mmf2, net2, _, _ = load_decode_case(native, "bigram")
    b = net2.arrays()
    assert net2.out_syms == ["AB", "AB", "CD", "eee", "", "G", "H", "I"]
    assert (b["kind"] == 1).sum() == 8                                               # AB has two pronunciations
    assert np.isclose(b["pronProb"][b["kind"] == 1].min(), np.log(0.3), atol=1e-6)


def test_network_errors(native, tmp_path):
    d = os.path.join(GOLD, "loop")
    mmf = native.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    (tmp_path / "dict").write_text("p0 p0\np1 nosuchphone\n")
    with pytest.raises(native.HtkAmdError):
        native.Net(os.path.join(d, "net.slf"), str(tmp_path / "dict"), mmf)
    (tmp_path / "dict2").write_text("p0 p0\n")                                       # word of the lattice missing
    with pytest.raises(native.HtkAmdError):
        native.Net(os.path.join(d, "net.slf"), str(tmp_path / "dict2"), mmf)
    (tmp_path / "bad.slf").write_text("VERSION=1.0\nN=2 L=1\nI=0 W=!NULL\nI=1 W=!NULL\nJ=0 S=0 E=5\n")
    with pytest.raises(native.HtkAmdError):
        native.Net(str(tmp_path / "bad.slf"), os.path.join(d, "dict"), mmf)


# ----------------------------------------------------------------------------------------- known gap: HRec's dynamic instance order
TIES2 = ["ties2/tie_122", "ties2/tie_220"]


@pytest.mark.parametrize("case", TIES2)
def test_oracle_decode_exact_ties_follow_hrec_instance_order(native, oracle, case):
    """HRec keeps its instances in a list that is re-ordered as they are activated (AttachInst / MoveToRecent / ReOrderList,
    HRec.c:1123-1270): which of two EXACTLY tied tokens reaches a node first depends on that history.  Two manufactured files
    (tests/fuzz_ties_vs_ref.py, seed 5) on which a static pull order picks the other, equally scored homophone for one word: the oracle
    walks the reference's list (oracle/orc_ilist.h) and writes HVite's labels."""
    mmf, net, feats, expected = load_decode_case(native, case)
    om = oracle.Model(mmf.packed())
    arrays = net.arrays()
    for opts, per in expected.items():
        for u, X in enumerate(feats):
            words, total = oracle.decode(om, X, arrays, **parse_opts(opts))
            assert format_words(words, net.out_syms) == per["u%d" % u], (case, opts, u)
