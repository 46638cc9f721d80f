"""N-best recognition and lattice output (HVite -n N [M] -z lat) against the reference's files (tests/golden/make_nbest_golden.py):
CPU   the oracle's token-set decoder (oracle/orc_decode_n.c) through the product's host code -- htkamd_lattice_write must give the SLF
      file HVite wrote, byte for byte, and htkamd_lattice_nbest its alternative transcriptions, line for line;
GPU   the token-set kernel (htk_amd/csrc/decode_n.hip) must give the oracle's lattice: same nodes, same arcs, same float scores."""
import json
import os

import numpy as np
import pytest

from decode_util import GOLD, load_decode_case, parse_opts

NB = os.path.join(GOLD, "nbest")
TAGS = json.load(open(os.path.join(NB, "index.json")))


def _case(native, tag):
    meta = json.load(open(os.path.join(NB, tag, "nbest.json")))
    name = meta["case"] + (":" + meta["slf"] if meta["slf"] != "net" else "")
    mmf, net, feats, _ = load_decode_case(native, name)
    t = meta["opts"].split()
    meta["align"] = (1 if "-m" in t else 0) | (2 if "-f" in t else 0)      # -m / -f with -n: alignment records inside the arcs (HRec.c:1582-1656)
    p = parse_opts(" ".join(x for x in t if x not in ("-m", "-f")))
    return meta, mmf, net, feats, p


def _with_prons(lat, net):
    a = net.arrays()
    lat = dict(lat)
    lat["nodePron"] = np.array([a["model"][n] if n >= 0 else -1 for n in lat["nodeNet"]], np.int32)
    return lat


def _labels(alt, net, frame_dur=100000):
    return ["%d %d %s %f" % (s * frame_dur, e * frame_dur, net.out_syms[w], np.float32(sc)) for w, s, e, sc in alt if w >= 0 and net.out_syms[w] != ""]


def _check_files(native, lat, net, meta, tag, u, tmp_path, mmf=None):
    out = str(tmp_path / ("u%d.lat" % u))
    native.lattice_write(lat, net, out, utterance="nbtmp/u%d.mfc" % u, lm_name=meta["slf"] + ".slf", vocab_name="dict", mmf=mmf if meta["align"] else None)
    assert open(out).read() == open(os.path.join(NB, tag, "u%d.lat" % u)).read(), (tag, u)
    if meta["align"]:
        # model / state level alternatives: TranscriptionFromLattice over the arcs' records, HVite's -o formatting, the label file's lines
        alts = native.lattice_nbest_align(lat, net, mmf, meta["nTrans"], states=bool(meta["align"] & 2), models=bool(meta["align"] & 1))
        assert alts == meta["nbest"]["u%d" % u], (tag, u)
        return
    alts = native.lattice_nbest(lat, net, meta["nTrans"])
    assert [_labels(a, net) for a in alts] == meta["nbest"]["u%d" % u], (tag, u)


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_token_sets_and_host_lattice_code_equal_hvite(native, oracle, tag, tmp_path):
    meta, mmf, net, feats, p = _case(native, tag)
    om = oracle.Model(mmf.packed())
    n = 0
    for u, X in enumerate(feats):
        lat = oracle.decode_nbest(om, X, net.arrays(), meta["nToks"], align=meta["align"], **p)
        assert lat is not None
        lat = _with_prons(lat, net)
        lat.update(lmScale=p["lmScale"], wordPen=p["wordPen"], prScale=p["prScale"])
        if meta["align"]:
            lat["alModel"] = np.array([net.arrays()["model"][n] for n in lat["alNode"]], np.int32); lat["alignModels"] = bool(meta["align"] & 1)
        _check_files(native, lat, net, meta, tag, u, tmp_path, mmf)
        n += len(lat["arcStart"])
    assert n > 20


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_token_set_kernel_equals_oracle_and_hvite(native, oracle, tag, tmp_path):
    meta, mmf, net, feats, p = _case(native, tag)
    model = native.Model(mmf.packed())
    om = oracle.Model(mmf.packed())
    dec = native.Decoder(model, net, lmScale=p["lmScale"])
    lats = dec.run_lattice(feats, meta["nToks"], align=meta["align"], **p)
    for u, X in enumerate(feats):
        ref = oracle.decode_nbest(om, X, net.arrays(), meta["nToks"], align=meta["align"], **p)
        got = lats[u]
        assert got is not None and ref is not None
        assert got["total"] == ref["total"]
        for k in ("nodeFrame", "nodeNet", "nodeLike"):
            assert np.array_equal(got[k], ref[k]), (tag, u, k)
        arcs = lambda l: sorted(zip(l["arcStart"].tolist(), l["arcEnd"].tolist(), l["arcAc"].tolist(), l["arcLm"].tolist(), l["arcPr"].tolist(), l["arcScore"].tolist()))
        assert arcs(got) == arcs(ref), (tag, u)                     # the order in which the arcs are listed is not part of the lattice
        if meta["align"]:                                           # every arc's records: the oracle's, value for value
            am = net.arrays()["model"]
            def recs(l, model_of):
                out = []
                for j in range(len(l["arcStart"])):
                    a0, a1 = int(l["arcAlignOff"][j]), int(l["arcAlignOff"][j + 1])
                    out.append((int(l["arcStart"][j]), int(l["arcEnd"][j]), float(l["arcScore"][j]),
                                tuple((int(l["alState"][q]), int(model_of(l, q)), int(l["alDur"][q]), float(l["alLike"][q])) for q in range(a0, a1))))
                return sorted(out)
            assert recs(got, lambda l, q: l["alModel"][q]) == recs(ref, lambda l, q: am[l["alNode"][q]]), (tag, u)
        _check_files(native, got, net, meta, tag, u, tmp_path, mmf)


@pytest.mark.gpu
def test_token_set_kernel_edge_cases(native):
    mmf, net, feats, _ = load_decode_case(native, "bigram")
    model = native.Model(mmf.packed())
    dec = native.Decoder(model, net)
    assert dec.run_lattice([], 3) == []
    res = dec.run_lattice([feats[0], feats[0][:1]], 3, genBeam=250.0)
    assert res[0] is not None and res[1] is None                     # one frame cannot reach the end of the network
    with pytest.raises(native.HtkAmdError):
        dec.run_lattice([feats[0]], 9)                               # at most 8 tokens per state
    with pytest.raises(native.HtkAmdError):
        dec.run_lattice([feats[0]], 3, genBeam=250.0, maxNodes=4)    # a lattice that does not fit is an error, not a truncated file


@pytest.mark.gpu
def test_token_set_kernel_wide_fan_in(native, oracle, tmp_path):
    """A 400-word loop: the loop's null node has 400 predecessors, which the kernel merges as contiguous runs per thread and then run by
    run (decode_n.hip); the oracle merges them one after the other.  Same nodes, same arcs; the relative likelihoods of alternatives are
    floats re-based in another association, so their acoustic scores are compared to 1e-3."""
    from htk_amd import synth
    s = synth.generate(150, 4, 400, 3, 70, 17, D=13)
    d = tmp_path
    synth.write_mmf(str(d / "MMF"), s, kind="USER")
    names = ["p%d" % i for i in range(400)]
    (d / "hmmlist").write_text("\n".join(names) + "\n")
    (d / "dict").write_text("".join("%s %s\n" % (n, n) for n in names))
    V = len(names)
    with open(d / "net.slf", "w") as f:
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
    model = native.Model(mmf.packed()); om = oracle.Model(mmf.packed())
    lats = native.Decoder(model, net).run_lattice(s.feats, 3, genBeam=150.0)
    for u, got in enumerate(lats):
        ref = oracle.decode_nbest(om, s.feats[u], net.arrays(), 3, genBeam=150.0)
        assert got is not None and ref is not None and got["total"] == ref["total"]
        key = lambda l: sorted(zip(l["nodeFrame"][l["arcStart"]].tolist(), l["nodeNet"][l["arcStart"]].tolist(), l["nodeFrame"][l["arcEnd"]].tolist(),
                                   l["nodeNet"][l["arcEnd"]].tolist(), l["arcLm"].tolist(), l["arcAc"].tolist()))
        g, r = key(got), key(ref)
        assert len(g) == len(r) and len(g) > 50
        assert [x[:5] for x in g] == [x[:5] for x in r]
        assert np.allclose([x[5] for x in g], [x[5] for x in r], atol=1e-3)
