"""Helpers shared by the decoding tests: load a golden case, turn HVite options into decoder parameters, format labels."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "decode")


def load_decode_case(native, name):
    d = os.path.join(GOLD, name)
    mmf = native.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    net = native.Net(os.path.join(d, "net.slf"), os.path.join(d, "dict"), mmf)
    z = np.load(os.path.join(d, "feats.npz"))
    feats = [z["u%d" % u] for u in range(len(z.files))]
    expected = json.load(open(os.path.join(d, "expected.json")))
    return mmf, net, feats, expected


def parse_opts(opts: str) -> dict:
    """HVite switches -> decoder parameters (HVite.c:81-95 defaults: -s 1.0 -p 0.0 -r 1.0, beams off)."""
    t = opts.split()
    p = dict(genBeam=1.0e10, wordBeam=1.0e10, lmScale=1.0, wordPen=0.0, prScale=1.0)
    key = {"-t": "genBeam", "-v": "wordBeam", "-s": "lmScale", "-p": "wordPen", "-r": "prScale"}
    for i in range(0, len(t), 2):
        p[key[t[i]]] = float(t[i + 1])
    return p


def format_words(words, out_syms, frame_dur=100000):
    """The lines of the .rec file: start end outsym score (%f of the float); words without output symbol are dropped
    (TranscriptionFromLattice HRec.c:2342-2356)."""
    return ["%d %d %s %f" % (s * frame_dur, e * frame_dur, out_syms[w], np.float32(sc)) for w, s, e, sc in words if out_syms[w] != ""]
