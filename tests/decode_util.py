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
    if "-b" in t:                                                # -b word: alignment only (boundary word), not a decoder parameter
        del t[t.index("-b"):t.index("-b") + 2]
    t = [x for x in t if x != "-m"]
    p = dict(genBeam=1.0e10, wordBeam=1.0e10, lmScale=1.0, wordPen=0.0, prScale=1.0)
    key = {"-t": "genBeam", "-v": "wordBeam", "-s": "lmScale", "-p": "wordPen", "-r": "prScale", "-u": "maxActive"}
    for i in range(0, len(t), 2):
        p[key[t[i]]] = int(t[i + 1]) if t[i] == "-u" else float(t[i + 1])
    return p


def format_words(words, out_syms, frame_dur=100000):
    """The lines of the .rec file: start end outsym score (%f of the float); words without output symbol are dropped
    (TranscriptionFromLattice HRec.c:2342-2356)."""
    return ["%d %d %s %f" % (s * frame_dur, e * frame_dur, out_syms[w], np.float32(sc)) for w, s, e, sc in words if out_syms[w] != ""]


def format_model_labels(words, lms, align, pron_models, phys_names, out_syms, lmScale, wordPen, frame_dur=100000):
    """The lines HVite -m writes for a recognised utterance (TranscriptionFromLattice HRec.c:2289-2337): one per model,
    `start end model score`, the first model of a word followed by the word's output symbol and its LM score
    LArcTotLMLike = lmlike*lmscale + wdpenalty (HNet.h:252).  `align` = forced alignment of the recognised model chain."""
    lines, q = [], 0
    for (w, s, e, sc), lm in zip(words, lms):
        for k, m in enumerate(pron_models[w]):
            ln = "%d %d %s %f" % (align["modStart"][q] * frame_dur, align["modEnd"][q] * frame_dur, phys_names[m], np.float32(align["modScore"][q]))
            if k == 0:
                aux = np.float32(np.float64(np.float32(np.float32(lm) * np.float32(lmScale))) + np.float64(np.float32(wordPen)))
                ln += " %s" % out_syms[w]
                if aux != 0.0:                                       # an auxiliary score of 0.0 is not written (HLabel.c SaveHTKLabels)
                    ln += " %f" % aux
            lines.append(ln)
            q += 1
    return lines
