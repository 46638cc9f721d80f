"""Helpers shared by the decoding tests: load a golden case, turn HVite options into decoder parameters, format labels."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "decode")


def load_decode_case(native, name):
    """name: a directory of tests/golden/decode, or "xwrd:net" / "xwrd:loop" (the two lattices of the cross-word case, built with its
    `config`: FORCECXTEXP = T, ALLOWXWRDEXP = T)."""
    name, _, sub = name.partition(":")
    d = os.path.join(GOLD, name)
    mmf = native.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    flags = 0
    if os.path.exists(os.path.join(d, "config")):
        cfg = dict((k.strip().upper(), v.strip().upper()) for k, v in (l.split("=") for l in open(os.path.join(d, "config")) if "=" in l))
        for key, bit in (("ALLOWXWRDEXP", native.NET_ALLOWXWRDEXP), ("FORCECXTEXP", native.NET_FORCECXTEXP),
                         ("FORCELEFTBI", native.NET_FORCELEFTBI), ("FORCERIGHTBI", native.NET_FORCERIGHTBI)):
            if cfg.get(key, "F").startswith("T"):
                flags |= bit
    sfx = "_" + sub if sub else ""
    net = native.Net(os.path.join(d, (sub if sub else "net") + ".slf"), os.path.join(d, "dict"), mmf, flags=flags)
    z = np.load(os.path.join(d, "feats%s.npz" % sfx))
    feats = [z["u%d" % u] for u in range(len(z.files))]
    expected = json.load(open(os.path.join(d, "expected%s.json" % sfx)))
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


def format_model_labels(words, lms, align, pron_models, phys_names, out_syms, lmScale, wordPen, frame_dur=100000, seq_models=None):
    """The lines HVite -m writes for a recognised utterance (TranscriptionFromLattice HRec.c:2289-2337): one per model,
    `start end model score`, the first model of a word followed by the word's output symbol and its LM score
    LArcTotLMLike = lmlike*lmscale + wdpenalty (HNet.h:252).  `align` = forced alignment of the recognised model chain."""
    rows, q = [], 0
    for pos, ((w, s, e, sc), lm) in enumerate(zip(words, lms)):
        for k, m in enumerate(seq_models[pos] if seq_models is not None else pron_models[w]):       # seq_models: cross-word networks
            ln = "%d %d %s %f" % (align["modStart"][q] * frame_dur, align["modEnd"][q] * frame_dur, phys_names[m], np.float32(align["modScore"][q]))
            aux = None
            if k == 0:
                aux = np.float32(np.float64(np.float32(np.float32(lm) * np.float32(lmScale))) + np.float64(np.float32(wordPen)))
                ln += " %s" % out_syms[w]
            rows.append((ln, aux))
            q += 1
    # a score column is written when any label of the file has a non-zero score in it (SaveHTKLabels HLabel.c:1493-1517)
    any_aux = any(a is not None and a != 0.0 for _, a in rows)
    return [ln + (" %f" % a if (a is not None and any_aux) else "") for ln, a in rows]
