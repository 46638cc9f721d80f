"""Readers for artefacts written by the reference build in oracle/_ref (TEST INFRASTRUCTURE).

* read_fbdump : binary dump of oracle/ref_fbdump.c (alpha/beta/outprob/occupation per utterance)
* read_acc    : HERest accumulator file, layout of DumpAccs (HTrain.c:1453-1505): per physical HMM in
                HMMScan order: quoted name, int32 nEgs, per not-yet-seen state WtAcc (M floats + occ) followed
                by per not-yet-seen Gaussian MuAcc (D floats + occ) and VaAcc (D floats + occ), per not-yet-seen
                transP TrAcc (N*N + N floats), int32 marker 123456; tail: float totalPr, int32 totalT
                (HERest.c:546-548).  Big-endian (HShell.c:1638 WriteFloat).
* read_rec    : label file written by HVite (start end label score ...)
"""
from __future__ import annotations

import struct

import numpy as np


def read_fbdump(path: str) -> list[dict]:
    out = []
    with open(path, "rb") as f:
        (n,) = struct.unpack("i", f.read(4))
        for _ in range(n):
            ok, T, Q, maxN = struct.unpack("iiii", f.read(16))
            if not ok:
                out.append(dict(ok=0, T=T, Q=Q))
                continue
            (pr,) = struct.unpack("d", f.read(8))
            rd = lambda dt, cnt: np.frombuffer(f.read(cnt * np.dtype(dt).itemsize), dtype=dt).copy()
            qLo = rd(np.int32, T); qHi = rd(np.int32, T); aLo = rd(np.int32, T); aHi = rd(np.int32, T)
            k = T * Q * maxN
            beta = rd(np.float64, k).reshape(T, Q, maxN)
            alpha = rd(np.float64, k).reshape(T, Q, maxN)
            outp = rd(np.float32, k).reshape(T, Q, maxN)
            occ = rd(np.float32, k).reshape(T, Q, maxN)
            out.append(dict(ok=1, T=T, Q=Q, maxN=maxN, pr=pr, qLo=qLo, qHi=qHi, aLo=aLo, aHi=aHi,
                            beta=beta, alpha=alpha, outp=outp, occ=occ))
    return out


def read_acc(path: str, pk: dict, names: list[str]) -> dict:
    """pk: packed model dict (htk_amd.synth.SynthSet.packed() layout); names[h] = physical HMM name."""
    D = int(pk["vecSize"])
    S, Cn, G = int(pk["numStates"]), int(pk["numComp"]), int(pk["numGauss"])
    idx = {n: i for i, n in enumerate(names)}
    a = dict(mu=np.zeros((G, D), np.float32), muOcc=np.zeros(G, np.float32),
             va=np.zeros((G, D), np.float32), vaOcc=np.zeros(G, np.float32),
             wt=np.zeros(Cn, np.float32), wtOcc=np.zeros(S, np.float32),
             tr=np.zeros(int(pk["transOff"][-1]), np.float32),
             trOcc=np.zeros(int(np.sum(pk["transN"])), np.float32),
             nEgs=np.zeros(int(pk["numPhys"]), np.int32), order=[])
    occOff = np.concatenate([[0], np.cumsum(pk["transN"])])
    seenS, seenG, seenT = set(), set(), set()
    buf = open(path, "rb").read()
    pos = 0
    nH = int(pk["numPhys"])

    def floats(n):
        nonlocal pos
        v = np.frombuffer(buf, dtype=">f4", count=n, offset=pos).astype(np.float32)
        pos += 4 * n
        return v

    for _ in range(nH):
        assert buf[pos:pos + 1] == b'"', "quoted physical HMM name expected"
        e = buf.index(b'"', pos + 1)
        name = buf[pos + 1:e].decode()
        pos = e + 1
        assert buf[pos:pos + 1] == b"\n"
        pos += 1
        h = idx[name]
        a["order"].append(h)
        (a["nEgs"][h],) = struct.unpack(">i", buf[pos:pos + 4]); pos += 4
        ti = int(pk["hmmTrans"][h]); N = int(pk["transN"][ti])
        for j in range(N - 2):
            s = int(pk["hmmState"][pk["hmmStateOff"][h] + j])
            if s in seenS:
                continue
            seenS.add(s)
            c0, c1 = int(pk["stateCompOff"][s]), int(pk["stateCompOff"][s + 1])
            a["wt"][c0:c1] = floats(c1 - c0)
            a["wtOcc"][s] = floats(1)[0]
            for c in range(c0, c1):
                g = int(pk["compGauss"][c])
                if g in seenG:
                    continue
                seenG.add(g)
                a["mu"][g] = floats(D); a["muOcc"][g] = floats(1)[0]
                a["va"][g] = floats(D); a["vaOcc"][g] = floats(1)[0]
        if ti not in seenT:
            seenT.add(ti)
            o = int(pk["transOff"][ti])
            a["tr"][o:o + N * N] = floats(N * N)
            a["trOcc"][occOff[ti]:occOff[ti] + N] = floats(N)
        (mark,) = struct.unpack(">i", buf[pos:pos + 4]); pos += 4
        assert mark == 123456, "marker"
    a["totalPr"] = floats(1)[0]
    (a["totalT"],) = struct.unpack(">i", buf[pos:pos + 4]); pos += 4
    assert pos == len(buf), (pos, len(buf))
    return a


def read_rec(path: str) -> list[tuple]:
    rows = []
    for line in open(path):
        p = line.split()
        if not p or p[0] in ("#!MLF!#", ".") or p[0].startswith('"'):
            continue
        rows.append(tuple(p))
    return rows


def read_mmf_text(path: str, names: list[str], state_names: list[str] | None = None) -> dict:
    """Minimal text-MMF reader for checking (TEST TOOLING; the product parser is htk_amd/host/mmf.c).
    Handles ~o, ~t "x" <TRANSP>, ~s "x" [<NUMMIXES>] {<MIXTURE> m w <MEAN> <VARIANCE> [<GCONST>]},
    ~h "x" <BEGINHMM> <NUMSTATES> {<STATE> j ~s "x"} ~t "x" <ENDHMM>; returns packed-layout arrays with tied
    states in order of first definition and physical HMMs in the order of `names`."""
    import re
    toks = re.findall(r'~\w|"[^"]*"|<[^>]+>|[^\s<>~"]+', open(path).read())
    i = 0
    D = None
    states, sidx = [], {}
    trans, tidx = [], {}
    hmms = {}

    def num(k):
        nonlocal i
        v = [float(x) for x in toks[i:i + k]]
        i += k
        return v

    def parse_state():
        nonlocal i
        M = 1
        if toks[i].upper() == "<NUMMIXES>":
            M = int(toks[i + 1]); i += 2
        comps = [(0.0, None, None, None)] * M     # components missing from the file have weight 0
        if M == 1 and toks[i].upper() != "<MIXTURE>":
            idxs = [0]
        else:
            idxs = None
        k = 0
        while True:
            w = 1.0
            if idxs is None:
                if i >= len(toks) or toks[i].upper() != "<MIXTURE>":
                    break
                k = int(toks[i + 1]) - 1; w = float(toks[i + 2]); i += 3
            assert toks[i].upper() == "<MEAN>"; n = int(toks[i + 1]); i += 2; mu = num(n)
            assert toks[i].upper() == "<VARIANCE>"; n = int(toks[i + 1]); i += 2; va = num(n)
            gc = None
            if i < len(toks) and toks[i].upper() == "<GCONST>":
                gc = float(toks[i + 1]); i += 2
            comps[k] = (w, mu, va, gc)
            if idxs is not None:
                break
        return comps

    while i < len(toks):
        t = toks[i]
        if t == "~o":
            i += 1
            while i < len(toks) and not toks[i].startswith("~"):
                if toks[i].upper() == "<VECSIZE>":
                    D = int(toks[i + 1]); i += 2
                elif toks[i].upper() == "<STREAMINFO>":
                    i += 2 + int(toks[i + 1])
                else:
                    i += 1
        elif t == "~t":
            name = toks[i + 1].strip('"'); i += 2
            assert toks[i].upper() == "<TRANSP>"; N = int(toks[i + 1]); i += 2
            tidx[name] = len(trans); trans.append((N, num(N * N)))
        elif t == "~s":
            name = toks[i + 1].strip('"'); i += 2
            sidx[name] = len(states); states.append(parse_state())
        elif t == "~h":
            name = toks[i + 1].strip('"'); i += 2
            assert toks[i].upper() == "<BEGINHMM>"; i += 1
            assert toks[i].upper() == "<NUMSTATES>"; N = int(toks[i + 1]); i += 2
            st = []
            while toks[i].upper() == "<STATE>":
                i += 2
                assert toks[i] == "~s"; st.append(sidx[toks[i + 1].strip('"')]); i += 2
            assert toks[i] == "~t"; ti = tidx[toks[i + 1].strip('"')]; i += 2
            assert toks[i].upper() == "<ENDHMM>"; i += 1
            hmms[name] = (ti, st)
        else:
            raise ValueError("unexpected token %r" % t)
    if state_names is not None:            # SaveHMMSet writes macros in hash-table order: re-index by name
        order = [sidx[n] for n in state_names]
        remap = {old: new for new, old in enumerate(order)}
        states = [states[o] for o in order]
        hmms = {n: (ti, [remap[x] for x in st]) for n, (ti, st) in hmms.items()}
    G = sum(len(s) for s in states)
    mean = np.zeros((G, D), np.float32); var = np.zeros((G, D), np.float32)
    w = np.zeros(G, np.float32); gc = np.full(G, np.nan, np.float32)
    off = [0]; g = 0
    for s in states:
        for (ww, mu, va, gg) in s:
            w[g] = ww
            if mu is not None:
                mean[g] = mu; var[g] = va
            else:
                mean[g] = np.nan; var[g] = np.nan
            if gg is not None:
                gc[g] = gg
            g += 1
        off.append(g)
    tp = []
    for (N, v) in trans:
        a = np.array(v, np.float64)
        with np.errstate(divide="ignore"):
            tp.append(np.where(a > 0, np.log(a), -1.0e10).astype(np.float32))
    return dict(vecSize=D, numStates=len(states), numComp=G, numGauss=G,
                stateCompOff=np.array(off, np.int32), compWeight=w, compGauss=np.arange(G, dtype=np.int32),
                mean=mean, var=var, gconst=gc, numTrans=len(trans),
                transN=np.array([t[0] for t in trans], np.int32),
                transOff=np.concatenate([[0], np.cumsum([t[0] ** 2 for t in trans])]).astype(np.int32),
                transP=np.concatenate(tp), transLin=[np.array(t[1], np.float32) for t in trans],
                numPhys=len(names), hmmTrans=np.array([hmms[n][0] for n in names], np.int32),
                hmmStateOff=np.concatenate([[0], np.cumsum([len(hmms[n][1]) for n in names])]).astype(np.int32),
                hmmState=np.concatenate([hmms[n][1] for n in names]).astype(np.int32))
