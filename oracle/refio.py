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
