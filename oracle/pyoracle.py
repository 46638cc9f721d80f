"""ctypes binding of the CPU oracle (oracle/htk_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "libhtk_oracle.so")
SRCS = ["htk_oracle.c", "orc_viterbi.c", "orc_mfcc.c", "orc_decode.c", "orc_decode_n.c"]

LZERO = -1.0e10
LSMALL = -0.5e10
UPMEANS, UPVARS, UPTRANS, UPMIXES = 1, 2, 4, 8
UPALL = 15
NOPRUNE = 1.0e20


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, s) for s in SRCS if os.path.exists(os.path.join(HERE, s))]
    deps = srcs + [os.path.join(HERE, "htk_oracle.h"), os.path.join(HERE, "orc_ilist.h")]
    if (not force) and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    os.makedirs(BUILD, exist_ok=True)
    # no FMA contraction, no fast-math: the reference is SSE2 scalar float
    cmd = ["gcc", "-O2", "-std=gnu99", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_ladd.restype = C.c_double
        _lib.orc_ladd.argtypes = [C.c_double, C.c_double]
        _lib.orc_mix_log_weight.restype = C.c_float
        _lib.orc_mix_log_weight.argtypes = [C.c_float]
        _lib.orc_idoutp.restype = C.c_float
        _lib.orc_state_outp.restype = C.c_float
        _lib.orc_soutp.restype = C.c_float
        _lib.orc_min_dur.restype = C.c_int
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


class CModel(C.Structure):
    _fields_ = [("D", C.c_int), ("S", C.c_int), ("C", C.c_int), ("G", C.c_int), ("nT", C.c_int), ("H", C.c_int),
                ("stateCompOff", C.c_void_p), ("compLogWt", C.c_void_p), ("compGauss", C.c_void_p),
                ("mean", C.c_void_p), ("ivar", C.c_void_p), ("gconst", C.c_void_p),
                ("transN", C.c_void_p), ("transOff", C.c_void_p), ("transP", C.c_void_p),
                ("hmmTrans", C.c_void_p), ("hmmStateOff", C.c_void_p), ("hmmState", C.c_void_p),
                ("NSt", C.c_int), ("dimStream", C.c_void_p), ("tiedMix", C.c_int), ("compWeight", C.c_void_p), ("var", C.c_void_p), ("msIntended", C.c_int)]


class CAccs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs")]


class CFbCfg(C.Structure):
    _fields_ = [("pruneInit", C.c_double), ("pruneInc", C.c_double), ("pruneLim", C.c_double),
                ("minFrwdP", C.c_float), ("uFlags", C.c_int)]


class CFbDump(C.Structure):
    _fields_ = [("beta", C.c_void_p), ("alpha", C.c_void_p), ("outp", C.c_void_p),
                ("qLo", C.c_void_p), ("qHi", C.c_void_p), ("aLo", C.c_void_p), ("aHi", C.c_void_p),
                ("occ", C.c_void_p), ("nEval", C.c_longlong)]


class CUpdCfg(C.Structure):
    _fields_ = [("minEgs", C.c_int), ("minVar", C.c_float), ("mixWeightFloor", C.c_float), ("uFlags", C.c_int),
                ("singleProcess", C.c_int)]


class CUpdStats(C.Structure):
    _fields_ = [("nFloorVar", C.c_int), ("nFloorVarMix", C.c_int), ("nSkippedHmm", C.c_int)]


class Model:
    """Packed model (same field names as include/htk_amd.h) prepared the way HERest/HVite prepare an HMMSet:
    FixDiagGConst if no gconst (HModel.c:206-208), ConvDiagC (HUtil.c:413), ConvLogWt (HUtil.c:474)."""

    def __init__(self, pk: dict, ms_intended: bool = False):
        L = lib()
        self.pk = pk
        # several streams (pk["numStreams"] > 1): stateCompOff per (state, stream), Gaussians in undivided rows (htk_oracle.h)
        self.NSt = int(pk.get("numStreams", 1) or 1)
        self.dimStream = np.ascontiguousarray(pk["dimStream"], np.int32) if self.NSt > 1 else None
        self.ms_intended = bool(ms_intended)
        self.tiedMix = int(pk.get("hsKind", 0) or 0) == 1
        self.D = int(pk["vecSize"]); self.S = int(pk["numStates"]); self.C = int(pk["numComp"])
        self.G = int(pk["numGauss"]); self.nT = int(pk["numTrans"]); self.H = int(pk["numPhys"])
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self.stateCompOff = i32(pk["stateCompOff"]); self.compWeight = f32(pk["compWeight"]).copy()
        self.compGauss = i32(pk["compGauss"]); self.mean = f32(pk["mean"]).reshape(self.G, self.D).copy()
        self.var = f32(pk["var"]).reshape(self.G, self.D).copy()
        self.transN = i32(pk["transN"]); self.transOff = i32(pk["transOff"]); self.transP = f32(pk["transP"]).copy()
        self.hmmTrans = i32(pk["hmmTrans"]); self.hmmStateOff = i32(pk["hmmStateOff"]); self.hmmState = i32(pk["hmmState"])
        if pk.get("gconst") is None:
            self.gconst = np.empty(self.G, np.float32)
            gs = np.zeros(self.G, np.int32)
            if self.NSt > 1:
                for e in range(self.S * self.NSt):
                    gs[self.compGauss[self.stateCompOff[e]:self.stateCompOff[e + 1]]] = e % self.NSt
            for g in range(self.G):
                L.orc_fix_diag_gconst_ms(C.c_int(self.D), _p(self.var[g]), _p(self.dimStream), C.c_int(int(gs[g])), C.c_void_p(self.gconst.ctypes.data + 4 * g))
        else:
            self.gconst = f32(pk["gconst"]).copy()
        self.refresh()

    def refresh(self):
        """Recompute the derived arrays (ivar, log weights) after the parameters changed."""
        L = lib()
        self.ivar = np.empty_like(self.var)
        L.orc_conv_diagc(C.c_int(self.G * self.D), _p(self.var), _p(self.ivar))
        self.compLogWt = np.array([L.orc_mix_log_weight(C.c_float(w)) for w in self.compWeight], np.float32)
        self.c = CModel(self.D, self.S, self.C, self.G, self.nT, self.H,
                        _p(self.stateCompOff), _p(self.compLogWt), _p(self.compGauss),
                        _p(self.mean), _p(self.ivar), _p(self.gconst),
                        _p(self.transN), _p(self.transOff), _p(self.transP),
                        _p(self.hmmTrans), _p(self.hmmStateOff), _p(self.hmmState),
                        self.NSt, _p(self.dimStream), int(self.tiedMix), _p(self.compWeight), _p(self.var), int(self.ms_intended))

    @property
    def maxN(self):
        return int(self.transN.max())

    def state_outp(self, s: int, x: np.ndarray, want_mix=False):
        x = np.ascontiguousarray(x, np.float32)
        M = int(self.stateCompOff[s + 1] - self.stateCompOff[s])
        mix = np.empty(M, np.float32) if want_mix else None
        v = lib().orc_state_outp(C.byref(self.c), C.c_int(s), _p(x), _p(mix))
        return (v, mix) if want_mix else v

    def soutp(self, s: int, x: np.ndarray):
        x = np.ascontiguousarray(x, np.float32)
        return lib().orc_soutp(C.byref(self.c), C.c_int(s), _p(x))

    def soutp_block(self, X: np.ndarray, states: np.ndarray, diagc: bool = False) -> np.ndarray:
        """SOutP's rounding (double log-sum, one float at the end); diagc: DOutP's xmm*xmm/var form"""
        X = np.ascontiguousarray(X, np.float32); states = np.ascontiguousarray(states, np.int32)
        out = np.empty((X.shape[0], len(states)), np.float32)
        lib().orc_soutp_block(C.byref(self.c), _p(self.var) if diagc else None, _p(X), C.c_int(X.shape[0]), _p(states), C.c_int(len(states)), _p(out))
        return out

    def score_block_diagc(self, X: np.ndarray, states: np.ndarray) -> np.ndarray:
        X = np.ascontiguousarray(X, np.float32); states = np.ascontiguousarray(states, np.int32)
        out = np.empty((X.shape[0], len(states)), np.float32)
        lib().orc_score_block_diagc(C.byref(self.c), _p(self.var), _p(X), C.c_int(X.shape[0]), _p(states), C.c_int(len(states)), _p(out))
        return out

    def score_block(self, X: np.ndarray, states: np.ndarray) -> np.ndarray:
        X = np.ascontiguousarray(X, np.float32); states = np.ascontiguousarray(states, np.int32)
        out = np.empty((X.shape[0], len(states)), np.float32)
        lib().orc_score_block(C.byref(self.c), _p(X), C.c_int(X.shape[0]), _p(states), C.c_int(len(states)), _p(out))
        return out


class Accs:
    def __init__(self, m: Model):
        self.m = m
        self.mu = np.zeros((m.G, m.D), np.float32); self.muOcc = np.zeros(m.G, np.float32)
        self.va = np.zeros((m.G, m.D), np.float32); self.vaOcc = np.zeros(m.G, np.float32)
        self.wt = np.zeros(m.C, np.float32); self.wtOcc = np.zeros(m.S * getattr(m, "NSt", 1), np.float32)
        self.tr = np.zeros(int(m.transOff[-1]), np.float32); self.trOcc = np.zeros(int(m.transN.sum()), np.float32)
        self.nEgs = np.zeros(m.H, np.int32)
        self.c = CAccs(*[_p(getattr(self, n)) for n in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs")])


def fb_cfg(pruneInit=NOPRUNE, pruneInc=0.0, pruneLim=NOPRUNE, minFrwdP=10.0, uFlags=UPALL):
    return CFbCfg(pruneInit, pruneInc, pruneLim, minFrwdP, uFlags)


def fb_utt(m: Model, cfg: CFbCfg, X: np.ndarray, labs: np.ndarray, acc: Accs, dump: bool = False):
    """FBFile (HFB.c:1923) for one utterance.  Returns (rc, pr, dumpdict|None)."""
    X = np.ascontiguousarray(X, np.float32); labs = np.ascontiguousarray(labs, np.int32)
    T, Q = X.shape[0], len(labs)
    pr = C.c_double(0.0)
    d = None; cd = None
    if dump:
        maxN = m.maxN
        d = dict(beta=np.empty((T, Q, maxN)), alpha=np.empty((T, Q, maxN)),
                 outp=np.empty((T, Q, maxN), np.float32), occ=np.empty((T, Q, maxN), np.float32),
                 qLo=np.zeros(T, np.int32), qHi=np.zeros(T, np.int32), aLo=np.zeros(T, np.int32), aHi=np.zeros(T, np.int32))
        cd = CFbDump(_p(d["beta"]), _p(d["alpha"]), _p(d["outp"]), _p(d["qLo"]), _p(d["qHi"]),
                     _p(d["aLo"]), _p(d["aHi"]), _p(d["occ"]), 0)
    rc = lib().orc_fb_utt(C.byref(m.c), C.byref(cfg), _p(X), C.c_int(T), _p(labs), C.c_int(Q),
                          C.byref(acc.c), C.byref(pr), C.byref(cd) if cd is not None else None)
    if d is not None:
        d["nEval"] = cd.nEval
    return rc, pr.value, d


def update(m: Model, acc: Accs, minEgs=3, minVar=0.0, mixWeightFloor=0.0, uFlags=UPALL, singleProcess=False):
    """MLUpdateModels (HERest.c:1262) in place on m.mean / m.var / m.gconst / m.compWeight / m.transP."""
    cfg = CUpdCfg(minEgs, minVar, mixWeightFloor, uFlags, int(singleProcess))
    st = CUpdStats()
    lib().orc_update(C.byref(m.c), C.byref(acc.c), C.byref(cfg), _p(m.mean), _p(m.var), _p(m.gconst),
                     _p(m.compWeight), _p(m.transP), C.byref(st))
    m.refresh()
    return dict(nFloorVar=st.nFloorVar, nFloorVarMix=st.nFloorVarMix, nSkippedHmm=st.nSkippedHmm)


def viterbi_align(m: Model, X: np.ndarray, labs: np.ndarray, genBeam: float = 1.0e10):
    """HVite -a -f -m on a chain of physical models (HRec.c token passing).  Returns dict or None if no token survived."""
    X = np.ascontiguousarray(X, np.float32); labs = np.ascontiguousarray(labs, np.int32)
    T, Q = X.shape[0], len(labs)
    maxSeg = int(sum(m.transN[m.hmmTrans[h]] - 2 for h in labs))
    sq = np.zeros(maxSeg, np.int32); ss = np.zeros(maxSeg, np.int32); s0 = np.zeros(maxSeg, np.int32); s1 = np.zeros(maxSeg, np.int32)
    sc = np.zeros(maxSeg, np.float64)
    m0 = np.zeros(Q, np.int32); m1 = np.zeros(Q, np.int32); msc = np.zeros(Q, np.float64)
    tot = C.c_double(0.0)
    n = lib().orc_viterbi_align(C.byref(m.c), _p(X), C.c_int(T), _p(labs), C.c_int(Q), C.c_float(genBeam), C.c_int(maxSeg),
                                _p(sq), _p(ss), _p(s0), _p(s1), _p(sc), _p(m0), _p(m1), _p(msc), C.byref(tot))
    if n < 0:
        return None
    return dict(n=n, q=sq[:n], state=ss[:n], start=s0[:n], end=s1[:n], score=sc[:n], modStart=m0, modEnd=m1, modScore=msc,
                total=tot.value)


def format_rec(res: dict, labs, names, frame_dur: int = 100000):
    """The label lines HVite -a -f -m writes (start end s<j> score [model modelscore [word]]), HRec.c:2300-2335."""
    lines = []
    lastq = -1
    for k in range(res["n"]):
        q = int(res["q"][k])
        s = "%d %d s%d %f" % (res["start"][k] * frame_dur, res["end"][k] * frame_dur, res["state"][k], np.float32(res["score"][k]))
        if q != lastq:
            nm = names[labs[q - 1]]
            s += " %s %f %s" % (nm, np.float32(res["modScore"][q - 1]), nm)
            lastq = q
        lines.append(s)
    return lines


class CMfccCfg(C.Structure):
    _fields_ = [("sampPeriod", C.c_double), ("winDur", C.c_double), ("frPeriod", C.c_double),
                ("numChans", C.c_int), ("numCeps", C.c_int), ("cepLifter", C.c_int), ("preEmph", C.c_float),
                ("useHam", C.c_int), ("usePower", C.c_int), ("zMeanSource", C.c_int), ("rawEnergy", C.c_int), ("eNormalise", C.c_int),
                ("loFreq", C.c_float), ("hiFreq", C.c_float), ("cepScale", C.c_float), ("silFloor", C.c_float), ("eScale", C.c_float),
                ("hasC0", C.c_int), ("hasE", C.c_int), ("hasD", C.c_int), ("hasA", C.c_int), ("hasZ", C.c_int),
                ("delWin", C.c_int), ("accWin", C.c_int)]


def mfcc_cfg(kind="MFCC_0_D_A", sampPeriod=625.0, winDur=250000.0, frPeriod=100000.0, numChans=26, numCeps=12, cepLifter=22,
             preEmph=0.97, useHam=True, usePower=False, zMeanSource=False, rawEnergy=True, eNormalise=True,
             loFreq=-1.0, hiFreq=-1.0, cepScale=1.0, silFloor=50.0, eScale=0.1, delWin=2, accWin=2):
    """HParm defaults (HParm.c:337-367) for the MFCC path; `kind` is the TARGETKIND string."""
    q = kind.upper().split("_")
    assert q[0] == "MFCC"
    return CMfccCfg(sampPeriod, winDur, frPeriod, numChans, numCeps, cepLifter, preEmph, int(useHam), int(usePower), int(zMeanSource),
                    int(rawEnergy), int(eNormalise), loFreq, hiFreq, cepScale, silFloor, eScale,
                    int("0" in q[1:]), int("E" in q[1:]), int("D" in q[1:]), int("A" in q[1:]), int("Z" in q[1:]), delWin, accWin)


def add_qualifiers(stat: np.ndarray, hasD=True, hasA=False, delWin=2, accWin=2) -> np.ndarray:
    """AddQualifiers (HParm.c:1618) on a parameterised table."""
    stat = np.ascontiguousarray(stat, np.float32)
    T, n = stat.shape
    out = np.zeros((T, n * (1 + int(hasD) + int(hasA))), np.float32)
    lib().orc_add_qualifiers(_p(stat), C.c_int(T), C.c_int(n), C.c_int(hasD), C.c_int(hasA), C.c_int(delWin), C.c_int(accWin), _p(out))
    return out


def compv(X: np.ndarray, minVar=0.0):
    """HCompV's global mean / variance (float accumulators in frame order)."""
    X = np.ascontiguousarray(X, np.float32)
    T, D = X.shape
    mean = np.zeros(D, np.float32); var = np.zeros(D, np.float32)
    lib().orc_compv(_p(X), C.c_long(T), C.c_int(D), C.c_float(minVar), _p(mean), _p(var))
    return mean, var


def parm_qualify(stat: np.ndarray, nZeroMean=0, hasD=True, hasA=False, hasT=False, delWin=2, accWin=2, thirdWin=2, nullECol=-1,
                 v1Compat=False, simpleDiffs=False) -> np.ndarray:
    """AddQualifiers incl. third differentials and _Z on a table, then the _N column drop of ExtractObservation."""
    stat = np.ascontiguousarray(stat, np.float32)
    T, n = stat.shape
    cols = n * (1 + int(hasD) + int(hasA) + int(hasT)) - int(nullECol >= 0)
    out = np.zeros((T, cols), np.float32)
    got = lib().orc_parm_qualify2(_p(stat), C.c_int(T), C.c_int(n), C.c_int(nZeroMean), C.c_int(hasD), C.c_int(hasA), C.c_int(hasT),
                                  C.c_int(delWin), C.c_int(accWin), C.c_int(thirdWin), C.c_int(nullECol), C.c_int(v1Compat), C.c_int(simpleDiffs), _p(out))
    assert got == cols
    return out


def mfcc(wav: np.ndarray, cfg: CMfccCfg) -> np.ndarray:
    wav = np.ascontiguousarray(wav, np.int16)
    T = lib().orc_mfcc_frames(C.c_int(len(wav)), C.byref(cfg), None, None)
    cols = lib().orc_mfcc_cols(C.byref(cfg))
    out = np.zeros((max(T, 0), cols), np.float32)
    if T > 0:
        lib().orc_mfcc(_p(wav), C.c_int(len(wav)), C.byref(cfg), _p(out))
    return out


def decode_nbest(model: "Model", X: np.ndarray, net: dict, nToks: int, genBeam=1.0e10, wordBeam=1.0e10, nBeam=None, lmScale=1.0, wordPen=0.0, prScale=1.0,
                 maxNodes=20000, maxArcs=80000, maxActive=0, align=0, maxAlign=400000):
    """HRec with nToks > 1 (HVite -n): the lattice of CreateLattice as a dict of arrays, or None when no token reached the final node.
    nBeam defaults to genBeam (HVite.c:546).  align: 1 = model records (HVite -m), 2 = state records (-f), 3 = both: the lattice then has
    arcAlignOff / alState / alNode / alDur (frames) / alLike -- LatFromPaths' lAlign per arc."""
    if align:
        return _decode_nbest_align(model, X, net, nToks, genBeam, wordBeam, nBeam, lmScale, wordPen, prScale, maxNodes, maxArcs, maxActive, align, maxAlign)
    X = np.ascontiguousarray(X, np.float32)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    kind, mdl, pp, lo, ld, ll = i32(net["kind"]), i32(net["model"]), f32(net["pronProb"]), i32(net["linkOff"]), i32(net["linkDest"]), f32(net["linkLike"])
    nn = C.c_int(0); na = C.c_int(0); tot = C.c_double(0.0)
    nNet = np.zeros(maxNodes, np.int32); nFr = np.zeros(maxNodes, np.int32); nLk = np.zeros(maxNodes, np.float64)
    aS = np.zeros(maxArcs, np.int32); aE = np.zeros(maxArcs, np.int32); aAc = np.zeros(maxArcs, np.float32); aLm = np.zeros(maxArcs, np.float32)
    aPr = np.zeros(maxArcs, np.float32); aSc = np.zeros(maxArcs, np.float64)
    rc = lib().orc_decode_nbest(C.byref(model.c), _p(X), C.c_int(X.shape[0]), C.c_int(len(kind)), _p(kind), _p(mdl), _p(pp), _p(lo), _p(ld), _p(ll),
                                C.c_int(int(net["initial"])), C.c_int(int(net["final"])), C.c_float(genBeam), C.c_float(wordBeam),
                                C.c_float(genBeam if nBeam is None else nBeam), C.c_float(lmScale), C.c_float(wordPen), C.c_float(prScale), C.c_int(nToks), C.c_int(maxActive),
                                C.c_int(maxNodes), C.c_int(maxArcs), _p(nNet), _p(nFr), _p(nLk), _p(aS), _p(aE), _p(aAc), _p(aLm), _p(aPr), _p(aSc),
                                C.byref(nn), C.byref(na), C.byref(tot))
    if rc == -1:
        return None
    if rc < 0:
        raise RuntimeError("orc_decode_nbest failed (%d)" % rc)
    n, a = nn.value, na.value
    return dict(nodeNet=nNet[:n].copy(), nodeFrame=nFr[:n].copy(), nodeLike=nLk[:n].copy(), arcStart=aS[:a].copy(), arcEnd=aE[:a].copy(),
                arcAc=aAc[:a].copy(), arcLm=aLm[:a].copy(), arcPr=aPr[:a].copy(), arcScore=aSc[:a].copy(), total=tot.value)


def _decode_nbest_align(model, X, net, nToks, genBeam, wordBeam, nBeam, lmScale, wordPen, prScale, maxNodes, maxArcs, maxActive, align, maxAlign):
    X = np.ascontiguousarray(X, np.float32)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    kind, mdl, pp, lo, ld, ll = i32(net["kind"]), i32(net["model"]), f32(net["pronProb"]), i32(net["linkOff"]), i32(net["linkDest"]), f32(net["linkLike"])
    nn = C.c_int(0); na = C.c_int(0); tot = C.c_double(0.0)
    nNet = np.zeros(maxNodes, np.int32); nFr = np.zeros(maxNodes, np.int32); nLk = np.zeros(maxNodes, np.float64)
    aS = np.zeros(maxArcs, np.int32); aE = np.zeros(maxArcs, np.int32); aAc = np.zeros(maxArcs, np.float32); aLm = np.zeros(maxArcs, np.float32)
    aPr = np.zeros(maxArcs, np.float32); aSc = np.zeros(maxArcs, np.float64)
    aOff = np.zeros(maxArcs + 1, np.int32); alS = np.zeros(maxAlign, np.int32); alN = np.zeros(maxAlign, np.int32); alD = np.zeros(maxAlign, np.int32); alL = np.zeros(maxAlign, np.float32)
    rc = lib().orc_decode_nbest_align(C.byref(model.c), _p(X), C.c_int(X.shape[0]), C.c_int(len(kind)), _p(kind), _p(mdl), _p(pp), _p(lo), _p(ld), _p(ll),
                                      C.c_int(int(net["initial"])), C.c_int(int(net["final"])), C.c_float(genBeam), C.c_float(wordBeam),
                                      C.c_float(genBeam if nBeam is None else nBeam), C.c_float(lmScale), C.c_float(wordPen), C.c_float(prScale), C.c_int(nToks), C.c_int(maxActive),
                                      C.c_int(int(align)), C.c_int(maxNodes), C.c_int(maxArcs), _p(nNet), _p(nFr), _p(nLk), _p(aS), _p(aE), _p(aAc), _p(aLm), _p(aPr), _p(aSc),
                                      C.byref(nn), C.byref(na), C.byref(tot), C.c_int(maxAlign), _p(aOff), _p(alS), _p(alN), _p(alD), _p(alL))
    if rc == -1:
        return None
    if rc < 0:
        raise RuntimeError("orc_decode_nbest_align failed (%d)" % rc)
    n, a = nn.value, na.value
    k = int(aOff[a])
    return dict(nodeNet=nNet[:n].copy(), nodeFrame=nFr[:n].copy(), nodeLike=nLk[:n].copy(), arcStart=aS[:a].copy(), arcEnd=aE[:a].copy(),
                arcAc=aAc[:a].copy(), arcLm=aLm[:a].copy(), arcPr=aPr[:a].copy(), arcScore=aSc[:a].copy(), total=tot.value,
                arcAlignOff=aOff[:a + 1].copy(), alState=alS[:k].copy(), alNode=alN[:k].copy(), alDur=alD[:k].copy(), alLike=alL[:k].copy())


def decode(model: "Model", X: np.ndarray, net: dict, genBeam=1.0e10, wordBeam=1.0e10, lmScale=1.0, wordPen=0.0, prScale=1.0, maxWords=4096, maxActive=0):
    """HRec 1-best decoding over a flat network (dict with the fields of htkamd_net_desc).
    Returns (list of (pron, startFrame, endFrame, score), totalLike) or (None, LZERO)."""
    X = np.ascontiguousarray(X, np.float32)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    kind, mdl, pp, lo, ld, ll = i32(net["kind"]), i32(net["model"]), f32(net["pronProb"]), i32(net["linkOff"]), i32(net["linkDest"]), f32(net["linkLike"])
    wp = np.zeros(maxWords, np.int32); ws = np.zeros(maxWords, np.int32); we = np.zeros(maxWords, np.int32); sc = np.zeros(maxWords, np.float32)
    tot = C.c_double(0.0)
    n = lib().orc_decode_u(C.byref(model.c), _p(X), C.c_int(X.shape[0]), C.c_int(len(kind)), _p(kind), _p(mdl), _p(pp), _p(lo), _p(ld), _p(ll),
                           C.c_int(int(net["initial"])), C.c_int(int(net["final"])), C.c_float(genBeam), C.c_float(wordBeam), C.c_float(lmScale),
                           C.c_float(wordPen), C.c_float(prScale), C.c_int(int(maxActive)), C.c_int(maxWords), _p(wp), _p(ws), _p(we), _p(sc), C.byref(tot))
    if n < 0:
        if n == -1:
            return None, tot.value
        raise RuntimeError("orc_decode failed (%d)" % n)
    return [(int(wp[i]), int(ws[i]), int(we[i]), float(sc[i])) for i in range(n)], tot.value
