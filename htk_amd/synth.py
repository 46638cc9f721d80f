"""Deterministic synthetic tied-state triphone sets + utterances.

The generator follows the random-draw ORDER of the survey's scratch generator, so the known
answers recorded there (e.g. `-6.124135e+01` for the first utterance of the 1k x 8 set, seed 1)
apply to the files written here.  It writes the files the reference tools read (text MMF, HTK
parameter files, label files, SCP, hmmlist, config, dict) and returns the same content as flat
numpy arrays in the packed layout of include/htk_amd.h, so one call feeds both the reference
(oracle/_ref) and the HIP path.

Nothing here is on the product path: it is workload/test tooling.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field

import numpy as np

LZERO = -1.0e10
PK_MFCC_E_D_A = 6 | 0o100 | 0o400 | 0o1000  # HParm.h parameter-kind bits: MFCC + _E + _D + _A

# transition matrix of the synthetic sets; linear probabilities as in an MMF
TRANSP_5 = np.array(
    [[0, 1, 0, 0, 0],
     [0, 0.6, 0.4, 0, 0],
     [0, 0, 0.6, 0.4, 0],
     [0, 0, 0, 0.7, 0.3],
     [0, 0, 0, 0, 0]], dtype=np.float32)


@dataclass
class SynthSet:
    D: int
    NS: int          # tied states
    M: int           # mixtures per state
    NP: int          # physical models
    means: np.ndarray   # [NS, M, D] f32
    var: np.ndarray     # [NS, M, D] f32
    w: np.ndarray       # [NS, M] f32 (linear)
    st: np.ndarray      # [NP, 3] tied-state index of emitting states 2..4
    seqs: list = field(default_factory=list)    # per utterance: int array of model indices
    feats: list = field(default_factory=list)   # per utterance: [T, D] f32
    outdir: str | None = None

    # ---- packed layout (include/htk_amd.h: htkamd_model_desc) ----
    def packed(self) -> dict:
        G = self.NS * self.M
        tp = TRANSP_5.astype(np.float64)
        with np.errstate(divide="ignore"):
            logt = np.where(tp > 0, np.log(tp), LZERO).astype(np.float32)  # HModel.c:2003 GetTransMat
        return dict(
            vecSize=self.D,
            numStates=self.NS,
            numComp=G,
            numGauss=G,
            stateCompOff=(np.arange(self.NS + 1) * self.M).astype(np.int32),
            compWeight=self.w.reshape(G).astype(np.float32),
            compGauss=np.arange(G, dtype=np.int32),
            mean=np.ascontiguousarray(self.means.reshape(G, self.D)),
            var=np.ascontiguousarray(self.var.reshape(G, self.D)),
            gconst=None,                      # no <GCONST> in the file: library applies FixDiagGConst
            numTrans=1,
            transN=np.array([5], np.int32),
            transOff=np.array([0, 25], np.int32),
            transP=logt.reshape(-1),
            numPhys=self.NP,
            hmmTrans=np.zeros(self.NP, np.int32),
            hmmStateOff=(np.arange(self.NP + 1) * 3).astype(np.int32),
            hmmState=self.st.reshape(-1).astype(np.int32),
        )


def _vec(v) -> str:
    return " " + " ".join("%e" % x for x in v)


def write_mmf(path: str, s: SynthSet, kind: str = "MFCC_E_D_A") -> None:
    D, M = s.D, s.M
    with open(path, "w") as f:
        f.write("~o\n<STREAMINFO> 1 %d\n<VECSIZE> %d<NULLD><%s><DIAGC>\n" % (D, D, kind))
        f.write('~t "T1"\n<TRANSP> 5\n 0 1 0 0 0\n 0 0.6 0.4 0 0\n 0 0 0.6 0.4 0\n 0 0 0 0.7 0.3\n 0 0 0 0 0\n')
        for i in range(s.NS):
            f.write('~s "S%d"\n' % i)
            if M > 1:
                f.write("<NUMMIXES> %d\n" % M)
            for m in range(M):
                if M > 1:
                    f.write("<MIXTURE> %d %e\n" % (m + 1, s.w[i, m]))
                f.write("<MEAN> %d\n%s\n<VARIANCE> %d\n%s\n" % (D, _vec(s.means[i, m]), D, _vec(s.var[i, m])))
        for p in range(s.NP):
            f.write('~h "p%d"\n<BEGINHMM>\n<NUMSTATES> 5\n' % p)
            for j in range(3):
                f.write('<STATE> %d\n~s "S%d"\n' % (j + 2, s.st[p, j]))
            f.write('~t "T1"\n<ENDHMM>\n')


def write_htk_param(path: str, X: np.ndarray, samp_period: int = 100000, kind: int = PK_MFCC_E_D_A) -> None:
    """HTK parameter file: 12-byte big-endian header + BE float32 rows (HTKBook speechio.tex:835-950)."""
    T, D = X.shape
    with open(path, "wb") as f:
        f.write(struct.pack(">iihh", T, samp_period, D * 4, kind))
        f.write(np.ascontiguousarray(X, dtype=">f4").tobytes())


def read_htk_param(path: str):
    with open(path, "rb") as f:
        n, period, size, kind = struct.unpack(">iihh", f.read(12))
        X = np.frombuffer(f.read(n * size), dtype=">f4").reshape(n, size // 4).astype(np.float32)
    return X, period, kind


def round_to_mmf_precision(a: np.ndarray) -> np.ndarray:
    """The text MMF carries '%e' (7 significant digits); the reference therefore sees
    float(strtod('%e' % x)), not x.  Apply the same round trip so the packed arrays equal
    what LoadHMMSet produces bit for bit."""
    flat = a.reshape(-1)
    out = np.array([float("%e" % x) for x in flat], dtype=np.float64).astype(np.float32)
    return out.reshape(a.shape)


def generate(NS: int, M: int, NP: int, NU: int, T: int, seed: int, D: int = 39,
             outdir: str | None = None, write_data: bool = True) -> SynthSet:
    """SURVEY.md Appendix F generator (same draw order).  If outdir is given the reference-format
    files are written there: hmm0/MMF, hmmlist, data/u%05d.mfc, lab/u%05d.lab, train.scp, config, dict."""
    rng = np.random.default_rng(seed)
    means = rng.normal(0, 3, size=(NS, M, D)).astype(np.float32)
    var = rng.uniform(0.5, 2.0, size=(NS, M, D)).astype(np.float32)
    w = rng.dirichlet(np.ones(M) * 5, size=NS).astype(np.float32)
    st = rng.integers(0, NS, size=(NP, 3))
    flat = st.reshape(-1)
    k = min(NS, flat.size)
    flat[:k] = rng.permutation(NS)[:k]          # every tied state used at least once
    st = flat.reshape(NP, 3)

    # what the reference will actually hold after reading the text MMF
    s = SynthSet(D=D, NS=NS, M=M, NP=NP,
                 means=round_to_mmf_precision(means), var=round_to_mmf_precision(var),
                 w=round_to_mmf_precision(w), st=st.astype(np.int32), outdir=outdir)
    if outdir:
        for d in ("data", "lab", "hmm0", "hmm1"):
            os.makedirs(os.path.join(outdir, d), exist_ok=True)
        raw = SynthSet(D=D, NS=NS, M=M, NP=NP, means=means, var=var, w=w, st=st)
        write_mmf(os.path.join(outdir, "hmm0", "MMF"), raw)
        with open(os.path.join(outdir, "hmmlist"), "w") as f:
            for p in range(NP):
                f.write("p%d\n" % p)
        with open(os.path.join(outdir, "dict"), "w") as f:
            for p in range(NP):
                f.write("p%d p%d\n" % (p, p))
        with open(os.path.join(outdir, "config"), "w") as f:
            f.write("TARGETKIND = MFCC_E_D_A\nBINARYACCFORMAT = T\n")
        scp = open(os.path.join(outdir, "train.scp"), "w")

    for u in range(NU):
        Q = max(1, T // 12)
        seq = rng.integers(0, NP, size=Q)
        fr = []
        per = T // Q
        for q, p in enumerate(seq):
            n = per if q < Q - 1 else T - per * (Q - 1)
            for j in range(3):
                nj = n // 3 if j < 2 else n - 2 * (n // 3)
                si = st[p, j]
                ms = rng.integers(0, M, size=nj)
                # frames are sampled from the UNROUNDED parameters, exactly as the survey generator did
                fr.append(means[si, ms] + rng.normal(0, 1, size=(nj, D)).astype(np.float32) * np.sqrt(var[si, ms]))
        X = np.concatenate(fr).astype(np.float32)
        assert X.shape[0] == T
        s.seqs.append(seq.astype(np.int32))
        s.feats.append(X)
        if outdir:
            fn = "%s/data/u%05d.mfc" % (outdir, u)
            if write_data:
                write_htk_param(fn, X)
                with open("%s/lab/u%05d.lab" % (outdir, u), "w") as f:
                    for p in seq:
                        f.write("p%d\n" % p)
            scp.write(fn + "\n")
    if outdir:
        scp.close()
    return s


def round_to_mmf_precision_bulk(a: np.ndarray) -> np.ndarray:
    """round_to_mmf_precision for millions of values: one format call, one parse."""
    flat = np.asarray(a, np.float32).reshape(-1)
    out = np.empty(flat.size, np.float32)
    for i in range(0, flat.size, 1 << 20):
        blk = flat[i:i + (1 << 20)]
        out[i:i + blk.size] = np.array((("%e " * blk.size) % tuple(blk.tolist())).split(), dtype=np.float64).astype(np.float32)
    return out.reshape(np.shape(a))


def generate_fast(NS: int, M: int, NP: int, NU: int, T: int, seed: int, D: int = 39, model_seed: int | None = None, mmf_round: bool = False,
                  utt_ids=None) -> SynthSet:
    """Vectorised variant for large workloads (bench.py): same model distribution and utterance structure as
    generate() (Q = T//12 models per utterance, equal thirds per state, frames drawn from the aligned state's
    GMM) but bulk random draws, no files and no MMF-precision round trip.  `model_seed` fixes the model
    independently of the utterance seed so that every rank of a sharded run holds the same HMM set.
    `mmf_round`: the set returned holds the parameters a text MMF of it carries ('%e'), as generate() does (the frames are drawn
    from the unrounded ones either way, so the data do not depend on the switch).
    `utt_ids`: generate exactly these utterances of the job `seed` (NU is ignored): utterance u has its own random stream, so that
    any split of a job over ranks yields the same utterances (bench.py --scaling strong: rank r of R takes r, r+R, ...)."""
    mrng = np.random.default_rng(seed if model_seed is None else model_seed)
    means = mrng.normal(0, 3, size=(NS, M, D)).astype(np.float32)
    var = mrng.uniform(0.5, 2.0, size=(NS, M, D)).astype(np.float32)
    w = mrng.dirichlet(np.ones(M) * 5, size=NS).astype(np.float32)
    st = mrng.integers(0, NS, size=(NP, 3))
    flat = st.reshape(-1)
    k = min(NS, flat.size)
    flat[:k] = mrng.permutation(NS)[:k]
    st = flat.reshape(NP, 3).astype(np.int32)
    s = SynthSet(D=D, NS=NS, M=M, NP=NP, means=means, var=var, w=w, st=st)
    rng = np.random.default_rng([seed, 1])
    Q = max(1, T // 12)
    per = T // Q
    # frame -> (model position, state) map shared by all utterances
    qpos, spos = [], []
    for q in range(Q):
        n = per if q < Q - 1 else T - per * (Q - 1)
        for j in range(3):
            nj = n // 3 if j < 2 else n - 2 * (n // 3)
            qpos += [q] * nj
            spos += [j] * nj
    qpos = np.array(qpos); spos = np.array(spos)
    sd = np.sqrt(var)
    if utt_ids is not None:
        for u in utt_ids:
            ru = np.random.default_rng([seed, 2, int(u)])
            seq = ru.integers(0, NP, size=Q)
            states = st[seq[qpos], spos]
            ms = ru.integers(0, M, size=T)
            X = means[states, ms] + ru.standard_normal((T, D), dtype=np.float32) * sd[states, ms]
            s.seqs.append(seq.astype(np.int32))
            s.feats.append(X.astype(np.float32))
        NU = 0
    seqs = rng.integers(0, NP, size=(NU, Q))
    for u in range(NU):
        states = st[seqs[u][qpos], spos]                      # [T]
        ms = rng.integers(0, M, size=T)
        X = means[states, ms] + rng.standard_normal((T, D), dtype=np.float32) * sd[states, ms]
        s.seqs.append(seqs[u].astype(np.int32))
        s.feats.append(X.astype(np.float32))
    if mmf_round:
        s.means = round_to_mmf_precision_bulk(means); s.var = round_to_mmf_precision_bulk(var); s.w = round_to_mmf_precision_bulk(w)
    return s


# --------------------------------------------------------------------------------------------
# Arbitrary packed model -> text MMF, and a small set with mixed topologies (tee model, 3/4/5-state
# models, 1..3 mixture components) for the parity tests of the general code paths.
# --------------------------------------------------------------------------------------------

def write_mmf_packed(path: str, pk: dict, hmm_names: list, kind: str = "USER") -> None:
    """Text MMF with ~t "T<i>", ~s "S<i>" and ~h macros for any packed model (no shared Gaussians)."""
    D = int(pk["vecSize"])
    with open(path, "w") as f:
        f.write("~o\n<STREAMINFO> 1 %d\n<VECSIZE> %d<NULLD><%s><DIAGC>\n" % (D, D, kind))
        for t in range(int(pk["numTrans"])):
            N = int(pk["transN"][t])
            tp = np.asarray(pk["transP"][pk["transOff"][t]:pk["transOff"][t + 1]], np.float64).reshape(N, N)
            lin = np.where(tp > -0.5e10, np.exp(tp), 0.0)
            f.write('~t "T%d"\n<TRANSP> %d\n' % (t, N))
            for i in range(N):
                f.write(" " + " ".join("%e" % v for v in lin[i]) + "\n")
        for s in range(int(pk["numStates"])):
            c0, c1 = int(pk["stateCompOff"][s]), int(pk["stateCompOff"][s + 1])
            f.write('~s "S%d"\n' % s)
            if c1 - c0 > 1:
                f.write("<NUMMIXES> %d\n" % (c1 - c0))
            for c in range(c0, c1):
                g = int(pk["compGauss"][c])
                if c1 - c0 > 1:
                    f.write("<MIXTURE> %d %e\n" % (c - c0 + 1, pk["compWeight"][c]))
                f.write("<MEAN> %d\n%s\n<VARIANCE> %d\n%s\n" % (D, _vec(pk["mean"][g]), D, _vec(pk["var"][g])))
        for h, name in enumerate(hmm_names):
            t = int(pk["hmmTrans"][h]); N = int(pk["transN"][t])
            f.write('~h "%s"\n<BEGINHMM>\n<NUMSTATES> %d\n' % (name, N))
            for j in range(N - 2):
                f.write('<STATE> %d\n~s "S%d"\n' % (j + 2, pk["hmmState"][pk["hmmStateOff"][h] + j]))
            f.write('~t "T%d"\n<ENDHMM>\n' % t)


TOPO_NAMES = ["a", "b", "sp", "c", "d", "e"]


def make_topo_set(seed: int = 21, D: int = 13, NU: int = 5, outdir: str | None = None):
    """Packed model with: two 5-state models, a 3-state TEE model "sp" (a_13 > 0), a 4-state model with a skip, a plain
    3-state model and a second 5-state model sharing a tied state; states with 1, 2 and 3 mixture components.
    Returns (pk, names, seqs, feats).  Parameters are rounded to the MMF's '%e' precision so that the packed arrays
    equal what the reference loads from the file written to `outdir`."""
    rng = np.random.default_rng(seed)
    lin = [np.array([[0, 1, 0, 0, 0], [0, .6, .4, 0, 0], [0, 0, .5, .5, 0], [0, 0, 0, .7, .3], [0, 0, 0, 0, 0]]),
           np.array([[0, .7, .3], [0, .6, .4], [0, 0, 0]]),                                   # tee
           np.array([[0, 1, 0, 0], [0, .5, .3, .2], [0, 0, .6, .4], [0, 0, 0, 0]]),          # skip 2 -> 4
           np.array([[0, 1, 0], [0, .8, .2], [0, 0, 0]])]
    transN = np.array([m.shape[0] for m in lin], np.int32)
    transOff = np.concatenate([[0], np.cumsum(transN ** 2)]).astype(np.int32)
    with np.errstate(divide="ignore"):
        # GetTransMat (HModel.c:2003): the file value is read into a float, the log is taken in double
        transP = np.concatenate([np.where(m.reshape(-1) > 0,
                                          np.log(np.array([float("%e" % v) for v in m.reshape(-1)], np.float32).astype(np.float64)), LZERO)
                                 for m in lin]).astype(np.float32)
    nmix = [1, 2, 3, 2, 1, 3, 2, 2, 1, 3, 2]                    # 11 tied states
    S = len(nmix); G = sum(nmix)
    stateCompOff = np.concatenate([[0], np.cumsum(nmix)]).astype(np.int32)
    mean = round_to_mmf_precision(rng.normal(0, 2.0, size=(G, D)).astype(np.float32))
    var = round_to_mmf_precision(rng.uniform(0.5, 2.0, size=(G, D)).astype(np.float32))
    w = np.concatenate([rng.dirichlet(np.ones(m) * 4) for m in nmix]).astype(np.float32)
    w = round_to_mmf_precision(w)
    #            a(T0)      b(T0)      sp(T1) c(T2)   d(T3)  e(T0, shares state 1 with a)
    hmmTrans = np.array([0, 0, 1, 2, 3, 0], np.int32)
    hmmStates = [[0, 1, 2], [3, 4, 5], [6], [7, 8], [9], [10, 1, 4]]
    hmmStateOff = np.concatenate([[0], np.cumsum([len(x) for x in hmmStates])]).astype(np.int32)
    pk = dict(vecSize=D, numStates=S, numComp=G, numGauss=G, stateCompOff=stateCompOff, compWeight=w,
              compGauss=np.arange(G, dtype=np.int32), mean=mean, var=var, gconst=None,
              numTrans=len(lin), transN=transN, transOff=transOff, transP=transP, numPhys=len(hmmStates),
              hmmTrans=hmmTrans, hmmStateOff=hmmStateOff, hmmState=np.concatenate(hmmStates).astype(np.int32))
    seqs, feats = [], []
    for u in range(NU):
        L = int(rng.integers(4, 9))
        seq = []
        for k in range(L):
            h = int(rng.choice([0, 1, 3, 4, 5]))
            seq.append(h)
            if k < L - 1 and rng.random() < 0.5:
                seq.append(2)                              # tee model, never first/last, never twice in a row
        fr = []
        for h in seq:
            for s in hmmStates[h]:
                n = int(rng.integers(0 if h == 2 else 2, 5))   # the tee model may be skipped entirely
                for _ in range(n):
                    c = int(rng.integers(stateCompOff[s], stateCompOff[s + 1]))
                    fr.append(mean[c] + rng.normal(0, 1, D) * np.sqrt(var[c]))
        X = np.array(fr, np.float32)
        seqs.append(np.array(seq, np.int32)); feats.append(X)
    if outdir:
        for d in ("data", "lab", "hmm0", "hmm1"):
            os.makedirs(os.path.join(outdir, d), exist_ok=True)
        write_mmf_packed(os.path.join(outdir, "hmm0", "MMF"), pk, TOPO_NAMES)
        with open(os.path.join(outdir, "hmmlist"), "w") as f:
            f.write("\n".join(TOPO_NAMES) + "\n")
        with open(os.path.join(outdir, "config"), "w") as f:
            f.write("BINARYACCFORMAT = T\n")
        with open(os.path.join(outdir, "train.scp"), "w") as scp:
            for u in range(NU):
                fn = "%s/data/u%05d.mfc" % (outdir, u)
                write_htk_param(fn, feats[u], kind=9)      # USER
                with open("%s/lab/u%05d.lab" % (outdir, u), "w") as f:
                    f.write("\n".join(TOPO_NAMES[h] for h in seqs[u]) + "\n")
                scp.write(fn + "\n")
    return pk, list(TOPO_NAMES), seqs, feats


def write_bigram_slf(path: str, names, seed: int = 7, n_succ: int = 5):
    """A back-off bigram word network over `names` in SLF form, the shape ProcessBoBiGram gives (HTKTools/HBuild.c:368-461): every word has
    `n_succ` explicit successors with bigram scores and an arc to the back-off !NULL node, which reaches every word with its unigram score.
    Nodes: 0 start (!NULL), 1 back-off (!NULL), 2..V+1 words, V+2 end (!NULL).  (Same draw order as tests/golden/make_config3_golden.py.)"""
    V = len(names)
    rng = np.random.default_rng(seed)
    uni = rng.dirichlet(np.ones(V) * 2.0)
    arcs = [(0, 1, 0.0)]
    for w in range(V):
        arcs.append((1, 2 + w, float(np.log(uni[w]))))
        succ = rng.choice(V, size=n_succ, replace=False)
        p = rng.dirichlet(np.ones(n_succ + 1))
        for k, s_ in enumerate(succ):
            arcs.append((2 + w, 2 + int(s_), float(np.log(0.8 * p[k]))))
        arcs.append((2 + w, 1, float(np.log(0.8 * p[n_succ]))))        # back-off weight
        arcs.append((2 + w, V + 2, float(np.log(0.2))))
    with open(path, "w") as f:
        f.write("VERSION=1.0\nN=%d L=%d\nI=0 W=!NULL\nI=1 W=!NULL\n" % (V + 3, len(arcs)))
        for i, n in enumerate(names):
            f.write("I=%d W=%s\n" % (2 + i, n))
        f.write("I=%d W=!NULL\n" % (V + 2))
        for j, (a, b, l) in enumerate(arcs):
            f.write("J=%d S=%d E=%d l=%.3f\n" % (j, a, b, l))
    return len(arcs)


def write_wav(path: str, pcm: np.ndarray, rate: int = 16000):
    """16-bit mono RIFF/WAVE file (what SOURCEFORMAT = WAV reads, HWave.c:1393)."""
    import struct
    pcm = np.ascontiguousarray(pcm, "<i2")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, 2 * rate, 2, 16) + b"data" + struct.pack("<I", pcm.nbytes))
        f.write(pcm.tobytes())
