"""The bench's own workload (BASELINE config[2] per GPU: 5 000 tied states x 16 mixtures, 6 000 models, 1 250 utterances of 500 frames,
the shard of rank 0) pushed through the REFERENCE's HERest -- one process, and eight `-p k` processes merged by `-p 0`
(HERest.c:514-557) -- so that the re-estimated model the HIP path writes in the mode bench.py measures can be held against the
reference's, entry by entry, next to the reference's own 1-process-vs-8-way difference.

Used by tests/golden/make_config3_herest_golden.py (writes the committed fixture: a seeded sample of states) and by the `-m gpu` tests
(the sample from the fixture; every entry of the set when oracle/_ref/HERest is on the box).  Test infrastructure only."""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden", "c3_herest.npz")

NS, M, NP, NU, T, D = 5000, 16, 6000, 1250, 500, 39
MIN_EGS, MIN_VAR = 3, 0.01                      # HERest -m 3 (its default) -v 0.01: bench.py's update


def workload(nu: int = NU):
    """bench.py's rank-0 shard, with the parameters at the precision a text MMF carries (what the reference reads)."""
    from htk_amd import synth
    s = synth.generate_fast(NS, M, NP, nu, T, seed=1000, model_seed=3, mmf_round=True)
    return s, s.packed()


def write_files(d: str, s, pk) -> list:
    from htk_amd import synth
    names = ["p%d" % i for i in range(NP)]
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
    with open(os.path.join(d, "hmmlist"), "w") as f:
        f.write("\n".join(names) + "\n")
    for u, (x, q) in enumerate(zip(s.feats, s.seqs)):
        synth.write_htk_param(os.path.join(d, "u%05d.mfc" % u), x, kind=9)
        with open(os.path.join(d, "u%05d.lab" % u), "w") as f:
            f.write("\n".join(names[int(h)] for h in q) + "\n")
    open(os.path.join(d, "config"), "w").write("BINARYACCFORMAT = T\n")
    return names


def _herest(d, args, scp=None, log=None):
    cmd = [os.path.join(REF, "HERest"), "-C", os.path.join(d, "config"), "-H", os.path.join(d, "MMF"), "-L", d, "-m", str(MIN_EGS), "-v", str(MIN_VAR), "-B"]
    if scp:
        cmd += ["-S", scp]
    return subprocess.Popen(cmd + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def run_reference(d: str, nu: int, ways: int):
    """ways == 1: one HERest process over all utterances; ways > 1: `ways` processes `-p k` over round-robin shards (HERest.c:543-550)
    followed by `-p 0` over their dumps.  Returns (output directory, log of the process that wrote the model, list of .acc files)."""
    out = os.path.join(d, "out%d" % ways)
    os.makedirs(out, exist_ok=True)
    hl = os.path.join(d, "hmmlist")
    if ways == 1:
        scp = os.path.join(d, "all.scp")
        open(scp, "w").write("\n".join(os.path.join(d, "u%05d.mfc" % u) for u in range(nu)) + "\n")
        p = _herest(d, ["-M", out, "-T", "1", hl], scp)
        log = p.communicate()[0]
        assert p.returncode == 0, log
        return out, log, []
    procs = []
    for k in range(ways):
        scp = os.path.join(d, "shard%d.scp" % k)
        open(scp, "w").write("\n".join(os.path.join(d, "u%05d.mfc" % u) for u in range(k, nu, ways)) + "\n")
        procs.append(_herest(d, ["-M", out, "-p", str(k + 1), hl], scp))
    for p in procs:
        lg = p.communicate()[0]
        assert p.returncode == 0, lg
    accs = [os.path.join(out, "HER%d.acc" % (k + 1)) for k in range(ways)]
    p = _herest(d, ["-M", out, "-p", "0", "-T", "1", hl] + accs)
    log = p.communicate()[0]
    assert p.returncode == 0, log
    return out, log, accs


def read_model(path: str, pk: dict) -> dict:
    """The parameters of an MMF the reference wrote, in the numbering of `pk` (tied states are matched through the models' state lists)."""
    from htk_amd import capi
    m = capi.Mmf([path], hmm_list=None)
    q = m.packed()
    nameTo = {n: h for h, n in enumerate(m.phys_names)}
    H = int(pk["numPhys"])
    stMap = np.full(int(pk["numStates"]), -1, np.int64)            # state of pk -> state of q
    for h in range(H):
        hq = nameTo["p%d" % h]
        a, b = pk["hmmState"][pk["hmmStateOff"][h]:pk["hmmStateOff"][h + 1]], q["hmmState"][q["hmmStateOff"][hq]:q["hmmStateOff"][hq + 1]]
        stMap[a] = b
    assert (stMap >= 0).all()
    Mx = int(pk["stateCompOff"][1] - pk["stateCompOff"][0])
    comp = (q["stateCompOff"][stMap][:, None] + np.arange(Mx)[None, :]).reshape(-1)          # component of q for every component of pk
    g = q["compGauss"][comp]
    out = dict(mean=q["mean"][g], var=q["var"][g], compWeight=q["compWeight"][comp], transP=q["transP"].copy())
    m.close()
    return out


def load_accs(pk: dict, files) -> np.ndarray:
    """Sum of the reference's accumulator dumps as one vector in the library's layout (LoadAccs adds, HTrain.c:1625)."""
    from htk_amd import capi
    names = ["p%d" % i for i in range(int(pk["numPhys"]))]
    lay = capi.accs_layout(pk)
    vec = np.zeros(int(lay.total), np.float64)
    for f in files:
        capi.accs_load_file(pk, vec, names, f)
    return vec


def trans_lin(logtp):
    return np.where(np.asarray(logtp) > -0.5e10, np.exp(np.asarray(logtp, np.float64)), 0.0)


def compare(got: dict, ref1: dict, ref8: dict, occ: np.ndarray, min_occ: float = 2.0, init_mean=None) -> dict:
    """Entry-by-entry comparison of re-estimated parameters with the reference's (north_star: mean / variance within 1e-4 relative).
    `got`, `ref1`, `ref8`: dicts with mean [G,D], var [G,D], compWeight [C], transP (log); occ [G] = the Gaussians' occupancies.
    The bar of an entry is 1e-4 of its scale -- |ref| for weights and transition probabilities, max(|ref|, sigma) for means
     -- widened by the reference's OWN difference between one process and an 8-way merge at that entry where that
    is larger (an entry the reference itself does not reproduce to 1e-4 cannot be asked of anybody else).
    Variances: HERest forms  var = va/occ - (mu/occ)^2  from sums about the PREVIOUS mean (HFB.c:1671-1678, HERest.c:1045-1122), so what
    carries 1e-4 is the second moment about that mean, var + (mean_new - mean_old)^2: with `init_mean` that is the scale (it equals
    the variance itself but for a Gaussian whose mean moved by more than its width -- 7 of the headline set's 2.9 M entries then sit
    between 1e-4 and 1.8e-4 of their own value in the tolerance-class mode, one does in the reference's own 1-vs-8 difference).
    `n_above_1e4` always counts against the variance itself.  Returns counts and worst ratios; the caller asserts.
    Groups: `mean` / `var` = the Gaussians of at least `min_occ` frames of occupancy; `mean_low_occ` / `var_low_occ` = the others that exist in
    the reference's model (until round 5 they were left out)."""
    r = {}
    # A component HERest's update turned off (its new weight fell to zero: MINMIX, HERest.c:1014-1057) is no longer IN the reference's model --
    # PutMixPDF writes no <MIXTURE> for it, and a reader finds default values in its place: it is compared through `weight.zeros_equal` only.
    # Every other entry is held to the bar, in two groups: the Gaussians of at least `min_occ` frames, and (round 5) the rest.
    exists = ref1["compWeight"].astype(np.float64) > 0 if ref1["compWeight"].shape[0] == occ.shape[0] else np.ones(occ.shape[0], bool)
    sel = exists & (occ >= min_occ)
    low = exists & ~(occ >= min_occ)
    s1 = np.sqrt(np.abs(ref1["var"].astype(np.float64)))
    vscale = np.abs(ref1["var"].astype(np.float64))
    if init_mean is not None:
        vscale = vscale + (ref1["mean"].astype(np.float64) - np.asarray(init_mean, np.float64)) ** 2
    for k, scale in (("mean", np.maximum(np.abs(ref1["mean"].astype(np.float64)), s1)), ("var", vscale)):
        for tag, pick in ((k, sel), (k + "_low_occ", low)):
            if not pick.any():
                continue
            e = np.abs(got[k].astype(np.float64) - ref1[k])[pick]
            self_ = np.abs(ref8[k].astype(np.float64) - ref1[k])[pick]
            sc = scale[pick]
            own = np.abs(ref1[k].astype(np.float64))[pick] if k == "var" else sc
            r[tag] = dict(n=int(e.size), worst_rel=float((e / own).max()), n_above_1e4=int((e > 1e-4 * own).sum()),
                          n_self_above_1e4=int((self_ > 1e-4 * own).sum()), self_worst_rel=float((self_ / own).max()),
                          n_above_1e4_not_self=int(((e > 1e-4 * own) & ~(self_ > 1e-4 * own)).sum()),
                          n_fail=int((e > np.maximum(1e-4 * sc, 2.0 * self_)).sum()), worst_rel_scaled=float((e / sc).max()),
                          p9999_rel=float(np.quantile(e / sc, 0.9999)), self_p9999_rel=float(np.quantile(self_ / sc, 0.9999)))
    w, w1, w8 = got["compWeight"].astype(np.float64), ref1["compWeight"].astype(np.float64), ref8["compWeight"].astype(np.float64)
    e, self_ = np.abs(w - w1), np.abs(w8 - w1)
    pos = w1 > 0
    r["weight"] = dict(n=int(e.size), worst_rel=float((e[pos] / w1[pos]).max()), n_above_1e4=int((e > 1e-4 * w1).sum()), n_self_above_1e4=int((self_ > 1e-4 * w1).sum()),
                       n_fail=int((e > np.maximum(1e-4 * w1, 2.0 * self_)).sum()), zeros_equal=bool(((w == 0) == (w1 == 0)).all()))
    t, t1, t8 = trans_lin(got["transP"]), trans_lin(ref1["transP"]), trans_lin(ref8["transP"])
    nz = t1 > 0
    r["trans"] = dict(worst_rel=float((np.abs(t - t1)[nz] / t1[nz]).max()), self_worst_rel=float((np.abs(t8 - t1)[nz] / t1[nz]).max()),
                      zeros_equal=bool(((t == 0) == (t1 == 0)).all()))
    return r
