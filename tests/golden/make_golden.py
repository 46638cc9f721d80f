#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (oracle/_ref, built from
/root/reference by oracle/Makefile) on seeded synthetic inputs.  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

Each .npz holds the inputs (packed model, label sequences, features) and what the reference produced:
per-utterance log-probability, beams, beta/alpha/output-probability/occupation trellises (from the
ref_fbdump harness, which drives the reference's own HFB.c), the accumulator file (HER1.acc) and the
re-estimated model (HERest single pass, text MMF).  The fixtures are data; no reference source is stored.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from htk_amd import synth          # noqa: E402
from oracle import refio           # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.dirname(os.path.abspath(__file__))


def run(cmd, cwd):
    r = subprocess.run(cmd, shell=True, cwd=cwd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s\n%s\n%s" % (cmd, r.stdout[-2000:], r.stderr[-2000:]))
    return r.stdout


def pack_case(name, pk, hmm_names, state_names, seqs, feats, d, tflag="", herest_extra="", keep_trellis=2):
    scp = open(os.path.join(d, "train.scp")).read().split()
    run("%s/ref_fbdump -C config %s -H hmm0/MMF -L lab -p 1 -M hmm1 hmmlist dump.bin %s" % (REF, tflag, " ".join(scp)), d)
    ref = refio.read_fbdump(os.path.join(d, "dump.bin"))
    acc = refio.read_acc(os.path.join(d, "hmm1", "HER1.acc"), pk, hmm_names)
    log = run("%s/HERest -C config -T 1 %s %s -H hmm0/MMF -M hmm1 -L lab -S train.scp hmmlist" % (REF, tflag, herest_extra), d)
    upd = refio.read_mmf_text(os.path.join(d, "hmm1", "MMF"), hmm_names, state_names)
    out = {}
    for k, v in pk.items():
        if v is not None:
            out["pk_" + k] = np.asarray(v)
    out["nUtt"] = np.int32(len(seqs))
    for u, (q, x) in enumerate(zip(seqs, feats)):
        out["seq_%d" % u] = np.asarray(q, np.int32)
        out["feat_%d" % u] = np.asarray(x, np.float32)
        r = ref[u]
        out["ok_%d" % u] = np.int32(r["ok"])
        if r["ok"]:
            out["pr_%d" % u] = np.float64(r["pr"])
            for k in ("qLo", "qHi", "aLo", "aHi"):
                out["%s_%d" % (k, u)] = r[k].astype(np.int16)
            if u < keep_trellis:
                for k in ("beta", "alpha", "outp", "occ"):
                    out["%s_%d" % (k, u)] = r[k]
    for k in ("mu", "muOcc", "va", "vaOcc", "wt", "wtOcc", "tr", "trOcc", "nEgs"):
        out["acc_" + k] = acc[k]
    out["acc_totalPr"] = np.float32(acc["totalPr"]); out["acc_totalT"] = np.int32(acc["totalT"])
    for k in ("mean", "var", "compWeight", "gconst"):
        out["upd_" + k] = upd[k]
    out["upd_transLin"] = np.concatenate(upd["transLin"])
    out["herest_log"] = np.array(log)
    out["tflag"] = np.array(tflag)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "ok:", [int(ref[u]["ok"]) for u in range(len(seqs))], "%.1f kB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1e3))


def main():
    if not os.path.exists(os.path.join(REF, "ref_fbdump")):
        sys.exit("oracle/_ref is missing: run `make -C oracle` first (needs /root/reference)")
    with tempfile.TemporaryDirectory() as d:
        s = synth.generate(60, 4, 40, 4, 120, 5, outdir=d)
        names = ["p%d" % i for i in range(40)]; snames = ["S%d" % i for i in range(60)]
        pack_case("fb_small", s.packed(), names, snames, s.seqs, s.feats, d, herest_extra="-m 1")
        pack_case("fb_small_prune", s.packed(), names, snames, s.seqs, s.feats, d, tflag="-t 30.0", herest_extra="-m 1")
    with tempfile.TemporaryDirectory() as d:
        pk, names, seqs, feats = synth.make_topo_set(outdir=d)
        snames = ["S%d" % i for i in range(int(pk["numStates"]))]
        pack_case("fb_topo", pk, names, snames, seqs, feats, d, herest_extra="-m 1", keep_trellis=5)
        pack_case("fb_topo_prune", pk, names, snames, seqs, feats, d, tflag="-t 20.0 10.0 100.0", herest_extra="-m 1", keep_trellis=5)
    # HVite -a -f -m alignments (.rec label files): plain set, with a beam, mixed topologies (tee model left out of the
    # labels: the reference's LatFromPaths aborts on a skipped tee model under PHNALG, HRec.c:1625)
    rec = {}
    with tempfile.TemporaryDirectory() as d:
        s = synth.generate(60, 4, 40, 4, 120, 5, outdir=d)
        for tag, tflag in (("small", ""), ("small_t40", "-t 40.0")):
            os.makedirs(os.path.join(d, "ali_" + tag))
            run("%s/HVite -C config -a -f -m %s -H hmm0/MMF -L lab -X lab -l ali_%s -y rec -S train.scp dict hmmlist" % (REF, tflag, tag), d)
            for u in range(4):
                rec["%s_%d" % (tag, u)] = np.array(open(os.path.join(d, "ali_" + tag, "u%05d.rec" % u)).read())
    with tempfile.TemporaryDirectory() as d:
        pk, names, seqs, feats = synth.make_topo_set(outdir=d)
        os.makedirs(os.path.join(d, "lab2")); os.makedirs(os.path.join(d, "ali"))
        for u, q in enumerate(seqs):
            open(os.path.join(d, "lab2", "u%05d.lab" % u), "w").write("\n".join(names[h] for h in q if h != 2) + "\n")
        open(os.path.join(d, "dict"), "w").write("".join("%s %s\n" % (n, n) for n in sorted(names)))
        run("%s/HVite -C config -a -f -m -H hmm0/MMF -L lab2 -X lab -l ali -y rec -S train.scp dict hmmlist" % REF, d)
        for u in range(len(seqs)):
            rec["topo_%d" % u] = np.array(open(os.path.join(d, "ali", "u%05d.rec" % u)).read())
    np.savez_compressed(os.path.join(OUT, "hvite_rec.npz"), **rec)
    print("hvite_rec", len(rec), "files")
    # known answers recorded in SURVEY.md Appendix F for the 1k x 8 set (seed 1): per-frame log prob of the first 3 utterances
    with tempfile.TemporaryDirectory() as d:
        s = synth.generate(1000, 8, 2000, 3, 500, 1, outdir=d)
        log = run("%s/HERest -C config -T 1 -H hmm0/MMF -M hmm1 -L lab -S train.scp hmmlist" % REF, d)
        vals = [float(l.split("=")[1]) for l in log.splitlines() if "Utterance prob per frame" in l]
        np.savez_compressed(os.path.join(OUT, "c2_known.npz"), per_frame=np.array(vals))
        print("c2_known", vals)


if __name__ == "__main__":
    main()
