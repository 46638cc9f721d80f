"""One embedded Baum-Welch pass over HTK files through the C ABI -- the data flow of
    HERest -C cfg -H mmf|-d dir -S scp -L labdir|-I mlf -t f [i l] -v minVar -w floor -m minEgs -M outdir hmmlist
(not a command-line clone: only what the path needs).  Example with the HTKDemo fixtures of this repository:

    python examples/herest_pass.py --hmmlist tests/golden/demo/bcplist --hmmdir tests/golden/demo/hmm1 \
        --data tests/golden/demo/train/*.mfc --labdir tests/golden/demo/labels --target-kind MFCC_E_D \
        --prune 2000 --minvar 0.05 --mixfloor 3 --outdir /tmp/hmm2
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from htk_amd import capi  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--hmmlist", required=True)
    ap.add_argument("--mmf", action="append", default=[], help="master macro file(s) (HERest -H)")
    ap.add_argument("--hmmdir", help="directory with one definition per model (HERest -d)")
    ap.add_argument("--scp", help="script file: one parameter file per line (HERest -S)")
    ap.add_argument("--data", nargs="*", default=[], help="parameter files")
    ap.add_argument("--labdir", help="directory of label files (HERest -L)")
    ap.add_argument("--mlf", help="master label file (HERest -I)")
    ap.add_argument("--target-kind", default=None, help="e.g. MFCC_E_D: qualifiers _D/_A missing in the files are computed (TARGETKIND)")
    ap.add_argument("--prune", nargs="+", type=float, default=None, help="HERest -t f [i l]")
    ap.add_argument("--minvar", type=float, default=0.0, help="HERest -v")
    ap.add_argument("--mixfloor", type=float, default=0.0, help="HERest -w (multiples of MINMIX)")
    ap.add_argument("--minegs", type=int, default=3, help="HERest -m")
    ap.add_argument("--score", choices=["exact", "mfma"], default="exact")
    ap.add_argument("--outdir", required=True, help="HERest -M")
    ap.add_argument("--binary", action="store_true", help="HERest -B")
    args = ap.parse_args(argv)

    mmf = capi.Mmf(files=args.mmf, hmm_list=args.hmmlist, hmm_dir=args.hmmdir)
    pk = mmf.packed()
    model = capi.Model(pk)
    files = list(args.data)
    if args.scp:
        files += [l.strip() for l in open(args.scp) if l.strip()]
    if not files:
        sys.exit("no data files")
    mlf = capi.Mlf(args.mlf) if args.mlf else None
    want = (args.target_kind or mmf.kind).upper().split("_")
    stat, seqs = [], []
    for f in files:
        X, period, kind = capi.parm_read(f)
        base = os.path.splitext(os.path.basename(f))[0]
        labs = mlf.find(base + ".lab") if mlf else capi.labels_read(os.path.join(args.labdir or os.path.dirname(f), base + ".lab"))
        if labs is None:
            sys.exit("no transcription for %s" % f)
        try:
            seqs.append(np.array([mmf.logical[n] for n, _, _, _ in labs], np.int32))
        except KeyError as e:
            sys.exit("%s: label %s is not in the HMM list" % (f, e))
        stat.append(X)
    have_d = bool(kind & 0o400)                                     # _D already in the files
    need_d, need_a = "D" in want[1:], "A" in want[1:]
    if need_d and not have_d:
        dX, frameOff, cols = capi.parm_add_qualifiers(stat, hasD=True, hasA=need_a)
    else:
        X = np.ascontiguousarray(np.concatenate(stat), np.float32)
        dX, cols = capi.DevArray(X), X.shape[1]
        frameOff = np.concatenate([[0], np.cumsum([x.shape[0] for x in stat])]).astype(np.int32)
    if cols != pk["vecSize"]:
        sys.exit("data have %d columns, the models %d" % (cols, pk["vecSize"]))
    labOff = np.concatenate([[0], np.cumsum([len(q) for q in seqs])]).astype(np.int32)
    fb, acc = capi.ForwardBackward(model), capi.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, np.concatenate(seqs))
    pr = {}
    if args.prune:
        t = args.prune
        pr = dict(pruneInit=t[0], pruneInc=t[1] if len(t) > 1 else 0.0, pruneLim=t[2] if len(t) > 2 else t[0])
    fb.execute(capi.fb_config(scoreMode=1 if args.score == "mfma" else 0, **pr), acc)
    logp, status = fb.results()
    for f, st in zip(files, status):
        if st != 1:
            print("WARNING [-7324]  %s - bad data or over pruning" % f)
    a = acc.download()
    stats = model.update(acc, a["vec"], minEgs=args.minegs, minVar=args.minvar, mixWeightFloor=args.mixfloor * 1.0e-5)
    os.makedirs(args.outdir, exist_ok=True)
    p = model.get_params()
    single = os.path.join(args.outdir, os.path.basename(args.mmf[0])) if args.mmf else None
    mmf.write(p, one_file=single, out_dir=None if single else args.outdir, binary=args.binary)
    if stats["nFloorVar"]:
        print("Total %d floored variance elements in %d different mixes" % (stats["nFloorVar"], stats["nFloorVarMix"]))
    print("Reestimation complete - average log prob per frame = %e" % (a["totalPr"] / max(a["totalT"], 1)))
    print("     - total frames seen          = %e" % a["totalT"])
    return a, stats


if __name__ == "__main__":
    main()
