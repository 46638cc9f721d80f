"""Recognition of parameter files over a word network through the C ABI -- the data flow of
    HVite -H mmf -S scp -i out.mlf -w net.slf -t f -s lmscale -p wordpen [-m] dict hmmlist
Example with this repository's fixtures (features are stored as .npz there, so this example takes HTK parameter files):

    python examples/hvite_decode.py --mmf MMF --hmmlist hmmlist --dict dict --net net.slf --out rec.mlf data/*.mfc
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from htk_amd import capi  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mmf", action="append", required=True)
    ap.add_argument("--hmmlist", required=True)
    ap.add_argument("--dict", required=True)
    ap.add_argument("--net", required=True, help="word network in SLF (HVite -w)")
    ap.add_argument("--beam", type=float, default=1.0e10, help="HVite -t")
    ap.add_argument("--wordbeam", type=float, default=1.0e10, help="HVite -v")
    ap.add_argument("--lmscale", type=float, default=1.0, help="HVite -s")
    ap.add_argument("--wordpen", type=float, default=0.0, help="HVite -p")
    ap.add_argument("--prscale", type=float, default=1.0, help="HVite -r")
    ap.add_argument("--models", action="store_true", help="model-level output (HVite -m)")
    ap.add_argument("--out", required=True, help="output MLF (HVite -i)")
    ap.add_argument("data", nargs="+")
    args = ap.parse_args(argv)

    mmf = capi.Mmf(files=args.mmf, hmm_list=args.hmmlist)
    net = capi.Net(args.net, args.dict, mmf)
    model = capi.Model(mmf.packed())
    feats, period = [], 100000
    for f in args.data:
        X, period, kind = capi.parm_read(f)
        feats.append(X)
    dec = capi.Decoder(model, net, lmScale=args.lmscale)
    res = dec.run(feats, genBeam=args.beam, wordBeam=args.wordbeam, wordPen=args.wordpen, prScale=args.prscale)
    al = None
    if args.models:
        chains = [np.array([m for w in (words or []) for m in net.pron_models[w[0]]], np.int32) for words, _ in res]
        X = np.ascontiguousarray(np.concatenate(feats), np.float32)
        frameOff = np.concatenate([[0], np.cumsum([x.shape[0] for x in feats])]).astype(np.int32)
        labOff = np.concatenate([[0], np.cumsum([len(c) for c in chains])]).astype(np.int32)
        dX = capi.DevArray(X)
        al = capi.Viterbi(model).align(dX.ptr.value, frameOff, labOff, np.concatenate(chains) if len(chains) else np.zeros(0, np.int32), genBeam=args.beam)
    with open(args.out, "w") as out:
        out.write("#!MLF!#\n")
        for u, (f, (words, total)) in enumerate(zip(args.data, res)):
            out.write('"%s.rec"\n' % os.path.splitext(f)[0])
            if words is None:
                print("No tokens survived to final node of network: %s" % f)
            elif not args.models:
                for w, s, e, sc in words:
                    if net.out_syms[w] != "":
                        out.write("%d %d %s %f\n" % (s * period, e * period, net.out_syms[w], np.float32(sc)))
            else:
                q = 0
                for (w, s, e, sc), lm in zip(words, dec.last_lm[u]):
                    for k, m in enumerate(net.pron_models[w]):
                        line = "%d %d %s %f" % (al[u]["modStart"][q] * period, al[u]["modEnd"][q] * period, mmf.phys_names[m], np.float32(al[u]["modScore"][q]))
                        if k == 0:
                            aux = np.float32(np.float64(np.float32(np.float32(lm) * np.float32(args.lmscale))) + np.float64(np.float32(args.wordpen)))
                            line += " %s %f" % (net.word_names[w], aux)
                        out.write(line + "\n")
                        q += 1
            out.write(".\n")
    return res


if __name__ == "__main__":
    main()
