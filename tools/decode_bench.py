"""Timing of network decoding at BASELINE config[3] size: 5k tied states x 16 mixtures, 6000 one-model words in a word loop
(HBuild shape, l = log(1/V)), 500-frame utterances, HVite -t 250.  Run on the GPU box: python tools/decode_bench.py [nUtt]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from htk_amd import capi, synth  # noqa: E402


def main():
    nU = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    S, M, V, T = (int(os.environ.get(k, d)) for k, d in (("DEC_STATES", 5000), ("DEC_MIX", 16), ("DEC_WORDS", 6000), ("DEC_FRAMES", 500)))
    s = synth.generate_fast(S, M, V, nU, T, seed=1234, model_seed=3)
    pk = s.packed()
    model = capi.Model(pk)
    d = tempfile.mkdtemp()
    names = ["p%d" % i for i in range(V)]
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names) if False else None
    # the network needs only name -> physical index: write a tiny stand-in HMM list + dict and reuse the packed model
    # (the Mmf object is only used by the builder for name lookup)
    t0 = time.time()
    synth.write_mmf_packed(os.path.join(d, "MMF"), pk, names)
    open(os.path.join(d, "hmmlist"), "w").write("\n".join(names) + "\n")
    open(os.path.join(d, "dict"), "w").write("".join("%s %s\n" % (n, n) for n in names))
    with open(os.path.join(d, "net.slf"), "w") as f:
        f.write("VERSION=1.0\nN=%d L=%d\nI=0 W=!NULL\nI=1 W=!NULL\n" % (V + 4, 2 * V + 3))
        for i, n in enumerate(names):
            f.write("I=%d W=%s\n" % (2 + i, n))
        f.write("I=%d W=!NULL\nI=%d W=!NULL\n" % (V + 2, V + 3))
        j = 0
        f.write("J=%d S=0 E=1 l=0.00\n" % j); j += 1
        f.write("J=%d S=%d E=1 l=0.00\n" % (j, V + 2)); j += 1
        for i in range(V):
            f.write("J=%d S=1 E=%d l=%.2f\n" % (j, 2 + i, np.log(1.0 / V))); j += 1
            f.write("J=%d S=%d E=%d l=0.00\n" % (j, 2 + i, V + 2)); j += 1
        f.write("J=%d S=%d E=%d l=0.00\n" % (j, V + 2, V + 3))
    mmf = capi.Mmf(files=[os.path.join(d, "MMF")], hmm_list=os.path.join(d, "hmmlist"))
    net = capi.Net(os.path.join(d, "net.slf"), os.path.join(d, "dict"), mmf)
    print("setup %.1f s; nodes %d links %d" % (time.time() - t0, net.desc.nNodes, net.desc.nLinks))
    dec = capi.Decoder(model, net)
    for rep in range(2):
        t0 = time.time()
        res = dec.run(s.feats, genBeam=250.0, scoreMode=int(os.environ.get("DEC_SCORE_MODE", "0")))
        dt = time.time() - t0
        ok = sum(1 for w, _ in res if w is not None)
        # word accuracy against the generating sequence (sanity: the models are well separated)
        hit = tot = 0
        for (w, _), q in zip(res, s.seqs):
            if w is None:
                continue
            rec = [net.out_syms[p] for p, _, _, _ in w]
            ref = ["p%d" % k for k in q]
            tot += len(ref); hit += sum(1 for a, b in zip(rec, ref) if a == b) if len(rec) == len(ref) else 0
        print("run %d: %d utterances (%d decoded) in %.3f s = %.1f utt/s, %.2f M frames/s; words correct %d/%d"
              % (rep, nU, ok, dt, nU / dt, nU * T / dt / 1e6, hit, tot))


if __name__ == "__main__":
    main()
