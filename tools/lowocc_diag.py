"""Where the Gaussians of fewer than two frames of occupancy differ from the reference's model on the headline workload (they were left out of the
comparison until round 5): per mode, how many differ, and for a sample of them the occupancy (reference accumulators and ours), the initial,
the reference's and our first mean component, the component weights.   python tools/lowocc_diag.py   (GPU box with oracle/_ref)"""
import os, sys, tempfile, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import c3_herest as c3
from htk_amd import capi as native
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import batch_arrays

s, pk = c3.workload()
with tempfile.TemporaryDirectory(prefix="lowocc_") as d:
    c3.write_files(d, s, pk)
    o1, log1, _ = c3.run_reference(d, c3.NU, 1)
    o8, log8, accs = c3.run_reference(d, c3.NU, 8)
    r1 = c3.read_model(os.path.join(o1, "MMF"), pk)
    vec = c3.load_accs(pk, accs)
lay = native.accs_layout(pk)
G = int(pk["numGauss"])
occ = vec[lay.muOcc:lay.muOcc + G]
init = pk["mean"].astype(np.float64)
utts = [dict(seq=q, feat=x) for q, x in zip(s.seqs, s.feats)]
X, frameOff, labOff, labs = batch_arrays(utts)
dX = native.DevArray(X)
for mode in (0, 6):
    model = native.Model(pk)
    fb, acc = native.ForwardBackward(model), native.Accs(model)
    fb.prepare(dX.ptr.value, frameOff, labOff, labs)
    fb.execute(native.fb_config(scoreMode=mode), acc)
    fb.results()
    a = acc.download()
    model.update_device(acc, minEgs=c3.MIN_EGS, minVar=c3.MIN_VAR)
    p = model.get_params()
    sig = np.sqrt(np.abs(r1["var"].astype(np.float64)))
    rel = np.abs(p["mean"].astype(np.float64) - r1["mean"]) / np.maximum(np.abs(r1["mean"]), sig)
    badg = np.nonzero((rel > 1e-4).any(1))[0]
    low = occ < 2.0
    print("mode", mode, "Gaussians off by > 1e-4 in a mean:", len(badg), "of them occ < 2:", int(low[badg].sum()), " occ == 0 in the reference:", int((occ[badg] == 0).sum()))
    ref_moved = (np.abs(r1["mean"].astype(np.float64) - init) > 0).any(1)
    our_moved = (np.abs(p["mean"].astype(np.float64) - init) > 0).any(1)
    print("   of those: the reference left the mean as it was in", int((~ref_moved[badg]).sum()), "; we left it in", int((~our_moved[badg]).sum()))
    for g in badg[:12]:
        st = g // c3.M
        print("   g %d state %d: occ ref %.4g ours %.4g | state occ ref %.4g | mean0 init %.5f ref %.5f ours %.5f | weight init %.4g ref %.4g ours %.4g | var0 ref %.4g ours %.4g"
              % (g, st, occ[g], a["muOcc"][g], occ[st * c3.M:(st + 1) * c3.M].sum(), init[g, 0], r1["mean"][g, 0], p["mean"][g, 0], pk["compWeight"][g], r1["compWeight"][g], p["compWeight"][g],
                 r1["var"][g, 0], p["var"][g, 0]))
