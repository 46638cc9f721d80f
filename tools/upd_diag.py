"""Device update against host update from the same accumulators: where do they differ (round-5 diagnosis of the fused kernel)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from htk_amd import capi as native
from test_gpu_parity import load_case, run_fb
case = load_case("fb_small")
pk, utts, prune = case["pk"], case["utts"], case["prune"]
mh, fb, acc, pr, st = run_fb(native, pk, utts, prune, debug=False)
md = native.Model(pk)
a = acc.download()
sh = mh.update(acc, a["vec"], minEgs=1)
accd = native.Accs(md); accd.upload_add(a["vec"])
sd = md.update_device(accd, minEgs=1)
print(sh); print(sd)
ph, pd = mh.get_params(), md.get_params()
print("G", ph["gconst"].shape, "D", ph["mean"].shape)
for k in ("mean", "var", "gconst", "compWeight", "transP"):
    d = np.asarray(ph[k]) != np.asarray(pd[k])
    print(k, int(d.sum()), "of", d.size, np.argwhere(d)[:10].tolist())
