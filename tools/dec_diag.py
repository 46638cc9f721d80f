"""The decode leg of bench.py alone (GPU box): python tools/dec_diag.py [n_decode]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from htk_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = synth.generate_fast(5000, 16, 6000, max(n, 8), 500, 3)
pk = s.packed()
X = np.concatenate(s.feats).astype(np.float32)
fo = np.concatenate([[0], np.cumsum([f.shape[0] for f in s.feats])]).astype(np.int64)
dX = torch.from_numpy(X).cuda()
out = bench.other_paths(s, pk, dX, fo, n_align=8, n_decode=n, cpu_utts=0)
d = out["hvite_decoding"]
print(json.dumps({k: d[k] for k in ("utterances", "ms", "words_correct", "exact_order_utterances", "model_instance_steps")}), d["roofline"]["ms"], d["score_roofline"]["ms"])
print("tolerance-class scores:", json.dumps(d.get("tolerance_class_scores")))
