"""Debug aid: full TV-L1 calc with the current MIFLOW_TB_* environment, saved to an .npy (compare two runs with --cmp)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b).max(-1)
    from opencv_contrib_amd import synth
    dd = np.sqrt(((a - b) ** 2).sum(-1))
    print("meanEPE %.3e ccorr %.3e within0.02 %.5f" % (dd.mean(), synth.ccorr_dissimilarity(b, a), (dd <= 0.02).mean()))
    print("max", d.max(), "mean", d.mean(), "argmax (y,x)", np.unravel_index(d.argmax(), d.shape), "shape", d.shape)
    ys, xs = np.where(d > 0.25 * d.max())
    print("rows with large diff:", np.unique(ys)[:40], "cols:", np.unique(xs)[:60])
    cols = d.max(0); rows = d.max(1)
    print("col profile (max over rows) top:", np.argsort(cols)[-12:], "row profile top:", np.argsort(rows)[-12:])
    sys.exit(0)
import torch
from opencv_contrib_amd import cuda, synth
h, w, tb, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
I0, I1, _ = synth.flow_pair(h, w, seed=78)
if os.environ.get('PERTURB'):
    rng = np.random.default_rng(1)
    I0 = (I0 + float(os.environ['PERTURB']) * rng.standard_normal(I0.shape)).astype(np.float32)
alg = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0, exactMath=bool(int(os.environ.get("EXACT", "0"))), timeBlock=tb, nscales=int(os.environ.get("NSCALES", "5")), warps=int(os.environ.get("WARPS", "5")))
f = alg.calc(torch.from_numpy(I0).cuda(), torch.from_numpy(I1).cuda())
torch.cuda.synchronize()
np.save(out, f.cpu().numpy())
