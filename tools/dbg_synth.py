import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np, torch
import nanomod_amd as nm
import helpers as H
L = nm._lib
rng = np.random.default_rng(3)
P, begin = 700, 9_990
sizes = rng.integers(0, 300, P); sizes[5] = 0; sizes[17] = 1
off = np.zeros(P + 1, np.int64); off[1:] = np.cumsum(sizes)
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
d_off = torch.from_numpy(off).cuda(); nmax = int(sizes.max())
out = torch.zeros(int(off[-1]), dtype=torch.float32, device='cuda:0')
det.synth_fill_csr(out, 77, begin, d_off, 1, 10000, 0.8); torch.cuda.synchronize()
ref = H.synth_ref(77, begin, P, 1, nmax, 10000, 0.8).reshape(P, nmax)
exp = np.concatenate([ref[i, :sizes[i]] for i in range(P)])
g = out.cpu().numpy()
bad = np.flatnonzero(g != exp)
print(len(bad), bad[:10], g[bad[:5]], exp[bad[:5]], np.searchsorted(off, bad[:10], side='right') - 1)
