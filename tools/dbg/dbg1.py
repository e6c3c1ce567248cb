import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import nanomod_amd as nm
L = nm._lib
def say(*a):
    print(*a, file=sys.stderr, flush=True)
P, n0, n1 = 120000, 200, 200
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='fisher', tests=L.TEST_ALL)
for tdt in (torch.int16, torch.float32):
    q0 = torch.empty(P * n0, dtype=tdt, device='cuda:0'); q1 = torch.empty(P * n1, dtype=tdt, device='cuda:0')
    for outl in (0, 10):
        det.synth_fill_events(q0, 7, 0, P, 0, n_per_pos=n0, plant_period=100, plant_shift_milli=800, spread_milli=200, outlier_permille=outl)
        det.synth_fill_events(q1, 7, 0, P, 1, n_per_pos=n1, plant_period=100, plant_shift_milli=800, spread_milli=200, outlier_permille=outl)
        torch.cuda.synchronize(); say('filled', tdt, outl)
        rid = torch.zeros(P, dtype=torch.int32, device='cuda:0')
        res = det.run(q0, q1, rid, stride0=n0, stride1=n1, npos=P)
        torch.cuda.synchronize(); say('ran')
        say(det.dispatch_stats())
say('done')
