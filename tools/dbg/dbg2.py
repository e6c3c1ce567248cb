import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, numpy as np
import nanomod_amd as nm, helpers as H
L = nm._lib
P = 4096
for n0, n1 in ((200, 200), (256, 256), (200, 150), (300, 300)):
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='fisher', tests=L.TEST_ALL)
    q0 = torch.empty(P * n0, dtype=torch.int16, device='cuda:0'); q1 = torch.empty(P * n1, dtype=torch.int16, device='cuda:0')
    det.synth_fill_events(q0, 7, 0, P, 0, n_per_pos=n0, spread_milli=200); det.synth_fill_events(q1, 7, 0, P, 1, n_per_pos=n1, spread_milli=200)
    rid = torch.zeros(P, dtype=torch.int32, device='cuda:0')
    det.timer = nm.EventTimer(16)
    res = det.run(q0, q1, rid, stride0=n0, stride1=n1, npos=P); torch.cuda.synchronize()
    print(n0, n1, det.dispatch_stats(), det.timer.read(L.KERNEL_RANK_STATS), flush=True)
