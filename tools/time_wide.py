"""K1 time of a fixed-stride all-tests batch (sizes n0 v n1): python tools/time_wide.py n0 n1 [positions]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import nanomod_amd as nm
L = nm._lib
n0, n1 = int(sys.argv[1]), int(sys.argv[2]); P = int(sys.argv[3]) if len(sys.argv) > 3 else 400000
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL)
dev = 'cuda:0'
s0 = torch.empty(P * n0, dtype=torch.float32, device=dev); s1 = torch.empty(P * n1, dtype=torch.float32, device=dev)
det.synth_fill(s0, 1, 0, P, 0, n0, 10000, 0.8); det.synth_fill(s1, 1, 0, P, 1, n1, 10000, 0.8)
rid = torch.zeros(P, dtype=torch.int32, device=dev)
for _ in range(2): det.run(s0, s1, rid, stride0=n0, stride1=n1, npos=P)
det.timer = nm.EventTimer(64)
for _ in range(5): det.run(s0, s1, rid, stride0=n0, stride1=n1, npos=P)
torch.cuda.synchronize()
k1, n = det.timer.read(L.KERNEL_RANK_STATS)
print('%s %d v %d: K1 %.3f ms, %.3g positions/s' % (os.environ.get('TAG', ''), n0, n1, k1 / n, P / (k1 / n * 1e-3)))
