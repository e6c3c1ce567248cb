"""K2 (finalize_kernel) time per test mask: python tools/time_finalize.py [positions] [reads]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import nanomod_amd as nm
L = nm._lib
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = 'cuda:0'
for name, tests in (('KS', L.TEST_KS), ('KS+MWU', L.TEST_KS | L.TEST_MWU), ('KS+Welch', L.TEST_KS | L.TEST_WELCH), ('all', L.TEST_ALL)):
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
    s0 = torch.empty(P * n, dtype=torch.float32, device=dev); s1 = torch.empty(P * n, dtype=torch.float32, device=dev)
    det.synth_fill(s0, 1, 0, P, 0, n, 10000, 0.8); det.synth_fill(s1, 1, 0, P, 1, n, 10000, 0.8)
    rid = torch.zeros(P, dtype=torch.int32, device=dev)
    for _ in range(2): det.run(s0, s1, rid, stride0=n, stride1=n, npos=P)
    det.timer = nm.EventTimer(64)
    for _ in range(5): det.run(s0, s1, rid, stride0=n, stride1=n, npos=P)
    torch.cuda.synchronize()
    out = []
    for kname in ('KERNEL_RANK_STATS', 'KERNEL_FINALIZE', 'KERNEL_COMBINE'):
        ms, cnt = det.timer.read(getattr(L, kname)); out.append('%s %.3f ms' % (kname[7:].lower(), ms / cnt))
    print('%-9s %s' % (name, '  '.join(out)))
