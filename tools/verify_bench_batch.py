#!/usr/bin/env python3
"""Dev tool (GPU box): the bench workload itself (F frames of 752x480, U distinct pairs cycled, device buffers, default
configuration) checked for correctness: every duplicate record byte-identical to its first copy, the first copy of a few
pairs equal to the oracle.   python tools/verify_bench_batch.py [F] [U]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pyoracle as po
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
U = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 752, 480
dev = torch.device("cuda:0")
pairs = [synth.make_stereo_pair(s, W, H) for s in range(U)]
uniq = torch.from_numpy(np.stack([np.stack(p) for p in pairs])).to(dev)
img = uniq[torch.arange(F, device=dev) % U].contiguous()
left, right = img[:, 0].contiguous(), img[:, 1].contiguous()
cfg = capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F)
fe = Frontend(cfg)
fe.set_stream(torch.cuda.current_stream().cuda_stream)
rb = int(fe.layout.record_bytes)
table = torch.zeros(F * rb, dtype=torch.uint8, device=dev)
fe.batch_run_device(F, left.data_ptr(), right.data_ptr(), W, W * H, table.data_ptr())
torch.cuda.synchronize()
recs = table.view(F, rb)
dup_bad = int((recs != recs[torch.arange(F, device=dev) % U]).any(dim=1).sum().item())
print("F=%d: %d records differ from the first copy of their pair" % (F, dup_bad))
host = recs[:U].cpu().numpy().reshape(-1)
bad = 0
for i in (0, 1, U // 2, U - 1):
    r = fe.parse_record(host, i)
    fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
    L, R = pairs[i]
    ok = True
    for eye, im, k in ((0, L, "L"), (1, R, "R")):
        n, kp, desc = fr.orb_extract(eye, im)
        ok &= n == len(r["kp" + k]) and kp.tobytes() == r["kp" + k].tobytes() and np.array_equal(desc, r["desc" + k])
        m, kl, ld = fr.line_extract(eye, im)
        ok &= m == len(r["kl" + k]) and kl.tobytes() == r["kl" + k].tobytes() and np.array_equal(ld, r["ldesc" + k])
    ur, dp, _, _ = fr.stereo_points()
    ok &= ur.tobytes() == r["uright"].tobytes() and dp.tobytes() == r["depth"].tobytes()
    disp, le, _ = fr.stereo_lines()
    ok &= disp.tobytes() == r["disp"].tobytes()
    print("pair %d vs oracle: %s" % (i, "OK" if ok else "MISMATCH"))
    bad += not ok
print("RESULT", "OK" if dup_bad == 0 and bad == 0 else "FAILED")
