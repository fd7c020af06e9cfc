#!/usr/bin/env python3
"""GPU box: latency of ONE stereo pair per call (config 2's shape) on frames cut from the photographs, pair by pair — which frames
are slow, and what the relaxation's round statistics and the kernel profile say about them.
  python tools/single_pair_real.py [pairs] [profile pair index]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pli_slam_amd import capi, realdata, synth
from pli_slam_amd.frontend import Frontend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prof_i = int(sys.argv[2]) if len(sys.argv) > 2 else -1
W, H = 752, 480
frames = realdata.frames_752x480(n, seed=17)
dev = torch.device("cuda:0")
fe = Frontend(capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=1))
fe.set_stream(torch.cuda.current_stream().cuda_stream)
tab = torch.zeros(fe.table_bytes(1), dtype=torch.uint8, device=dev)
def run(L, R, reps=6):
    dl, dr = torch.from_numpy(np.ascontiguousarray(L)).to(dev), torch.from_numpy(np.ascontiguousarray(R)).to(dev)
    for rep in range(reps + 2):
        if rep == 2:
            torch.cuda.synchronize(); ts = time.perf_counter()
        fe.batch_run_device(1, dl.data_ptr(), dr.data_ptr(), W, W * H, tab.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - ts) / reps * 1e3
for i in range(n):
    ms = run(frames[i][0], frames[i][1])
    rec = fe.parse_record(tab.cpu().numpy(), 0)
    print("real pair %2d: %.2f ms  lines %d/%d  rounds %s" % (i, ms, len(rec["klL"]), len(rec["klR"]), list(fe.lsd_round_stats())), flush=True)
for s in range(4):
    L, R = synth.make_stereo_pair(40 + s, W, H)
    print("synthetic %d: %.2f ms  rounds %s" % (s, run(L, R), list(fe.lsd_round_stats())), flush=True)
if prof_i >= 0:
    fe.prof_enable(True); fe.prof_reset()
    run(frames[prof_i][0], frames[prof_i][1], reps=4)
    rep = fe.prof_report()
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-22s calls %4d  %.3f ms" % (k, v[0], v[1]))
