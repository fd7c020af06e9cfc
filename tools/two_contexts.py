#!/usr/bin/env python3
"""Dev tool (GPU box): throughput of K contexts driven by K host threads, each with F/K frames per step, against one context with F
(FULL=1 in the environment: every context takes F frames per step — K batches in flight, free running)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = 4
W, H = 752, 480
dev = torch.device("cuda", 0)
pairs = [synth.make_stereo_pair(s, W, H) for s in range(32)]
imgs = torch.from_numpy(np.stack([np.stack(p) for p in pairs])).to(dev)
for K in (1, 2, 4):
    full = os.environ.get("FULL") == "1"
    f = F if full else F // K
    ctxs = []
    for k in range(K):
        fe = Frontend(capi.default_config(W, H, max_frames=f), device=0)
        st = torch.cuda.Stream(device=dev)
        fe.set_stream(st.cuda_stream)
        d = imgs[torch.arange(f, device=dev) % 32].contiguous()
        l, r = d[:, 0].contiguous(), d[:, 1].contiguous()
        t = torch.zeros(f * int(fe.layout.record_bytes), dtype=torch.uint8, device=dev)
        ctxs.append((fe, st, l, r, t))
    torch.cuda.synchronize()
    def work(c, n):
        fe, st, l, r, t = c
        for _ in range(n):
            fe.batch_run_device(f, l.data_ptr(), r.data_ptr(), W, W * H, t.data_ptr())
        fe.sync()
    for c in ctxs: work(c, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, steps)) for c in ctxs]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = f * K
    print("contexts %d x %d frames: %.0f frames/s (%.1f ms per %d frames)" % (K, f, tot * steps / dt, dt / steps * 1e3 * F / tot, F), flush=True)
    del ctxs
