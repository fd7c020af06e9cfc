#!/usr/bin/env python3
"""GPU box (dev tool): the headline batch on ONE context against the same batches alternating over 2 / 3 contexts of one device (each
with its own streams and planes): does a batch's front pass / tile sort / round 1 fill the holes of the previous batch's later rounds?
   python tools/pipeline_probe.py [F] [steps] [--real]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

args = [a for a in sys.argv[1:] if not a.startswith("--")]
F = int(args[0]) if args else 256
steps = int(args[1]) if len(args) > 1 else 12
W, H = 752, 480
dev = torch.device("cuda:0")
nuniq = min(F, 256)
if "--real" in sys.argv:
    from pli_slam_amd import realdata
    pairs = realdata.frames_752x480(nuniq, seed=17, w=W, h=H)
else:
    with ThreadPoolExecutor(32) as ex:
        pairs = list(ex.map(lambda s_: synth.make_stereo_pair(s_, W, H), range(nuniq)))
images = np.stack([np.stack(p) for p in pairs])
d_uniq = torch.from_numpy(images).to(dev)
d_img = d_uniq[torch.arange(F, device=dev) % nuniq].contiguous()
d_left, d_right = d_img[:, 0].contiguous(), d_img[:, 1].contiguous()
cfg = capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F)
ref = None
for nctx in (1, 2, 3, 1, 2):
    fes = [Frontend(cfg, dev=False) for _ in range(nctx)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nctx)]
    tables = [torch.zeros(F * int(fes[0].layout.record_bytes), dtype=torch.uint8, device=dev) for _ in range(nctx)]
    for fe, s in zip(fes, streams):
        fe.set_stream(s.cuda_stream)

    def step(i):
        k = i % nctx
        fes[k].batch_run_device(F, d_left.data_ptr(), d_right.data_ptr(), W, W * H, tables[k].data_ptr())

    for i in range(2 * nctx):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = True
    host = [t.cpu().numpy() for t in tables]
    if ref is None:
        ref = host[0].copy()
    same = all(np.array_equal(h, ref) for h in host)
    slow = [fe.lsd_round_stats() for fe in fes]
    print("contexts %d: %.1f frames/s  %.3f ms per batch   tables identical to the first run: %s   round stats %s" %
          (nctx, F * steps / dt, dt / steps * 1e3, same, slow), flush=True)
    del fes, tables
    torch.cuda.synchronize()
