#!/usr/bin/env python3
"""Round-count robustness of the LSD relaxation: batches of 256 DISTINCT seeded stereo pairs, and the hostile
images of tests/test_gpu_parity.py::test_lsd_hostile_images at 752x480, through `lsd_mode` auto at F = 256.  Per batch: the rounds
the slowest image needed, the rounds launched without a host look, images that took the device-side fallback, ms per step.
  python tools/rounds_sweep.py [nbatches] > profiles/rNN_rounds_sweep.json
  python tools/rounds_sweep.py --real [nbatches] > profiles/rNN_rounds_real.json     batches of 256 DISTINCT 752x480 windows of the real
                                   photographs (pli_slam_amd/realdata.py), and every photograph alone (one window of it filling a batch)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from concurrent.futures import ThreadPoolExecutor
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend

W, H, F = 752, 480, 256
REAL = len(sys.argv) > 1 and sys.argv[1] == "--real"
if REAL:
    del sys.argv[1]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = capi.default_config(W, H, orb_nfeatures=1200, lsd_nfeatures=100, max_frames=F)
fe = Frontend(cfg)
dev = torch.device("cuda", 0)
fe.set_stream(torch.cuda.current_stream().cuda_stream)
table = torch.zeros(F * int(fe.layout.record_bytes), dtype=torch.uint8, device=dev)


def run(images, reps=3):
    d = torch.from_numpy(images).to(dev)
    l, r = d[:, 0].contiguous(), d[:, 1].contiguous()
    out = []
    for i in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fe.batch_run_device(F, l.data_ptr(), r.data_ptr(), W, W * H, table.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        st = fe.lsd_round_stats()
        out.append({"ms": round(ms, 2), "launched_without_look": st[0], "needed": st[1], "fallback_images_total": st[2]})
    return out


if REAL:
    from pli_slam_amd import realdata
    res = {"workload": "752x480 windows of real photographs (tests/golden/real), 1200 kp + <=100 lines, F = 256, lsd_mode auto", "batches": []}
    for b in range(nb):
        frames = realdata.frames_752x480(F, seed=100 + b)
        res["batches"].append({"frames": "256 distinct windows, seed %d" % (100 + b), "calls": run(np.stack([np.stack(p) for p in frames]))})
    names = sorted(k for k in realdata.photos() if k != "motorcycle_right")
    for i, name in enumerate(names):          # one photograph at a time: frames i, i + 13, ... of a seeded set are windows of photograph i
        frames = [f for j, f in enumerate(realdata.frames_752x480(len(names) * 8, seed=7)) if j % len(names) == i]
        images = np.stack([np.stack(frames[j % len(frames)]) for j in range(F)])
        res["batches"].append({"frames": "photograph %s: 8 windows, cycled" % name, "calls": run(images, reps=2)})
    needed = [c["needed"] for b in res["batches"] for c in b["calls"]]
    ms = [c["ms"] for b in res["batches"][:nb] for c in b["calls"][1:]]
    res["summary"] = {"rounds_min": min(needed), "rounds_max": max(needed), "ms_median_of_the_mixed_batches": float(np.median(ms)),
                      "fallback_images_total": res["batches"][-1]["calls"][-1]["fallback_images_total"]}
    print(json.dumps(res, indent=1))
    sys.exit(0)
res = {"workload": "752x480, 1200 kp + <=100 lines, F = 256, lsd_mode auto", "batches": []}
with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
    for b in range(nb):
        pairs = list(ex.map(lambda s: synth.make_stereo_pair(s, W, H), range(1000 + b * F, 1000 + (b + 1) * F)))
        images = np.stack([np.stack(p) for p in pairs])
        res["batches"].append({"seeds": [1000 + b * F, 1000 + (b + 1) * F - 1], "calls": run(images)})
# a batch whose instants come from few scenes (a real stream: consecutive frames look alike)
with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
    pairs = list(ex.map(lambda i: synth.make_stereo_pair(500 + i // 32, W, H, t=i % 32), range(F)))
res["batches"].append({"seeds": "8 scenes x 32 consecutive instants", "calls": run(np.stack([np.stack(p) for p in pairs]))})
# hostile images (the relaxation's worst cases), each filling a whole batch with its left image = right image
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:H, 0:W]
hostile = {
    "noise": rng.integers(0, 256, (H, W), dtype=np.uint8),
    "checker": (((xx // 16) + (yy // 16)) % 2 * 200 + 20).astype(np.uint8),
    "stripes": ((np.sin((xx + 2 * yy) / 5.0) * 0.5 + 0.5) * 255).astype(np.uint8),
    "ramp": ((xx * 255) // (W - 1)).astype(np.uint8),
}
hostile["mixed"] = np.where(xx < W // 2, hostile["noise"], hostile["stripes"]).astype(np.uint8)
for name, img in hostile.items():
    images = np.broadcast_to(img, (F, 2, H, W)).copy()
    res["batches"].append({"seeds": "hostile: " + name, "calls": run(images, reps=2)})
needed = [c["needed"] for b in res["batches"][:nb + 1] for c in b["calls"]]
ms = [c["ms"] for b in res["batches"][:nb + 1] for c in b["calls"][1:]]
res["summary_seeded"] = {"rounds_min": min(needed), "rounds_median": float(np.median(needed)), "rounds_max": max(needed),
                         "ms_min": min(ms), "ms_median": float(np.median(ms)), "ms_max": max(ms),
                         "distinct_pairs": nb * F + F}
print(json.dumps(res, indent=1))
