#!/usr/bin/env python3
"""Per-kernel achieved bandwidth of one bench.py JSON line: algorithmic bytes of bench.py's table x images per launch / the
HIP-event time of the kernel per step.   python tools/roofline_table.py profiles/r01_bench_default.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pli_slam_amd import capi

d = json.load(open(sys.argv[1]))
F = d["config"]["frames_per_gpu"]
W, H, nkp, nl = 752, 480, 1200, 100
cfg = capi.default_config(W, H, orb_nfeatures=nkp, lsd_nfeatures=nl, max_frames=1)
_, g = bench.path_bytes_per_frame(W, H, cfg.orb_nlevels, cfg.orb_scale_factor, nkp, nl, cfg.lsd_scale)
k = dict(d["roofline"]["kernel_ms_per_step"])
side = d["roofline"].get("side_stream_kernel_ms_per_step", {})      # (round 4 on: the ORB chain's wait-inflated times are listed apart)
k.update(side)
print("| kernel | ms / step | algorithmic GB / step | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|")
tot_b = 0.0
for name, ms in sorted(k.items(), key=lambda kv: -kv[1]):
    per = bench.kernel_bytes_per_image(name, g, nkp, nl, W, H)
    if name == "k_resize_level":
        per = 2 * g["SP"] - g["P0"]          # every level read once and written once, except level 0 (read only) ...
    if per is None or ms <= 0:
        continue
    images = F if name.startswith("k_stereo") else 2 * F
    gb = per * images / 1e9
    if gb / (ms * 1e-3) > 8000:              # above the peak: the launch ended at once (ordered-list kernels in key mode)
        continue
    tot_b += gb
    print("| `%s`%s | %.2f | %.2f | %.0f | %.1f %% |" % (name, " (side stream: includes its wait for wave slots)" if name in side else "", ms, gb,
                                                       gb / (ms * 1e-3), 100 * gb / (ms * 1e-3) / 8000))
print("| whole step | %.1f | %.1f | %.0f | %.1f %% |" % (d["ms_per_step"], d["config"]["bytes_per_frame"] * F / 1e9,
                                                     d["roofline"]["path_achieved"], 100 * d["roofline"]["path_frac"]))
