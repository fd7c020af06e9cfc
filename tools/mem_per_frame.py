#!/usr/bin/env python3
"""Dev tool (GPU box): HBM taken by a context per stereo frame (sequential and auto LSD modes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pli_slam_amd import capi
from pli_slam_amd.frontend import Frontend
torch.cuda.init()
for F, mode in ((256, 0), (1024, 0), (1024, 2), (2048, 0)):
    free0, _ = torch.cuda.mem_get_info()
    fe = Frontend(capi.default_config(752, 480, max_frames=F, lsd_mode=mode))
    free1, tot = torch.cuda.mem_get_info()
    print("F=%d mode=%d: %.2f GB, %.2f MB per frame (device %.0f GB)" % (F, mode, (free0 - free1) / 1e9, (free0 - free1) / F / 1e6, tot / 1e9))
    del fe
