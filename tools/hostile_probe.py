import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pli_slam_amd import capi
from pli_slam_amd.frontend import Frontend
W, H = 752, 480
yy, xx = np.mgrid[0:H, 0:W]
stripes = ((np.sin((xx + 2 * yy) / 5.0) * 0.5 + 0.5) * 255).astype(np.uint8)
os.environ["PLI_RX_TRACE"] = "1"
fe = Frontend(capi.default_config(W, H, orb_nfeatures=300, lsd_nfeatures=0, max_frames=1, lsd_mode=3))
m, kl, ld = fe.line_extract(0, stripes)
print("lines", m, fe.lsd_round_stats())
