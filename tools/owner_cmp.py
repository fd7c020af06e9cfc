import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
L, R = synth.make_stereo_pair(0)
cfg = capi.default_config(752, 480, lsd_nfeatures=100)
fe = Frontend(cfg)
res = {}
for big in ("100000000", "48", "8"):
    os.environ["PLI_JR_BIG"] = big
    n, kl, ld = fe.line_extract(0, L)
    raw = fe.debug_fetch(0, capi.DBG_LSD_OWNER).view(np.int32)
    segs = fe.debug_fetch(0, capi.DBG_LSD_SEGMENTS)
    ns = int(segs[:4].view(np.int32)[0])
    res[big] = (raw[0], raw[1:].copy(), segs[4:4 + 16 * ns].view(np.float32).reshape(-1, 4).copy())
    sizes = fe.debug_fetch(0, capi.DBG_LSD_SIZES).view(np.int32)
    own = raw[1:]
    valid = own != 0x7fffffff
    true_sizes = np.bincount(own[valid])
    alive = np.flatnonzero(true_sizes > 0)
    bad = alive[sizes[alive] != true_sizes[alive]]
    print("BIG", big, "rounds", raw[0], "nseg", ns, "regions", alive.size, "size mismatches", bad.size, bad[:8], sizes[bad[:8]], true_sizes[bad[:8]], "n>=16 true", (true_sizes >= 16).sum())
a = res["100000000"]
for big in ("48", "8"):
    b = res[big]
    d = np.flatnonzero(a[1] != b[1])
    print("BIG", big, "owner diffs", d.size, d[:10], a[1][d[:5]], b[1][d[:5]])
    if a[2].shape == b[2].shape:
        dr = np.flatnonzero((a[2] != b[2]).any(1)); print("  seg rows differing", dr.size, dr[:10])
    else:
        print("  seg count differs", a[2].shape, b[2].shape)
