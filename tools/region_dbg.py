import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pli_slam_amd import capi, synth
from pli_slam_amd.frontend import Frontend
from oracle import pyoracle as po
RANK = int(sys.argv[1]) if len(sys.argv) > 1 else 3838
L, R = synth.make_stereo_pair(0)
cfg = capi.default_config(752, 480, lsd_nfeatures=100)
fe = Frontend(cfg); fe.debug_enable(True)
os.environ["PLI_JR_BIG"] = "48"; os.environ["PLI_JR_BIGBLOCKS"] = "1"; os.environ["PLI_DBG_RANK"] = str(RANK)
n, kl, ld = fe.line_extract(0, L)
raw = fe.debug_fetch(0, capi.DBG_LSD_OWNER).view(np.int32); own = raw[1:]
sizes = fe.debug_fetch(0, capi.DBG_LSD_SIZES).view(np.int32)
dq = fe.debug_fetch(0, 14).view(np.int32)
cnt = dq[0]; q = dq[1:1 + cnt]
W = 902
gpu = [((e >> 16), (e & 0xFFFF)) for e in q]
print("rank", RANK, "gpu cnt", cnt, "recorded", sizes[RANK], "true", int((own == RANK).sum()))
# CPU regrowth with the final owner map
ang = fe.debug_fetch(0, capi.DBG_LSD_ANGLE).view(np.float32)
order = fe.debug_fetch(0, capi.DBG_LSD_ORDER).view(np.int32); order = order[1:1 + order[0]]
import math
D2R = math.pi / 180; prec = math.pi * 22.5 / 180
s = order[RANK]; H = 576
ra = float(ang[s]) * D2R
sx = np.float32(math.cos(ra)); sy = np.float32(math.sin(ra))
reg = [(s // W, s % W)]; mine = {s}
k = 0
while k < len(reg):
    py, px = reg[k]
    for yy in range(max(py - 1, 0), min(py + 1, H - 1) + 1):
        for xx in range(max(px - 1, 0), min(px + 1, W - 1) + 1):
            p = yy * W + xx
            if ang[p] == -1024 or p in mine or own[p] < RANK: continue
            a = float(ang[p]) * D2R
            nt = abs(ra - a)
            if nt > 1.5 * math.pi: nt = abs(nt - 2 * math.pi)
            if nt <= prec:
                mine.add(p); reg.append((yy, xx))
                af = float(np.float32(a))
                sx = np.float32(sx + np.float32(math.cos(af))); sy = np.float32(sy + np.float32(math.sin(af)))
                ra = po.fast_atan2(float(sy), float(sx)) * D2R
    k += 1
print("cpu cnt", len(reg))
own2 = fe.debug_fetch(0, 15).view(np.int32).reshape(-1, 2)
rounds = raw[0]
pc = (rounds - 1) & 1
gset = set(gpu)
for (yy, xx) in reg:
    if (yy, xx) not in gset:
        p = yy * W + xx
        print("missing", (yy, xx), "prev", own2[p, pc], "cur", own2[p, 1 - pc], "rank[p]", int(np.flatnonzero(order == p)[0]), "size of cur owner", sizes[own2[p, 1 - pc]] if own2[p, 1 - pc] < len(sizes) else None)
for i in range(max(len(reg), len(gpu))):
    a = reg[i] if i < len(reg) else None; b = gpu[i] if i < len(gpu) else None
    if a != b:
        print("first divergence at", i, "cpu", a, "gpu", b); print("cpu around", reg[max(0, i - 3):i + 3]); print("gpu around", gpu[max(0, i - 3):i + 3]); break
else:
    print("identical lists")
d = np.flatnonzero((own2[:, 0] != own2[:, 1]) & (own2[:, pc] != 0x7fffffff))
print("pixels where emit-round claims differ from the fixed point:", d.size)
if d.size:
    lo = np.minimum(own2[d, 0], own2[d, 1])
    i = np.argmin(lo)
    p = d[i]
    print("lowest rank involved:", lo[i], "at pixel", (p // W, p % W), "prev", own2[p, pc], "cur", own2[p, 1 - pc], "sizes prev-owner", sizes[own2[p, pc]], "cur-owner", sizes[own2[p, 1 - pc]])
    srt = np.argsort(lo)[:12]
    for j in srt:
        p = d[j]; print("  pix", (p // W, p % W), "prev", own2[p, pc], "cur", own2[p, 1 - pc], "rank[p]", int(np.flatnonzero(order == p)[0]))
