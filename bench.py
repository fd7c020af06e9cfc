#!/usr/bin/env python3
"""Headline benchmark: stereo frames/s of the point+line front-end on MI355X.

  python bench.py --gpus N --steps K --warmup W [--frames-per-gpu F]

One "step" = one pass of the whole per-frame hot path (ORB extract x2, LSD/LBD
extract x2, stereo point + line matching) over a batch of F (default 2048)
synthetic EuRoC-shaped stereo frames (752x480, 1200 ORB features, 100 lines) that
are already resident in HBM, followed for N > 1 by the RCCL gather of the per-frame
result tables to rank 0.  Weak scaling: every rank processes its own F frames.
Rank 0 prints ONE JSON line (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np


# algorithmic bytes per stereo frame of the whole path.
def path_bytes_per_frame(W, H, nlevels, scale, n_kp, n_l, lsd_scale=1.2):
    P = []
    s = np.float32(1.0)
    for l in range(nlevels):
        inv = np.float32(1.0) / s
        P.append(int(np.rint(np.float32(W) * inv)) * int(np.rint(np.float32(H) * inv)))
        s = np.float32(s * np.float32(scale))
    P0, SP = P[0], sum(P)
    Pp = int(np.rint(W * lsd_scale)) * int(np.rint(H * lsd_scale))
    ell = 0.1 * max(W, H)
    orb = P0 + 2 * (SP - P0) + 2 * SP
    lsd = P0 + 2 * Pp + 16 * Pp
    lbd = P0 + 2 * P0 + 4 * P0 + 252 * n_l * ell
    out = 52 * n_kp + 100 * n_l
    b_img = orb + lsd + lbd + out
    return 2 * b_img + 64 * (n_kp + n_l), {"P0": P0, "SP": SP, "Pp": Pp}


# Algorithmic bytes of ONE launch of each kernel per image (terms of the §8(d) formula;
# DESIGN.md "Kernels" lists the derivation).  Used for the roofline of the dominant kernel.
def kernel_bytes_per_image(name, g, n_kp, n_l, W, H):
    P0, SP, Pp = g["P0"], g["SP"], g["Pp"]
    ell = 0.1 * max(W, H)
    table = {
        "k_ingest": 2 * P0,
        "k_resize_level": None,                       # per-level launches, see below
        "k_fast_cells": SP,                           # read every pyramid pixel once
        "k_octree": 8 * 10 * n_kp,                    # candidate list in/out
        "k_blur_orb": 2 * SP,                         # pyramid read + blurred pyramid write
        "k_describe": SP + 52 * n_kp,                 # blurred pyramid read (patches) + tables
        "k_kp_counts": 64,
        "k_blur_lsd": 2 * P0,
        "k_resize_lsd": P0 + Pp,
        "k_lsd_grad": Pp + 16 * Pp,                   # scaled u8 read; angle/modgrad/cos/sin write
        "k_lsd_hist": 4 * Pp,
        "k_lsd_scan": 0,
        "k_lsd_scatter": 8 * Pp,
        "k_lsd_grow": 8 * Pp,                         # f32 angle + f32 modgrad read once
        "k_lsd_grow2": 8 * Pp,                        # the same kernel, two image waves per block (large batches)
        # relaxation mode (lsd_relax.hip), per launch; the growers touch the same angle/modgrad planes once
        # per round in the ideal case
        "k_rx_grow": 8 * Pp,
        "k_rx_grow_big": 8 * Pp,
        "k_rx_grow_wave": 8 * Pp,
        "k_rx_seed": 12 * Pp,                         # rank + owner pair read
        "k_rx_classify": 12 * Pp,
        "k_rx_diff": 8 * Pp,
        "k_rx_guess": 8 * Pp + 8 * Pp,
        "k_rx_rank": 8 * Pp,
        "k_rx_rect": 0,
        "k_rx_count": 0,
        "k_rx_emit": 0,
        "k_keylines": 100 * n_l,
        "k_blur_lbd": 2 * P0,
        "k_sobel": P0 + 4 * P0,
        "k_lbd": 252 * n_l * ell + 100 * n_l,
        "k_stereo_points": 64 * n_kp,
        "k_stereo_median": 8 * n_kp,
        "k_stereo_lines": 64 * n_l,
    }
    return table.get(name)


def cpu_baseline(images, cfg_bytes, budget_s=20.0):
    """The oracle (CPU restatement of the reference path) timed on the host cores of this box."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as po
    cores = os.cpu_count() or 1
    nthreads = max(1, min(cores, 64))
    frames = [po.Frame(po.Config.from_buffer_copy(cfg_bytes)) for _ in range(nthreads)]
    nimg = images.shape[0]
    # calibrate on one frame, then size the sample to ~budget_s of wall time
    t0 = time.perf_counter()
    frames[0].run(images[0, 0], images[0, 1])
    t1 = time.perf_counter() - t0
    total = int(max(nthreads, min(24 * nthreads, budget_s / max(t1, 1e-3) * nthreads * 0.6)))

    def work(tid):
        k = 0
        for i in range(tid, total, nthreads):
            frames[tid].run(images[i % nimg, 0], images[i % nimg, 1])
            k += 1
        return k

    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:
        done = sum(ex.map(work, range(nthreads)))
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "stereo frames/s", "cores": nthreads, "kind": "port",
            "sample": "%d stereo frames (cycled over the GPU batch) on %d threads, one oracle frame pipeline per "
                      "thread, %.1f s; single frame single thread %.3f s" % (done, nthreads, dt, t1)}


def parity_check(fe, cfg, d_table, images, F, nuniq, rec_bytes):
    """The table of the last timed step against the oracle (part of the cpu_baseline leg, outside the timed region):
    every record must equal the first record of its image pair, and the first and last distinct pairs must equal the
    oracle's frame (keypoints, descriptors, keylines, LBD, uRight/depth, line disparities), byte for byte."""
    import torch
    from oracle import pyoracle as po
    recs = d_table.view(F, rec_bytes)
    idx = torch.arange(F, device=recs.device) % nuniq
    dup_bad = int((recs != recs[idx]).any(dim=1).sum().item())
    host = recs[:nuniq].cpu().numpy().reshape(-1)
    checked, bad = 0, 0
    for i in sorted({0, nuniq - 1}):
        r = fe.parse_record(host, i)
        fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
        ok = True
        for eye, k in ((0, "L"), (1, "R")):
            n, kp, desc = fr.orb_extract(eye, images[i, eye])
            ok &= n == len(r["kp" + k]) and kp.tobytes() == r["kp" + k].tobytes() and np.array_equal(desc, r["desc" + k])
            m, kl, ld = fr.line_extract(eye, images[i, eye])
            ok &= m == len(r["kl" + k]) and kl.tobytes() == r["kl" + k].tobytes() and np.array_equal(ld, r["ldesc" + k])
        ur, dp, _, _ = fr.stereo_points()
        ok &= ur.tobytes() == r["uright"].tobytes() and dp.tobytes() == r["depth"].tobytes()
        disp, le, _ = fr.stereo_lines()
        ok &= disp.tobytes() == r["disp"].tobytes()
        checked += 1
        bad += 0 if ok else 1
    return {"ok": dup_bad == 0 and bad == 0, "records": F, "records_differing_from_first_copy": dup_bad,
            "pairs_checked_against_oracle": checked, "pairs_mismatching": bad}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-gpu", type=int, default=2048)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1200)
    ap.add_argument("--nlines", type=int, default=100)
    ap.add_argument("--unique-frames", type=int, default=64)
    ap.add_argument("--streams", type=int, default=1, help="split the per-GPU batch over this many contexts/HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lsd-mode", type=int, default=0, help="0 auto, 1 relaxation, 2 sequential waves")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the front-end has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pli_slam_amd import capi, synth
    from pli_slam_amd.frontend import Frontend

    F, W, H = args.frames_per_gpu, args.width, args.height
    S = max(1, args.streams)
    assert F % S == 0, "--frames-per-gpu must be a multiple of --streams"
    Fs = F // S
    cfg = capi.default_config(W, H, orb_nfeatures=args.nfeatures, lsd_nfeatures=args.nlines, max_frames=Fs,
                              lsd_mode=args.lsd_mode if args.lsd_mode else (2 if 2 * Fs >= 640 else 1))   # = the library's auto rule
    fes = [Frontend(cfg, device=local_rank) for _ in range(S)]
    fe = fes[0]
    # synthetic stream: up to 64 distinct seeded stereo pairs per rank (seeds disjoint across ranks), cycled to F frames
    nuniq = min(F, args.unique_frames)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        pairs = list(ex.map(lambda s_: synth.make_stereo_pair(s_, W, H), range(rank * nuniq, rank * nuniq + nuniq)))
    images = np.stack([np.stack(p) for p in pairs])                    # (nuniq, 2, H, W) u8
    d_uniq = torch.from_numpy(images).to(dev)
    d_img = d_uniq[torch.arange(F, device=dev) % nuniq].contiguous()   # (F, 2, H, W) resident in HBM before timing
    d_left, d_right = d_img[:, 0].contiguous(), d_img[:, 1].contiguous()
    rec_bytes = int(fe.layout.record_bytes)
    d_table = torch.zeros(F * rec_bytes, dtype=torch.uint8, device=dev)
    from pli_slam_amd.sharding import TableGatherer
    # N > 1: the result tables are gathered to rank 0 over RCCL, double buffered, so that the gather of one step travels
    # while the kernels of the next run; every gather is complete before the closing barrier of the timed region
    gath = TableGatherer(F * rec_bytes, dev) if world > 1 else None
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]
    for f_, st_ in zip(fes, streams):
        f_.set_stream(st_.cuda_stream)

    def step():
        slot = gath.acquire() if gath else 0
        tbl = gath.table(slot) if gath else d_table
        for i, f_ in enumerate(fes):
            f_.batch_run_device(Fs, d_left[i * Fs:].data_ptr(), d_right[i * Fs:].data_ptr(), W, W * H,
                                tbl[i * Fs * rec_bytes:].data_ptr())
        for st_ in streams[1:]:
            torch.cuda.current_stream().wait_stream(st_)
        if gath:
            gath.submit(slot)

    def fence():
        if gath:
            gath.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    for f_ in fes:
        f_.prof_reset()
        f_.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = {}
    for f_ in fes:
        f_.prof_enable(False)
        for k_, (c_, ms_) in f_.prof_report().items():
            a_ = prof.get(k_, (0, 0.0))
            prof[k_] = (a_[0] + c_, a_[1] + ms_)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        b_frame, g = path_bytes_per_frame(W, H, cfg.orb_nlevels, cfg.orb_scale_factor, args.nfeatures, args.nlines,
                                          cfg.lsd_scale)
        fps = world * F * args.steps / dt
        # dominant kernel by HIP-event time on the stream it ran on
        dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else (None, (0, 0.0))
        name, (calls, total_ms) = dom
        per_img = kernel_bytes_per_image(name, g, args.nfeatures, args.nlines, W, H) if name else None
        avg_s = (total_ms / max(calls, 1)) * 1e-3
        peak = 8000.0
        if per_img is not None and avg_s > 0:
            achieved = per_img * 2 * Fs / avg_s / 1e9
        else:
            achieved = None
        traffic = None     # HBM bytes per launch from the committed PMC passes of the same workload, if any
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            k = tr["workloads"].get(str(F), {}).get(name) if (W, H) == (752, 480) else None
            if k:
                traffic = (2 * k["fetch_kb"] + k["write_kb"]) * 1024
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": peak, "unit": "GB/s",
                "frac": (achieved / peak) if achieved is not None else None, "traffic": traffic,
                "avg_launch_ms": avg_s * 1e3, "launches": calls,
                "path_achieved": fps / world * b_frame / 1e9, "path_frac": fps / world * b_frame / 1e9 / peak,
                "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in
                                       sorted(prof.items(), key=lambda kv: -kv[1][1])}}
        out = {
            "metric": "stereo frames/sec (ORB+LSD extract+match), 752x480 EuRoC",
            "value": fps, "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic (%d distinct seeded EuRoC-shaped pairs per GPU, cycled)" % nuniq,
            "config": {"workload": "1xMI355X: %dx%d stereo pairs, %d ORB kp (8 levels x1.2) + LSD/LBD (<=%d lines), "
                                   "extract + stereo Hamming match; batch of %d stereo frames per GPU per step" %
                                   (W, H, args.nfeatures, args.nlines, F),
                       "frames_per_gpu": F, "bytes_per_frame": b_frame,
                       "parallelism": "frame-batch data parallel, %d rank(s), gather of result tables to rank 0" % world},
            "roofline": roof,
        }
        if world == 1 and F > 1 and not args.no_cpu_baseline:
            # BASELINE.json configs[1] (a single stereo pair) beside the batch: latency of one pair through the same
            # library, device buffers, outside the timed region
            f1 = Frontend(capi.default_config(W, H, orb_nfeatures=args.nfeatures, lsd_nfeatures=args.nlines, max_frames=1),
                          device=local_rank)
            f1.set_stream(torch.cuda.current_stream().cuda_stream)
            t1 = torch.zeros(rec_bytes, dtype=torch.uint8, device=dev)
            for rep in range(12):
                if rep == 2:
                    torch.cuda.synchronize(); ts = time.perf_counter()
                f1.batch_run_device(1, d_left.data_ptr(), d_right.data_ptr(), W, W * H, t1.data_ptr())
            torch.cuda.synchronize()
            out["single_pair"] = {"ms": (time.perf_counter() - ts) / 10 * 1e3, "note": "one stereo pair per call, 10 calls"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(images, bytes(cfg))
            out["parity"] = parity_check(fe, cfg, d_table, images, F, nuniq, rec_bytes)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
