#!/usr/bin/env python3
"""Headline benchmark: stereo frames/s of the point+line front-end on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}] [--frames-per-gpu F]

One "step" = one pass of the whole per-frame hot path (ORB extract x2, LSD/LBD extract x2, stereo point + line
matching; with --config 3 also the frame-to-frame track matching) over a batch of synthetic EuRoC-shaped stereo
frames that are already resident in HBM, followed for N > 1 by the RCCL gather of the per-frame result tables to
rank 0.  Rank 0 prints ONE JSON line (the driver contract of the task statement).

Workloads (BASELINE.json `configs`):
  default      752x480, 1200 ORB kp + <=100 lines, F frames per GPU per step (weak scaling)
  --config 2   the same image size, ONE stereo pair per call (latency)
  --config 3   1280x720, 2000 kp + 200 lines, consecutive frames of one scene, + frame-to-frame track matching
  --config 4   752x480 stream in 256-frame batches sharded over the N ranks (32 per GPU on 8 GPUs), gather of the
               shard tables to rank 0 every batch; --inflight K keeps K batches per step in flight
  --config 5   3840x2160, 4000 kp + 500 lines

N > 1: one process per GPU.  Started by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the
environment, WORLD_SIZE must equal --gpus) or, when those are absent, bench.py starts the N rank processes itself
BEFORE anything touches the GPU.  --backend gloo --dry-tables exercises the sharding + gather path on CPU
(no kernels): used by tests/test_bench_launch.py.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

CONFIGS = {
    2: dict(W=752, H=480, nf=1200, nl=100, what="single 752x480 stereo pair per call (BASELINE configs[1])"),
    3: dict(W=1280, H=720, nf=2000, nl=200, what="1280x720 stereo, 8 levels, 2000 kp + 200 lines, + frame-to-frame track match "
                                                 "(BASELINE configs[2])"),
    4: dict(W=752, H=480, nf=1200, nl=100, what="752x480 stream, 256-frame batches sharded over the ranks, gather of the shard "
                                                "tables to rank 0 (BASELINE configs[3])"),
    5: dict(W=3840, H=2160, nf=4000, nl=500, what="3840x2160 stereo, 4000 kp + 500 lines (BASELINE configs[4])"),
}
BATCH4 = 256          # frames per batch of config 4


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
        "k_resize_level": None,                       # per-level launches
        "k_fast_cells": SP,                           # read every pyramid pixel once
        "k_octree": 8 * 10 * n_kp,                    # candidate list in/out
        "k_blur_orb": 2 * SP,                         # pyramid read + blurred pyramid write
        "k_describe": SP + 52 * n_kp,                 # blurred pyramid read (patches) + tables
        "k_kp_counts": 64,
        "k_blur_lsd": 2 * P0,
        "k_resize_lsd": P0 + Pp,
        "k_lsd_grad": Pp + 16 * Pp,                   # scaled u8 read; angle/modgrad/cos/sin write
        "k_lsd_front": P0 + 16 * Pp,                  # fused blur -> resize -> gradient of the CV_64F pipeline: u8 read; record write
        "k_lsd_hist": 4 * Pp,
        "k_lsd_scan": 0,
        "k_lsd_scatter": 8 * Pp,
        "k_lsd_grow": 8 * Pp,                         # f32 angle + f32 modgrad read once
        "k_lsd_grow2": 8 * Pp,
        # relaxations (lsd_relax.hip, lsd_tile.hip), per launch: in the ideal case a grower touches the angle/modgrad
        # planes once per round
        "k_tx_grow": 8 * Pp,
        "k_tx_grow_sparse": 8 * Pp,
        "k_tx_sort": 8 * Pp + 8 * Pp,                 # rank read, owner pair write; list write
        "k_tx_prep": 12 * Pp,
        "k_rx_grow": 8 * Pp,
        "k_rx_grow_big": 8 * Pp,
        "k_rx_grow_wave": 8 * Pp,
        "k_rx_seed": 12 * Pp,
        "k_rx_classify": 12 * Pp,
        "k_rx_diff": 8 * Pp,
        "k_rx_mark": 12 * Pp,
        "k_tx_diff2": 8 * Pp,
        "k_tx_round2": 20 * Pp,
        "k_tx_diffmark": 12 * Pp,
        "k_rx_guess": 8 * Pp + 8 * Pp,
        "k_rx_rect": 0,
        "k_rx_count": 0,
        "k_rx_emit": 0,
        "k_tx_collect": 4 * Pp,                       # key mode: the size plane read once (ids / owners only at the few candidates)
        "k_tx_emit_sorted": 0,
        "k_zero_ranges": 8 * Pp,                      # the two stamp planes (+ control blocks, cell tables, counters)
        "k_keylines": 100 * n_l,
        "k_blur_lbd": 2 * P0,
        "k_sobel": P0 + 4 * P0,
        "k_lbd": 252 * n_l * ell + 100 * n_l,
        "k_stereo_points": 64 * n_kp,
        "k_stereo_median": 8 * n_kp,
        "k_stereo_lines": 64 * n_l,
    }
    return table.get(name)


# --------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (CPU restatement of the reference path) timed on the host cores of this box
# --------------------------------------------------------------------------------------------------------------
def cpu_baseline(images, cfg_bytes, budget_s=18.0):
    """Three variants: single thread, the reference's own threading (4 threads per frame:
    ORB-L, ORB-R, LSD-L, LSD-R in parallel as Frame.cc:128-135, then the two stereo matchers serially), and all host
    cores frame-parallel.  `value` is the all-cores rate."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as po
    flags = po.use_native_build()
    cores = os.cpu_count() or 1
    nimg = images.shape[0]
    f0 = po.Frame(po.Config.from_buffer_copy(cfg_bytes))
    t0 = time.perf_counter()
    f0.run(images[0, 0], images[0, 1])
    t1 = time.perf_counter() - t0
    # reference-style: 4 threads per frame (ctypes releases the GIL)
    n4 = int(max(2, min(24, 4.0 / max(t1 / 2.0, 1e-3))))
    with ThreadPoolExecutor(4) as ex4:
        t0 = time.perf_counter()
        for i in range(n4):
            L, R = images[i % nimg, 0], images[i % nimg, 1]
            futs = [ex4.submit(f0.orb_extract, 0, L), ex4.submit(f0.orb_extract, 1, R),
                    ex4.submit(f0.line_extract, 0, L), ex4.submit(f0.line_extract, 1, R)]
            for f_ in futs:
                f_.result()
            f0.stereo_lines()
            f0.stereo_points()
        dt4 = time.perf_counter() - t0
    # frame-parallel on the host cores: one oracle frame pipeline per thread, on every logical CPU and on half of them (one thread per
    # physical core where SMT is on: the oracle is FP-heavy and the two settings differ by +-30 % from box to box); the better rate is
    # the baseline, both are reported
    frames = [po.Frame(po.Config.from_buffer_copy(cfg_bytes)) for _ in range(cores)]

    def run_on(nth, budget):
        deadline = time.perf_counter() + budget          # (time-boxed: every thread takes frames until the budget is spent)

        def work(tid):
            k = 0
            while time.perf_counter() < deadline:
                i = tid + k * nth
                frames[tid].run(images[i % nimg, 0], images[i % nimg, 1])
                k += 1
            return k
        t0_ = time.perf_counter()
        with ThreadPoolExecutor(nth) as ex:
            done_ = sum(ex.map(work, range(nth)))
        return done_, time.perf_counter() - t0_
    trials = {}
    for nth in sorted({cores, max(1, cores // 2)}, reverse=True):
        trials[nth] = run_on(nth, budget_s / 3)
    nthreads = max(trials, key=lambda n_: trials[n_][0] / trials[n_][1])
    done, dt = trials[nthreads]
    # ... and the shipped build of the oracle (-O3 -march=x86-64-v3) at that thread count: on some hosts it is the faster one
    shipped = None
    if flags.startswith("-O3 -march=native"):
        po.use_shipped_build()
        frames = [po.Frame(po.Config.from_buffer_copy(cfg_bytes)) for _ in range(nthreads)]
        shipped = run_on(nthreads, budget_s / 3)
        if shipped[0] / shipped[1] > done / dt:
            done, dt = shipped
            flags = "-O3 -march=x86-64-v3 (the shipped build; faster on this host than -O3 -march=native: %.1f frames/s)" % (
                trials[nthreads][0] / trials[nthreads][1])
    return {"value": done / dt, "unit": "stereo frames/s", "cores": nthreads, "kind": "port",
            "sample": "%d stereo frames (cycled over the GPU batch) on %d threads of the box's %d logical CPUs (the better of all / half "
                      "of them), one oracle frame pipeline per thread, oracle built %s, %.1f s" % (done, nthreads, cores, flags, dt),
            "by_threads_native_build": {str(n_): round(trials[n_][0] / trials[n_][1], 2) for n_ in trials},
            "shipped_build_at_that_thread_count": round(shipped[0] / shipped[1], 2) if shipped else None,
            "single_thread": {"value": 1.0 / t1, "seconds_per_frame": t1, "cores": 1},
            "ref4": {"value": n4 / dt4, "cores": 4, "sample": "%d stereo frames, the four extractors of a frame on 4 threads "
                                                              "(Frame.cc:128-135), matching serial, %.1f s" % (n4, dt4)},
            "host": {"logical_cpus": cores}}


def oracle_frame_equal(fe, r, images_pair, cfg):
    """One parsed record against a fresh oracle frame: every table, byte for byte."""
    from oracle import pyoracle as po
    fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
    ok = True
    for eye, k in ((0, "L"), (1, "R")):
        n, kp, desc = fr.orb_extract(eye, images_pair[eye])
        ok &= n == len(r["kp" + k]) and kp.tobytes() == r["kp" + k].tobytes() and np.array_equal(desc, r["desc" + k])
        m, kl, ld = fr.line_extract(eye, images_pair[eye])
        ok &= m == len(r["kl" + k]) and kl.tobytes() == r["kl" + k].tobytes() and np.array_equal(ld, r["ldesc" + k])
    ur, dp, _, _ = fr.stereo_points()
    ok &= ur.tobytes() == r["uright"].tobytes() and dp.tobytes() == r["depth"].tobytes()
    disp, le, _ = fr.stereo_lines()
    ok &= disp.tobytes() == r["disp"].tobytes() and le.tobytes() == r["le"].tobytes()
    return bool(ok)


def parity_check(fe, cfg, d_table, images, F, nuniq, rec_bytes, npairs=8):
    """The table of the last timed step against the oracle (part of the cpu_baseline leg, outside the timed region):
    every record must equal the first record of its image pair, and `npairs` distinct pairs (spread over the batch) must
    equal the oracle's frame (keypoints, descriptors, keylines, LBD, uRight/depth, line disparities, mvle_l), byte for byte."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    recs = d_table.view(F, rec_bytes)
    idx = torch.arange(F, device=recs.device) % nuniq
    dup_bad = int((recs != recs[idx]).any(dim=1).sum().item())
    host = recs[:nuniq].cpu().numpy().reshape(-1)
    pick = sorted({int(round(i * (nuniq - 1) / max(npairs - 1, 1))) for i in range(min(npairs, nuniq))})
    with ThreadPoolExecutor(min(len(pick), os.cpu_count() or 1)) as ex:
        oks = list(ex.map(lambda i: oracle_frame_equal(fe, fe.parse_record(host, i), images[i], cfg), pick))
    bad = sum(0 if o else 1 for o in oks)
    return {"ok": dup_bad == 0 and bad == 0, "records": F, "records_differing_from_first_copy": dup_bad,
            "pairs_checked_against_oracle": len(pick), "pairs_mismatching": bad}


def gathered_parity_check(fe, cfg, gathered, counts, rec_bytes, pair_of, per_rank=2):
    """N > 1, on the root: `per_rank` records of EVERY rank's gathered shard (first and last frame of the shard) against the
    oracle's frame for the stereo pair that rank processed there; plus, per rank, that no record of the shard is empty."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    empty = 0
    for r_, (tab, n) in enumerate(zip(gathered, counts)):
        host = tab[: n * rec_bytes].cpu().numpy()
        recs = host.reshape(n, rec_bytes) if n else host.reshape(0, rec_bytes)
        empty += int((~recs.any(axis=1)).sum())
        for i_ in sorted({0, n - 1} if per_rank >= 2 else {0}):
            if 0 <= i_ < n:
                jobs.append((r_, i_, host))
    with ThreadPoolExecutor(min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        oks = list(ex.map(lambda j: oracle_frame_equal(fe, fe.parse_record(j[2], j[1]), pair_of(j[0], j[1]), cfg), jobs))
    bad = [(j[0], j[1]) for j, o in zip(jobs, oks) if not o]
    return {"ok": not bad and empty == 0, "ranks_checked": len(counts), "records_gathered": int(sum(counts)), "empty_records": empty,
            "pairs_checked_against_oracle": len(jobs), "pairs_mismatching": len(bad), "mismatching_rank_frame": bad[:8]}


def halo_track_check(fe, cfg, track, d_halo_table, rec_bytes, prev_pair, prev_pose, own_pose, W, H):
    """Config 3 on several ranks: the track of this rank's FIRST frame was matched against the record that arrived through the
    1-frame halo.  The oracle computes the same track from scratch — the previous shard's last frame from its image pair, the
    projection of ORBmatcher.cc:2190-2244, the window search, match() of the line descriptors — and the two must be equal."""
    import torch
    from oracle import pyoracle as po
    torch.cuda.synchronize()
    tl = fe.track_layout()
    tr = fe.parse_track(track[2].cpu().numpy(), 1)                    # [halo | frame 0 | ...]: record 1 = own frame 0 against the halo
    own = fe.parse_record(d_halo_table[rec_bytes:2 * rec_bytes].cpu().numpy(), 0)
    fr = po.Frame(po.Config.from_buffer_copy(bytes(cfg)))
    n, kp, desc = fr.orb_extract(0, prev_pair[0])
    fr.orb_extract(1, prev_pair[1])
    m, kl, ld = fr.line_extract(0, prev_pair[0])
    fr.line_extract(1, prev_pair[1])
    ur, dp, _, _ = fr.stereo_points()
    tp = track[1]
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(cfg.orb_nlevels - 1, np.float32(cfg.orb_scale_factor), np.float32)])).astype(np.float32)
    q = po.track_queries(kp, dp, prev_pose, own_pose, tp.fx, tp.fy, tp.cx, tp.cy, tp.bf, tp.th, bool(tp.mono), sf)
    on, obest = po.search_by_projection(q, desc, own["kpL"], own["descL"], own["uright"], (0.0, float(W), 0.0, float(H)), bool(tp.check_orientation))
    ln, lm = po.match_lines(ld, own["ldescL"], float(tp.nnr_lines), True)
    ok = (tr["counts"][0] == n and tr["counts"][1] == on and np.array_equal(tr["best"], obest) and tr["counts"][2] == m
          and tr["counts"][3] == ln and np.array_equal(tr["lines"], lm) and on > 0)
    return {"checked": 1, "bad": 0 if ok else 1}


# --------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """--gpus N without a distributed environment: start the N rank processes (fresh children, before this process
    has touched the GPU) and return their exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def rank_evidence(rank, world, my_ms, device_id, shared_device):
    """What makes a multi-rank line self-proving (every rank calls it; the result is the same everywhere): how many ranks took part
    (an all-reduce of ones), WHICH devices they ran on (an all-gather of each rank's device identity: N distinct ones unless the run is
    a shared-device rehearsal), and every rank's own time over the timed region (min / max / the slowest rank: a straggler shows)."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    ms = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(ms, torch.tensor([my_ms], dtype=torch.float64, device=dev))
    ids = [None] * world
    dist.all_gather_object(ids, str(device_id))
    per = [float(t.item()) for t in ms]
    distinct = len(set(ids))
    return {"ranks_seen": int(one.item()), "devices": ids, "distinct_devices": distinct,
            "one_device_per_rank": bool(distinct == world),
            "rehearsal_on_shared_device": bool(shared_device or distinct < world),
            "per_rank_ms": {"min": min(per), "max": max(per), "slowest_rank": int(per.index(max(per))), "all": per}}


def device_identity(local_rank):
    """A string that differs between the GPUs of a node: the device's UUID when the runtime reports one, its PCI bus id otherwise."""
    import torch
    p = torch.cuda.get_device_properties(local_rank)
    uuid = getattr(p, "uuid", None)
    bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    return "%s uuid=%s pci=%s" % (p.name, uuid, bus)


def dry_tables(args, rank, world):
    """CPU exercise of the multi-rank path (sharding + gather of the result tables), no kernels: every rank fills the
    records of its shard with a pattern derived from the global frame index, rank 0 checks what arrives."""
    import torch
    import torch.distributed as dist
    from pli_slam_amd.sharding import TableGatherer, shard_range
    rec = 4096
    nbatch = BATCH4 if args.config == 4 else world * max(1, args.frames_per_gpu or 4)
    start, count = shard_range(nbatch, rank, world)
    counts = [shard_range(nbatch, r, world)[1] for r in range(world)]
    pad = max(counts)
    gath = TableGatherer(pad * rec, torch.device("cpu"), force=args.force_dist)
    ok = True
    t_start = time.perf_counter()
    for step in range(args.warmup + args.steps):
        slot = gath.acquire()
        t = gath.table(slot)
        t.zero_()
        for f in range(count):
            t[f * rec:(f + 1) * rec] = int((start + f + 7 * step) % 251)
        gath.submit(slot)
        gath.drain()
        if rank == 0:
            got = gath.gathered(slot)
            for r in range(world):
                s_r, c_r = shard_range(nbatch, r, world)
                for f in range(c_r):
                    ok &= bool((got[r][f * rec:(f + 1) * rec] == int((s_r + f + 7 * step) % 251)).all())
    my_ms = (time.perf_counter() - t_start) * 1e3
    dist.barrier()
    # (no devices in the dry run: a rank's identity is its process)
    ev = rank_evidence(rank, world, my_ms, "cpu pid=%d" % os.getpid(), False)
    if rank == 0:
        print(json.dumps({"dry_tables": True, "n_gpus": world, "backend": args.backend, "batch_frames": nbatch,
                          "shard_frames": counts, "gathered_bytes_per_step": sum(counts) * rec, "shards_ok": bool(ok),
                          "ranks": ev}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE.json configs[i-1]; 0 = headline batch")
    ap.add_argument("--frames-per-gpu", type=int, default=0, help="stereo frames per GPU per step (default depends on --config)")
    ap.add_argument("--inflight", type=int, default=1, help="config 4: 256-frame batches per step")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--nfeatures", type=int, default=0)
    ap.add_argument("--nlines", type=int, default=0)
    ap.add_argument("--unique-frames", type=int, default=256, help="distinct seeded stereo pairs per GPU (cycled to the batch size); 256 = every frame of the headline batch is distinct")
    ap.add_argument("--real-images", action="store_true",
                    help="the batch is cut from real photographs (pli_slam_amd/realdata.py, tests/golden/real/photos.npz) instead of "
                         "synthetic scenes: rate, relaxation rounds and fallbacks on natural gradients (752x480 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=18.0, help="budget of the CPU-baseline leg (the oracle on the host cores)")
    ap.add_argument("--no-host-leg", action="store_true")
    ap.add_argument("--no-large-batch-leg", action="store_true")
    ap.add_argument("--lsd-mode", type=int, default=0, help="0 auto, 1 relaxation, 2 sequential waves, 3 tile-sequential relaxation")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry-tables", action="store_true", help="CPU exercise of sharding + gather (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: initialise the process group anyway and run the multi-rank code path (TableGatherer on device "
                         "tensors, halo exchange, barriers, all-reduces, the gathered-table parity leg) in a world of one rank: executes "
                         "the RCCL calls of the 8-GPU run on a one-GPU box")
    ap.add_argument("--share-device", action="store_true",
                    help="all ranks run their kernels on GPU 0 (rehearsal of the multi-rank path on a one-GPU box; needs --backend gloo: "
                         "the tables are staged through the host for the gather and the halo)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: start it with --nproc-per-node %d (or without a "
                         "distributed environment: it starts the ranks itself)\n" % (args.gpus, world, args.gpus))
        sys.exit(2)

    import torch
    import torch.distributed as dist

    if args.dry_tables:
        if args.backend != "gloo":
            sys.exit("--dry-tables is the CPU exercise: use --backend gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()) if world == 1 else "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
        sys.exit(dry_tables(args, rank, world))

    if args.real_images and (args.config in (3, 5) or (args.width or 752) != 752 or (args.height or 480) != 480):
        sys.exit("--real-images cuts 752x480 windows from the photographs: not with --config 3 / 5 or another --width / --height")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the front-end has no CPU path")
    if args.share_device:
        if args.backend != "gloo":
            sys.exit("--share-device puts several ranks on one GPU, which RCCL refuses: use --backend gloo")
        local_rank = 0
        os.environ.setdefault("PLI_TX_TAIL", "0")       # several processes on one device: no spinning kernel (pli_frontend.h "Sharing a device")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl = torch.device("cpu") if args.backend == "gloo" else dev        # where the small control tensors of the collectives live
    multi = world > 1 or args.force_dist                                 # the multi-rank code path (a world of one rank when forced)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus

    from pli_slam_amd import capi, synth
    from pli_slam_amd.frontend import Frontend
    from pli_slam_amd.sharding import TableGatherer, shard_range

    C_ = CONFIGS.get(args.config, CONFIGS[2] if args.config == 0 else None)
    W, H = args.width or C_["W"], args.height or C_["H"]
    nfeat, nlines = args.nfeatures or C_["nf"], args.nlines or C_["nl"]
    if args.config == 2:
        F = 1
    elif args.config == 4:
        F = max(1, args.inflight) * shard_range(BATCH4, rank, world)[1]
    elif args.config == 5:
        F = args.frames_per_gpu or 16
    elif args.config == 3:
        F = args.frames_per_gpu or 64
    else:
        F = args.frames_per_gpu or 256
    Fmax = F
    if multi:                                         # equal table sizes for the gather (config 4 shards may differ by one)
        t = torch.tensor([F], device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        Fmax = int(t.item())
    # (config 3 on several ranks: the frame-to-frame matcher also sees the halo frame of the previous shard)
    cfg = capi.default_config(W, H, orb_nfeatures=nfeat, lsd_nfeatures=nlines, max_frames=Fmax + (1 if (args.config == 3 and multi) else 0),
                              lsd_mode=args.lsd_mode)
    fe = Frontend(cfg, device=local_rank, dev=False)      # (the product library: PLI_USE_DEV_LIB=1 — the tools — swaps in the development build)
    # synthetic stream: up to --unique-frames distinct seeded stereo pairs per rank (seeds disjoint across ranks), cycled to F
    # frames; config 3: consecutive frames t = 0..nuniq-1 of ONE scene (the motion of synth.make_stereo_pair)
    nuniq = min(F, args.unique_frames)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        if args.config == 3:
            # one scene for all ranks, rank r holds the time slice [r * nuniq, r * nuniq + nuniq): consecutive shards of ONE stream
            pairs = list(ex.map(lambda t_: synth.make_stereo_pair(100, W, H, t=rank * nuniq + t_), range(nuniq)))
        elif args.real_images:
            from pli_slam_amd import realdata
            pairs = realdata.frames_752x480(nuniq, seed=17 + rank, w=W, h=H)
        else:
            pairs = list(ex.map(lambda s_: synth.make_stereo_pair(s_, W, H), range(rank * nuniq, rank * nuniq + nuniq)))
    images = np.stack([np.stack(p) for p in pairs])                    # (nuniq, 2, H, W) u8
    d_uniq = torch.from_numpy(images).to(dev)
    d_img = d_uniq[torch.arange(F, device=dev) % nuniq].contiguous()   # (F, 2, H, W) resident in HBM before timing
    d_left, d_right = d_img[:, 0].contiguous(), d_img[:, 1].contiguous()
    rec_bytes = int(fe.layout.record_bytes)
    d_table = torch.zeros(Fmax * rec_bytes, dtype=torch.uint8, device=dev)
    # N > 1: the result tables are gathered to rank 0 over RCCL, double buffered, so that the gather of one step travels
    # while the kernels of the next run; every gather is complete before the closing barrier of the timed region
    gath = TableGatherer(Fmax * rec_bytes, dev, force=args.force_dist) if multi else None
    fe.set_stream(torch.cuda.current_stream().cuda_stream)
    track = None
    if args.config == 3:
        # frame-to-frame track matching of consecutive frames in the timed step (pli_batch_track): the poses a motion model would
        # predict for the synthetic camera motion (small translation + 0.5 deg roll per frame), SearchByProjection window th = 15
        def pose(t_):
            a_ = np.deg2rad(0.5 * t_)
            return np.array([[np.cos(a_), -np.sin(a_), 0, -0.02 * t_], [np.sin(a_), np.cos(a_), 0, -0.007 * t_], [0, 0, 1, 0.01 * t_]],
                            np.float32)
        tl = fe.track_layout()
        if multi:
            # f2f matching across shard borders: the record of the frame before this shard's first one arrives from
            # the previous rank (1-frame halo, point-to-point) and the matcher runs over [halo | own frames]
            from pli_slam_amd.sharding import exchange_halo
            halo_pose = pose((rank * nuniq - 1) % (world * nuniq))
            d_poses = torch.from_numpy(np.stack([halo_pose] + [pose(rank * nuniq + t_ % nuniq) for t_ in range(F)]).reshape(-1)).to(dev)
            d_halo_table = torch.zeros((F + 1) * rec_bytes, dtype=torch.uint8, device=dev)
            track = (d_poses, fe.track_params(th=15.0), torch.zeros((F + 1) * int(tl.record_bytes), dtype=torch.uint8, device=dev))
        else:
            d_poses = torch.from_numpy(np.stack([pose(t_ % nuniq) for t_ in range(F)]).reshape(-1)).to(dev)
            track = (d_poses, fe.track_params(th=15.0), torch.zeros(F * int(tl.record_bytes), dtype=torch.uint8, device=dev))

    last_slot = [0]

    def step():
        slot = gath.acquire() if gath else 0
        last_slot[0] = slot
        tbl = gath.table(slot) if gath else d_table
        fe.batch_run_device(F, d_left.data_ptr(), d_right.data_ptr(), W, W * H, tbl.data_ptr())
        if track is not None and multi:
            d_halo_table[rec_bytes:(F + 1) * rec_bytes].copy_(tbl[:F * rec_bytes])
            got = exchange_halo(d_halo_table, rec_bytes, F, counts=[F] * world, force=args.force_dist)
            if got:       # [halo | frames]: F + 1 consecutive records
                fe.batch_track_device(F + 1, d_halo_table.data_ptr(), track[0].data_ptr(), track[1], track[2].data_ptr())
            else:         # the first shard of the stream: nothing in front of its frame 0
                fe.batch_track_device(F, d_halo_table.data_ptr() + rec_bytes, track[0].data_ptr() + 48, track[1],
                                      track[2].data_ptr() + int(tl.record_bytes))
        elif track is not None:
            fe.batch_track_device(F, tbl.data_ptr(), track[0].data_ptr(), track[1], track[2].data_ptr())
        if gath:
            gath.submit(slot)

    def fence():
        if gath:
            gath.drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    fe.prof_reset()
    fe.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    fe.prof_enable(False)
    prof = fe.prof_report()
    frames_done = torch.tensor([float(F * args.steps)], dtype=torch.float64, device=ctl)
    ranks_ev = None
    if multi:
        ranks_ev = rank_evidence(rank, world, dt / args.steps * 1e3, device_identity(local_rank), args.share_device)
        t = torch.tensor([dt], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.all_reduce(frames_done, op=dist.ReduceOp.SUM)
    total_frames = float(frames_done.item())

    # N > 1: parity of what the ranks produced and of what arrived at the root (outside the timed region).  Every rank checks the
    # track of its first frame against the oracle's track over [last frame of the previous shard | own first frame] (config 3:
    # that track exists only through the halo exchange); rank 0 checks records of EVERY rank's gathered shard against the oracle.
    parity_multi = None
    if multi and not args.no_cpu_baseline:
        counts_r = [shard_range(BATCH4, r_, world)[1] * max(1, args.inflight) if args.config == 4 else F for r_ in range(world)]

        real_cache = {}

        def pair_of(r_, i_):        # the stereo pair behind record i_ of rank r_ (the seeds / instants the ranks drew above)
            nu = min(counts_r[r_], args.unique_frames)
            if args.config == 3:
                return synth.make_stereo_pair(100, W, H, t=r_ * nu + (i_ % nu))
            if args.real_images:    # (the windows rank r_ cut from the photographs: the same seed gives the same windows)
                if r_ not in real_cache:
                    from pli_slam_amd import realdata
                    real_cache[r_] = realdata.frames_752x480(nu, seed=17 + r_, w=W, h=H)
                return real_cache[r_][i_ % nu]
            return synth.make_stereo_pair(r_ * nu + (i_ % nu), W, H)
        halo = {"checked": 0, "bad": 0}
        if track is not None and rank > 0 and F > 0:
            halo = halo_track_check(fe, cfg, track, d_halo_table, rec_bytes, pair_of(rank - 1, nuniq - 1),
                                    pose((rank * nuniq - 1) % (world * nuniq)), pose(rank * nuniq), W, H)
        hb = torch.tensor([halo["checked"], halo["bad"]], dtype=torch.int64, device=ctl)
        dist.all_reduce(hb, op=dist.ReduceOp.SUM)
        if rank == 0:
            parity_multi = gathered_parity_check(fe, cfg, gath.gathered(last_slot[0]), counts_r, rec_bytes, pair_of)
            parity_multi["halo_tracks_checked_against_oracle"] = int(hb[0].item())
            parity_multi["halo_tracks_mismatching"] = int(hb[1].item())
            parity_multi["ok"] = bool(parity_multi["ok"] and hb[1].item() == 0)

    rc = 0
    if rank == 0:
        b_frame, g = path_bytes_per_frame(W, H, cfg.orb_nlevels, cfg.orb_scale_factor, nfeat, nlines, cfg.lsd_scale)
        fps = total_frames / dt
        # dominant kernel by HIP-event time on the stream it ran on.  The ORB chain runs on a side stream beside the (longer) line
        # chain: the event times of its kernels include the time they wait for compute units behind the line kernels, so they
        # do not compete (alone — PLI_SIDE_MAX=0 — none of them comes near the growers)
        orb_chain = ("k_resize_level", "k_fast_cells", "k_octree", "k_blur_orb", "k_describe", "k_kp_counts")
        cand = {k: v for k, v in prof.items() if k not in orb_chain} or prof
        dom = max(cand.items(), key=lambda kv: kv[1][1]) if cand else (None, (0, 0.0))
        name, (calls, total_ms) = dom
        per_img = kernel_bytes_per_image(name, g, nfeat, nlines, W, H) if name else None
        # a relaxation grower is launched once per round; ideally the rounds of a step together touch the angle / modgrad planes
        # once, so the algorithmic bytes of ONE launch are that pass divided by the launches of the step
        if per_img is not None and name in ("k_tx_grow_sparse", "k_rx_grow", "k_rx_grow_big", "k_rx_grow_wave"):
            per_img = per_img / max(calls / max(args.steps, 1), 1.0)
        avg_s = (total_ms / max(calls, 1)) * 1e-3
        peak = 8000.0
        achieved = per_img * 2 * F / avg_s / 1e9 if (per_img is not None and avg_s > 0) else None
        issue = None       # what the dominant kernel is bound by, from the committed SQ passes of the same workload
        step_traffic = None  # counter traffic of the WHOLE step (every kernel x its launches) against the algorithmic bytes
        traffic = None     # HBM bytes per launch from the committed PMC passes of the same workload, if any
        sector = None      # the growers are gather kernels: their ceiling is the rate of random 128-byte line requests the chip
        counters_note = None   # why a counter-derived field is null, or which committed workload stood in for this one
        try:               # sustains (tools/probes/gather_rate.hip, profiles/r02_gather_rate_probe.txt: 49 G/s), not the stream peak
            tpath = [p_ for p_ in (os.path.join(ROOT, "profiles", "r%02d_traffic.json" % n_) for n_ in (6, 5, 4, 3, 2)) if os.path.exists(p_)][0]
            tr = json.load(open(tpath))
            wkey = "%dx%d_F%d" % (W, H, F)
            src_F = F
            if wkey not in tr["workloads"]:
                # no counter pass was committed for this batch size (a shard of config 4, another --frames-per-gpu): the nearest committed
                # workload of the SAME image size stands in, its per-launch counters scaled by the ratio of the batch sizes (the
                # kernels of the path are per-image work), and the line says so
                same = sorted((abs(int(k_.split("_F")[1]) - F), k_) for k_ in tr["workloads"] if k_.startswith("%dx%d_F" % (W, H)))
                if same:
                    wkey = same[0][1]
                    src_F = int(wkey.split("_F")[1])
                    counters_note = "no committed counter pass for F = %d: %s of %s scaled by %d / %d" % (F, wkey, os.path.basename(tpath), F, src_F)
                else:
                    counters_note = "no committed counter pass for %dx%d in %s" % (W, H, os.path.basename(tpath))
            scale_F = F / float(src_F)
            k = tr["workloads"].get(wkey, {}).get(name)
            if k and scale_F != 1.0:
                k = dict(k, fetch_kb=k["fetch_kb"] * scale_F, write_kb=k["write_kb"] * scale_F)
            # bytes = 2 x FETCH_SIZE + WRITE_SIZE for EVERY kernel: the L2 reads memory in 128-byte lines — streams and the growers'
            # random 16-byte gathers alike (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ for each kernel of the step and for the patterns of known
            # size of tools/probes/read_amp.hip: profiles/r05_counter_calibration.txt) — and FETCH_SIZE tallies a request as 64 bytes.
            # (Rounds 2-4 and the first round-5 lines counted the growers' requests as 64-byte sectors: their traffic was understated.)
            wl = tr["workloads"].get(wkey, {})
            if wl and all("launches" in v_ for v_ in wl.values()):
                tot = scale_F * sum(v_["launches"] * (2 * v_["fetch_kb"] + v_["write_kb"]) * 1024 for kn_, v_ in wl.items() if "alias_of" not in v_)
                step_traffic = {"bytes_per_step": tot, "algorithmic_bytes_per_step": b_frame * F, "ratio": tot / (b_frame * F),
                                "hbm_GBps_at_this_rate": tot / (dt / args.steps) / 1e9, "source": os.path.basename(tpath), "source_workload": wkey,
                                "note": "sum over the kernels of the committed FETCH_SIZE / WRITE_SIZE passes x their launches per step; "
                                        "every read request is a 128-byte line (2 x FETCH_SIZE), the growers' gathers included"}
            iq = tr.get("issue", {}).get(wkey, {}).get(name)
            if iq and iq["avg_ns"] > 0 and scale_F == 1.0:
                simd_quads = 1024 * iq["avg_ns"] * 2.4 / 4.0        # quad-cycles all SIMDs of the chip offer during one launch (2.4 GHz)
                issue = {"valu_issue_frac": iq["active_valu_quad_cycles"] / simd_quads, "valu_wave_instructions": iq["valu"],
                         "salu_wave_instructions": iq["salu"], "launch_ns_under_profiler": iq["avg_ns"], "source": os.path.basename(tpath),
                         "note": "SQ_ACTIVE_INST_VALU / (1024 SIMDs x quad-cycles of the launch): the share of the chip's VALU issue "
                                 "time the kernel uses (profiler run, 2.4 GHz assumed)"}
            if k:
                # FETCH_SIZE tallies every read request as 64 bytes on gfx950 (MI355X_MICROARCH.md: x2); the requests are 128-byte lines
                # for the growers' gathers as well (profiles/r05_counter_calibration.txt)
                traffic = (2 * k["fetch_kb"] + k["write_kb"]) * 1024
                if name.startswith(("k_tx_grow", "k_lsd_grow", "k_rx_grow")) and avg_s > 0:
                    rate = k["fetch_kb"] * 1024 / 64 / avg_s / 1e9      # requests per second (FETCH_SIZE = requests x 64 B)
                    sector = {"achieved": rate, "peak": 49.0, "unit": "G 128-byte line requests/s (L2 misses)", "frac": rate / 49.0,
                              "note": "peak measured by tools/probes/gather_rate.hip (random gathers: 49 G lines/s = 6.3 TB/s, the HBM "
                                      "rate a stream reaches too); requests per launch from the committed FETCH_SIZE pass"}
        except Exception as e:     # (the headline does not depend on the committed counter files; the line says what went wrong)
            counters_note = "counter files not usable: %r" % (e,)
        traffic_rate = (traffic / avg_s / 1e9) if (traffic and avg_s > 0) else None
        if args.real_images and traffic is not None:
            # the committed counter passes are of the SYNTHETIC batch: the bytes stand as an upper bound for these (calmer) frames, a rate
            # formed with this run's launch time would not be a measurement
            traffic_rate, sector = None, None
            counters_note = ((counters_note + "; ") if counters_note else "") + "counter passes are of the synthetic batch, not of these frames"
        # every kernel of the step against the HBM peak: algorithmic bytes per image (table above) x images per step / its time per step
        # (a kernel of the ORB chain is timed on the side stream, beside the line chain: its fraction is a lower bound)
        kfrac = {}
        for kn, (kc, kms) in prof.items():
            b_img = kernel_bytes_per_image(kn, g, nfeat, nlines, W, H)
            if kn == "k_resize_level":           # seven launches: level k reads level k - 1 and writes itself
                lv, sc, b_img = [], np.float32(1.0), 0
                for l in range(cfg.orb_nlevels):
                    inv = np.float32(1.0) / sc
                    lv.append(int(np.rint(np.float32(W) * inv)) * int(np.rint(np.float32(H) * inv)))
                    sc = np.float32(sc * np.float32(cfg.orb_scale_factor))
                b_img = sum(lv[l - 1] + lv[l] for l in range(1, len(lv)))
            if b_img and kms > 0:
                fr_ = b_img * 2 * F * args.steps / (kms * 1e-3) / 1e9 / peak
                # (above the peak = the launch did not do that work: the ordered-list kernels in the key mode of the tile relaxation
                # end at once unless an image is left to the sequential grower)
                if fr_ <= 1.0:
                    kfrac[kn] = round(fr_, 4)
        side = orb_chain + ("k_stereo_points", "k_stereo_median", "k_blur_lbd", "k_sobel")
        grower = bool(name) and name.startswith(("k_tx_grow", "k_lsd_grow", "k_rx_grow"))
        # `frac` is against the HBM peak by ALGORITHMIC bytes, as the contract asks.  A region grower is a chain of dependent random
        # gathers at the occupancy limit of 8 waves per SIMD: `bound` names that category; every 16-byte gather that misses the L2 moves a
        # 128-byte line, so its MEASURED traffic (`traffic`, `traffic_GBps`) is tens of times its algorithmic bytes and about half of what
        # the memory system delivers to gathers (`sector_requests`) — `limiter` names the three shares (the measurements behind them:
        # DESIGN.md 5); `issue`, `sector_requests` and `step_traffic` carry what the committed counter passes of this workload say —
        # null, with `counters_note`, when there are none
        roof = {"bound": "latency (wave slots)" if grower else "hbm", "kernel": name, "achieved": achieved, "peak": peak, "unit": "GB/s",
                "frac": (achieved / peak) if achieved is not None else None, "traffic": traffic, "traffic_GBps": traffic_rate, "sector_requests": sector,
                "limiter": "VALU issue / dependent trips at 8 waves per SIMD / 128-byte lines for 8-byte gathers (DESIGN.md 5)" if grower else "hbm",
                "counters_note": counters_note,
                "issue": issue, "step_traffic": step_traffic,
                "avg_launch_ms": avg_s * 1e3, "launches": calls,
                "kernel_launches_per_step": round(sum(v[0] for v in prof.values()) / max(args.steps, 1), 1),
                "path_achieved": fps / world * b_frame / 1e9, "path_frac": fps / world * b_frame / 1e9 / peak,
                "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in
                                       sorted(prof.items(), key=lambda kv: -kv[1][1]) if k not in side},
                "side_stream_kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in
                                                   sorted(prof.items(), key=lambda kv: -kv[1][1]) if k in side},
                "kernel_hbm_frac": dict(sorted(kfrac.items(), key=lambda kv: -kv[1])),
                "kernel_ms_note": "HIP-event time between the launch's two events on the stream it ran on, summed per step.  "
                                  "side_stream_kernel_ms_per_step: the ORB chain, the stereo point matcher and the LBD's blur + Sobel run on a "
                                  "low-priority side stream beside the line chain: their event times INCLUDE the wait for wave slots behind the "
                                  "line kernels (alone, PLI_SIDE_MAX=0, 256 frames: k_resize_level 0.7 ms, k_fast_cells 3.0, k_octree 1.5, "
                                  "k_blur_orb 1.0, k_describe 1.0) and are not comparable with the main stream's; the line chain's kernels are "
                                  "stretched by them in turn (alone: k_tx_round2 1.8 ms, k_tx_grow_sparse 3.0; the chain forks behind round 2's owner pass of a "
                                  "large batch).  DESIGN.md 5 / 7 have every kernel alone"}
        what = C_["what"] if args.config else ("752x480 stereo pairs, extract + stereo Hamming match (BASELINE configs[1] shape, "
                                               "batched)")
        out = {
            "metric": "stereo frames/sec (ORB+LSD extract+match), 752x480 EuRoC" if (W, H) == (752, 480) else
                      "stereo frames/sec (ORB+LSD extract+match), %dx%d" % (W, H),
            "value": fps, "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.config == 4 else "weak",
            "vs_baseline": None, "dtype": "u8",
            "data": ("real photographs (%d distinct 752x480 windows of the scikit-image sample photographs per GPU, enlarged 1.5-2.2x, "
                     "cycled; the right eye of all but the Middlebury pair is the left one displaced)" % nuniq) if args.real_images else
                    "synthetic (%d distinct seeded EuRoC-shaped pairs per GPU, cycled)" % nuniq,
            "config": {"workload": "%dxMI355X: %s; %dx%d, %d ORB kp (8 levels x1.2) + LSD/LBD (<=%d lines); %d stereo frames per "
                                   "GPU per step" % (world, what, W, H, nfeat, nlines, F),
                       "baseline_config": args.config or None, "frames_per_gpu": F, "bytes_per_frame": b_frame,
                       "parallelism": "frame-batch data parallel, %d rank(s), gather of result tables to rank 0" % world},
            "roofline": roof,
            "library": os.path.basename(getattr(fe.L, "_name", "?")),
        }
        rs = fe.lsd_round_stats()
        out["lsd_arena_words_per_pixel"] = fe.lsd_arena_words() if hasattr(fe.L, "pli_lsd_arena_words") else None
        out["lsd_rounds"] = {"launched_without_host_look": rs[0], "needed_by_slowest_image": rs[1], "images_redone_by_device_fallback": rs[2],
                             "note": "relaxation rounds per step; an image that has not settled after the launched rounds is redone on the "
                                     "device by the sequential grower (exact, slow): needed = -1 would flag it"}
        if multi:
            out["gather"] = {"bytes_per_rank_per_step": Fmax * rec_bytes, "bytes_at_root_per_step": world * Fmax * rec_bytes,
                             "backend": args.backend, "staged_through_host": bool(gath.staged)}
            # who took part: ranks counted by an all-reduce, the devices they ran on (N distinct ones, or the line says "rehearsal"),
            # every rank's own ms per step (the headline uses the slowest)
            out["gather"].update({"ranks_seen": ranks_ev["ranks_seen"], "devices": ranks_ev["devices"],
                                  "distinct_devices": ranks_ev["distinct_devices"], "one_device_per_rank": ranks_ev["one_device_per_rank"],
                                  "rehearsal_on_shared_device": ranks_ev["rehearsal_on_shared_device"]})
            out["per_rank_ms"] = ranks_ev["per_rank_ms"]
            if ranks_ev["ranks_seen"] != world or (world > 1 and not args.share_device and not ranks_ev["one_device_per_rank"]):
                out["value"] = None                     # (not the run the line claims to be)
                rc = 3
            if args.share_device:
                out["config"]["parallelism"] += "; ALL ranks share GPU 0 (rehearsal, not a scaling measurement)"
            if args.force_dist and world == 1:
                out["config"]["parallelism"] += "; --force-dist: the multi-rank code path in a process group of ONE rank"
            if parity_multi is not None:
                out["parity"] = parity_multi
                if not parity_multi["ok"]:
                    out["value"] = None
                    rc = 3
        if args.config == 4:
            out["config"]["batch_frames"] = BATCH4
            out["config"]["batches_in_flight"] = max(1, args.inflight)
        if multi and not args.no_cpu_baseline:
            # the multi-rank line carries the CPU baseline too (rule d): rank 0 times the oracle on a sample of ITS shard after the
            # timed region; the other ranks wait at the closing barrier below
            out["cpu_baseline"] = cpu_baseline(images, bytes(cfg), budget_s=args.cpu_baseline_seconds)
        if not multi and not args.no_cpu_baseline:
            # BASELINE.json configs[1] (a single stereo pair) beside the batch: latency of one pair through the same library
            if F > 1 and (W, H) == (752, 480):
                f1 = Frontend(capi.default_config(W, H, orb_nfeatures=nfeat, lsd_nfeatures=nlines, max_frames=1), device=local_rank, dev=False)
                f1.set_stream(torch.cuda.current_stream().cuda_stream)
                t1 = torch.zeros(rec_bytes, dtype=torch.uint8, device=dev)
                for rep in range(12):
                    if rep == 2:
                        torch.cuda.synchronize(); ts = time.perf_counter()
                    f1.batch_run_device(1, d_left.data_ptr(), d_right.data_ptr(), W, W * H, t1.data_ptr())
                torch.cuda.synchronize()
                out["single_pair"] = {"ms": (time.perf_counter() - ts) / 10 * 1e3, "note": "one stereo pair per call, 10 calls"}
                # ... and the shape System::TrackStereo issues: one Frame from HOST images through pli_frame_extract (what the C++
                # adapters fuse the four extractor threads of Frame.cc:128-135 into), record back on the host
                for rep in range(12):
                    if rep == 2:
                        ts = time.perf_counter()
                    f1.frame_extract(images[rep % nuniq, 0], images[rep % nuniq, 1])
                out["single_pair"]["frame_extract_host_ms"] = (time.perf_counter() - ts) / 10 * 1e3
                del f1
            if not args.no_host_leg:
                out["host_inclusive"] = host_inclusive_leg(fe, images, F, nuniq, W, H, rec_bytes, args.steps)
            out["cpu_baseline"] = cpu_baseline(images, bytes(cfg), budget_s=args.cpu_baseline_seconds)
            # the table the last timed step left in d_table is the one that is checked
            out["parity"] = parity_check(fe, cfg, d_table, images, F, nuniq, rec_bytes)
            if not out["parity"]["ok"]:
                out["value"] = None                   # a fast path whose results differ from the reference's is not a result
                rc = 3
            if not args.no_large_batch_leg and not args.config and (W, H) == (752, 480) and F < LARGE_BATCH:
                # the same path at the batch size where the throughput of one GPU levels off (the sequential LSD schedule takes
                # over above 1280 frames per call; ~236 GB of the 288 GB HBM): informational, never the headline value
                try:
                    del fe
                    out["large_batch"] = large_batch_leg(capi, Frontend, nfeat, nlines, args.lsd_mode, local_rank, d_uniq, nuniq, W, H)
                except Exception as e:                # (the headline line must not depend on this leg)
                    out["large_batch"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


LARGE_BATCH = 2048


def large_batch_leg(capi, Frontend, nfeat, nlines, lsd_mode, device, d_uniq, nuniq, W, H, steps=3):
    """One GPU, LARGE_BATCH stereo frames per step (images resident in HBM, cycled from the same distinct pairs)."""
    import torch
    dev = d_uniq.device
    F = LARGE_BATCH
    fe = Frontend(capi.default_config(W, H, orb_nfeatures=nfeat, lsd_nfeatures=nlines, max_frames=F, lsd_mode=lsd_mode), device=device, dev=False)
    fe.set_stream(torch.cuda.current_stream().cuda_stream)
    d_img = d_uniq[torch.arange(F, device=dev) % nuniq].contiguous()
    d_left, d_right = d_img[:, 0].contiguous(), d_img[:, 1].contiguous()
    del d_img
    table = torch.zeros(F * int(fe.layout.record_bytes), dtype=torch.uint8, device=dev)
    fe.batch_run_device(F, d_left.data_ptr(), d_right.data_ptr(), W, W * H, table.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fe.batch_run_device(F, d_left.data_ptr(), d_right.data_ptr(), W, W * H, table.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames_per_gpu": F, "value": F * steps / dt, "unit": "stereo frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "note": "same path and shapes, %d stereo frames per step on one GPU (sequential LSD schedule)" % F}


def host_inclusive_leg(fe, images, F, nuniq, W, H, rec_bytes, steps):
    """The same step with the images starting in pinned HOST memory and the tables ending there: the library's
    pipelined host entry point (pinned staging, copy stream: H2D of batch i+1 and D2H of table i-1 overlap the kernels of i)."""
    idx = np.arange(F) % nuniq
    left, right, table = fe.host_buffers(F)
    left[:] = images[idx, 0].reshape(F, -1)
    right[:] = images[idx, 1].reshape(F, -1)
    n = max(2, steps)
    fe.host_submit(F, left, right, table[0]); fe.host_wait()          # warm-up
    t0 = time.perf_counter()
    for i in range(n):
        fe.host_submit(F, left, right, table[i & 1])
    fe.host_wait_all()
    dt = time.perf_counter() - t0
    return {"value": F * n / dt, "unit": "stereo frames/s", "steps": n,
            "note": "images in pinned host memory, tables returned to pinned host memory, double buffered over PCIe"}


if __name__ == "__main__":
    main()
