"""The multi-rank path on REAL kernels, rehearsed on the one GPU of the test box (-m gpu): `bench.py --gpus 2 --backend gloo
--share-device` starts two rank processes that both run their shard on GPU 0 — real `step()`, real `TableGatherer` and
`exchange_halo` (the tables staged through the host, gloo moves CPU tensors only) — and rank 0 checks records of EVERY rank's
gathered shard against the oracle; with --config 3 every rank also checks the track that exists only through the halo
exchange against the oracle's track over [last frame of the previous shard | own first frame].  The line a future 8-GPU run
prints has the same keys (`n_gpus`, `gather`, `parity`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
pytestmark = pytest.mark.gpu


def run_bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "2", "--warmup", "1",
                          "--cpu-baseline-seconds", "3", *extra], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_config4_two_ranks_real_kernels_gather_and_parity():
    d = run_bench("--config", "4")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["frames_per_gpu"] == 128 and d["config"]["batch_frames"] == 256
    assert d["gather"]["backend"] == "gloo" and d["gather"]["staged_through_host"] is True
    assert d["gather"]["bytes_at_root_per_step"] == 2 * d["gather"]["bytes_per_rank_per_step"] > 2 * 128 * 100_000
    p = d["parity"]
    assert p["ok"] is True and p["ranks_checked"] == 2 and p["records_gathered"] == 256 and p["empty_records"] == 0
    assert p["pairs_checked_against_oracle"] == 4 and p["pairs_mismatching"] == 0
    # the line proves who took part: two ranks counted by an all-reduce, their devices gathered — ONE distinct device here
    # (--share-device), so the line calls itself a rehearsal —, every rank's own time
    g_ = d["gather"]
    assert g_["ranks_seen"] == 2 and len(g_["devices"]) == 2 and g_["distinct_devices"] == 1
    assert g_["one_device_per_rank"] is False and g_["rehearsal_on_shared_device"] is True
    pr = d["per_rank_ms"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] == pr["all"][pr["slowest_rank"]] and pr["min"] > 0


def test_config3_two_ranks_halo_tracks_equal_the_oracle():
    d = run_bench("--config", "3", "--frames-per-gpu", "6")
    assert d["n_gpus"] == 2 and d["value"] > 0
    p = d["parity"]
    assert p["ok"] is True and p["ranks_checked"] == 2 and p["pairs_mismatching"] == 0 and p["pairs_checked_against_oracle"] == 4
    assert p["halo_tracks_checked_against_oracle"] == 1 and p["halo_tracks_mismatching"] == 0      # rank 1's first frame


def test_default_workload_two_ranks_weak_scaling_line():
    d = run_bench("--frames-per-gpu", "16")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["parity"]["ok"] is True
    assert d["parity"]["records_gathered"] == 32
    # the multi-rank line is complete by the bench contract: every key of the single-rank line, the CPU baseline
    # (rank 0 times the oracle after the timed region) and the roofline object with its counter fields or the note that says why not
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity", "gather"):
        assert k in d, k
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and cb["sample"]
    rf = d["roofline"]
    # (two processes share the GPU here: which kernel dominates a rank's 16-frame step varies, and some — region2rect — have no
    # algorithmic-byte figure: the fields must be there and sane, not positive)
    assert rf["bound"] and rf["kernel"] and rf["achieved"] is not None and rf["peak"] == 8000.0 and 0 <= rf["frac"] < 1
    assert rf["traffic"] is not None or rf["counters_note"]


# ---- the RCCL calls themselves -------------------------------------------------------------------------------
# `backend="nccl"` is RCCL on ROCm.  The two-rank rehearsals above use gloo with host staging because RCCL refuses two ranks on one
# device; what they cannot show is that the calls the 8-GPU run makes — init_process_group("nccl", device_id=...), all_reduce and
# barrier on device tensors, the asynchronous gather of DEVICE tables by TableGatherer, the halo walk — execute at all.  --force-dist
# runs the multi-rank code path of bench.py in a process group of ONE rank on RCCL.

def run_nccl_one_rank(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--backend", "nccl", "--force-dist", "--steps", "2", "--warmup", "1",
                          "--cpu-baseline-seconds", "3", *extra], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_rccl_path_executes_config4_gather_of_device_tables():
    d = run_nccl_one_rank("--config", "4", "--inflight", "1")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["frames_per_gpu"] == 256
    assert d["gather"]["backend"] == "nccl" and d["gather"]["staged_through_host"] is False
    p = d["parity"]                      # records of the table that CAME OUT of the RCCL gather against the oracle
    assert p["ok"] is True and p["records_gathered"] == 256 and p["empty_records"] == 0 and p["pairs_mismatching"] == 0


def test_rccl_path_executes_config3_halo_walk_and_track():
    d = run_nccl_one_rank("--config", "3", "--frames-per-gpu", "6")
    assert d["gather"]["backend"] == "nccl" and d["parity"]["ok"] is True and d["parity"]["pairs_mismatching"] == 0
    assert d["parity"]["halo_tracks_checked_against_oracle"] == 0      # one rank: no predecessor, the halo walk finds no neighbour
