"""bench.py as the driver starts it: `--gpus N` must mean N ranks.  CPU-only (gloo, --dry-tables: sharding + gather of the
result tables without kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus2_self_launches_two_ranks_and_root_receives_both_shards():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry-tables", "--config", "4",
                          "--steps", "2", "--warmup", "1"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["shards_ok"] is True
    assert d["batch_frames"] == 256 and d["shard_frames"] == [128, 128]          # config 4: the 256-frame batch over the ranks
    assert d["gathered_bytes_per_step"] == 256 * 4096


def test_gpus8_config4_is_32_frames_per_rank_and_256_records_at_the_root():
    """SURVEY 8e / BASELINE configs[3] exactly: `bench.py --gpus 8 --config 4` cuts the 256-frame batch into 32 frames per rank and the
    root receives all 256 records (eight gloo ranks on the CPU, dry tables)."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--dry-tables", "--config", "4",
                          "--steps", "2", "--warmup", "1"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["shards_ok"] is True
    assert d["batch_frames"] == 256 and d["shard_frames"] == [32] * 8
    # the line proves who took part: eight ranks counted by an all-reduce, eight distinct identities, every rank's own time
    r = d["ranks"]
    assert r["ranks_seen"] == 8 and r["distinct_devices"] == 8 and r["one_device_per_rank"] and not r["rehearsal_on_shared_device"]
    assert len(r["per_rank_ms"]["all"]) == 8 and r["per_rank_ms"]["max"] == max(r["per_rank_ms"]["all"])
    assert r["per_rank_ms"]["all"][r["per_rank_ms"]["slowest_rank"]] == r["per_rank_ms"]["max"]
    assert d["gathered_bytes_per_step"] == 256 * 4096


def test_gpus_must_match_world_size():
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--dry-tables"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr


def test_three_uneven_ranks_under_torchrun():
    """Started the way the driver does (torch.distributed.run), 3 ranks: shards of 86/85/85 frames."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                          "127.0.0.1", "--master-port", "29613", BENCH, "--gpus", "3", "--backend", "gloo", "--dry-tables",
                          "--config", "4", "--steps", "1", "--warmup", "0"], env=_clean_env(), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["shards_ok"] and d["shard_frames"] == [86, 85, 85]


def test_force_dist_one_rank_walks_the_gather_path():
    """--force-dist (the switch tests/test_multirank_gpu.py uses to execute the RCCL calls on the one GPU of the test box): a process
    group of ONE rank still issues the gather; here with gloo and dry tables."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--backend", "gloo", "--dry-tables", "--force-dist", "--steps", "2",
                          "--warmup", "1", "--frames-per-gpu", "5"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["shards_ok"] is True and d["shard_frames"] == [5]
