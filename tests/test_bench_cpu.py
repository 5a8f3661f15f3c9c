"""bench.py plumbing that needs no GPU: `--gpus N` without a launcher starts its own ranks (the driver's
command form is `python bench.py --gpus N ...`), the process group / grouped all-gather / max-over-ranks timing run over
gloo, and rank 0 prints exactly one JSON line carrying the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=e)
    return res


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config"}


def test_gpus_2_self_spawns_ranks_without_a_launcher():
    res = _run(["--gpus", "2", "--dry-run", "--steps", "10", "--warmup", "3", "--gather-group", "4", "--regions", "2", "--batch", "64"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                       # rank 0 only
    line = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(line)
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["dry_run"] is True
    assert line["config"]["global_batch"] == 128
    # 3 warm-up + 2 regions x 10 steps in groups of 4, a partial group flushed at every fence
    assert line["config"]["collectives"] >= (3 + 20) // 4
    assert line["config"]["groups_seen_by_sink"] == line["config"]["collectives"]
    assert line["config"]["sink_content_ok"] is True


def test_gpus_8_dry_run_with_ragged_groups():
    """the 8-rank shape of the N>1 loop without a node -- eight gloo ranks, groups of 5 steps against
    regions of 13 (every fence flushes a ragged group of 3), every rank's sink checks every peer's slice of every batch."""
    res = _run(["--gpus", "8", "--dry-run", "--steps", "13", "--warmup", "3", "--gather-group", "5", "--regions", "2", "--batch", "32"], timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 8 * 32 and line["config"]["group"] == 5
    # warm-up: 3 steps -> one ragged group; each region: 13 steps -> 5 + 5 + 3
    assert line["config"]["collectives"] == 1 + 2 * 3
    assert line["config"]["groups_seen_by_sink"] == line["config"]["collectives"]
    assert line["config"]["sink_content_ok"] is True


def test_strong_scaling_dry_run_shards_one_global_batch():
    """VERDICT r03 "missing" 3: `--scaling strong` -- a GLOBAL batch every rank holds, row-sharded, score slices all-gathered
    (north_star's partitioning; RowShardedPredictor) -- over four gloo ranks: the line says "strong", the global batch is the
    one asked for, and every rank saw every row's score."""
    res = _run(["--gpus", "4", "--dry-run", "--scaling", "strong", "--steps", "6", "--warmup", "2", "--gather-group", "3", "--regions", "1", "--batch", "256"])
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 4
    assert line["config"]["global_batch"] == 256 and line["config"]["batch_per_gpu"] == 64
    assert line["config"]["strong_row_shard_gather_ok"] is True and line["config"]["sink_content_ok"] is True
    res = _run(["--gpus", "4", "--dry-run", "--scaling", "strong", "--batch", "250"])
    assert res.returncode != 0 and "does not divide" in (res.stderr + res.stdout)


def test_single_rank_dry_run_and_world_size_mismatch():
    res = _run(["--dry-run", "--steps", "5", "--warmup", "1", "--regions", "1"])
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    # launched BY a launcher (RANK set) with a world size that contradicts --gpus: refuse instead of mis-reporting n_gpus
    res = _run(["--gpus", "2", "--dry-run"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)


def test_bench_without_gpu_refuses_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = _run(["--steps", "2", "--warmup", "1", "--cpu-seconds", "0"])
    assert res.returncode != 0 and "HIP device" in (res.stderr + res.stdout)
