"""
bench.py's output contract (one JSON line with the driver's keys + `roofline` + `cpu_baseline`) on a reduced
workload, and the N = 2 launch exactly as the driver issues it (torch.distributed.run, one process per rank) with
both ranks sharing the one GPU of the test box over gloo -- the path where a rank-0-only collective once deadlocked.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_single_gpu_line_has_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--blocks", "2", "--batch", "1",
           "--kernel-iters", "2", "--model", "tiny"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _last_json(p.stdout)
    assert KEYS <= set(r), KEYS - set(r)
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["unit"] == "images/s" and r["value"] > 0 and r["dtype"] == "bf16" and r["vs_baseline"] is None
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"], cb
    assert "whole training step" in r["metric"] and r["config"]["finite"] and r["config"]["adapted_modules"] == 8
    assert r["adapter_path"]["value"] > 0 and r["adapter_path"]["no_recompute"]["value"] > 0
    assert r["adapter_path"]["cpu_port"]["value"] > 0
    assert set(r["phases_ms"]) == {"forward", "matching (host LSAP)", "loss", "backward", "exchange + AdamW"}
    assert r["distributed"]["rank_device_ids"] == [0] and 0 < r["mfma_bound"]["frac"] < 1


@pytest.mark.gpu
def test_two_ranks_launched_like_the_driver_complete():
    port = 29600 + os.getpid() % 1000
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--blocks", "2", "--batch", "1", "--kernel-iters", "2", "--model", "tiny", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    r = _last_json(p.stdout)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert r["adapter_path"]["no_recompute"]["value"] > 0 and r["config"]["finite"]
    assert r["exchange_overlap"]["world"] == 2 and r["exchange_overlap"]["step_ms_exposed"] > 0
    # the per-bucket timeline (HIP events on the reducer's side stream against the end of backward): every bucket once, in index order
    tl = r["exchange_overlap"]["buckets_rank0"]
    assert [b["bucket"] for b in tl] == list(range(r["exchange_overlap"]["collectives_per_step"])) and len(tl) >= 1
    assert all(b["end_ms_after_backward_end"] >= b["start_ms_after_backward_end"] and b["bytes"] > 0 for b in tl)
    assert r["distributed"]["rank_device_ids"] == [0, 0] and r["distributed"]["backend"] == "gloo"
    assert sum(1 for l in p.stdout.splitlines() if l.startswith("{")) == 1      # rank 0 only


def test_gpus_flag_disagreeing_with_world_size_is_refused():
    """a line must never describe a different job size than the one launched."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bare_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher re-executes under torch.distributed.run (two ranks on the one GPU of the
    test box over gloo) and the line says n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "2",
           "--batch", "1", "--kernel-iters", "2", "--model", "tiny", "--no-cpu-baseline", "--no-fp8", "--no-overlap"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    r = _last_json(p.stdout)
    assert r["n_gpus"] == 2 and r["distributed"]["rank_device_ids"] == [0, 0] and r["config"]["global_batch"] == 2


def test_workload_label_names_the_configuration_that_runs():
    """`config.workload` is built on the CPU path of the contract too: the default is configs[1], r = 8 / batch 16 / fp8 is
    configs[4]'s mode, anything else is called a variation (a formatting slip here once broke every bench run)."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    base = dict(rank=16, batch=8, fp8_frozen=False, model="sam3", act_dtype="bf16", match_twice=False)
    a = bench.workload_label(argparse.Namespace(**base), 64, False)
    assert "r=16 alpha=32 (BASELINE configs[1])" in a and "batch 8/GPU" in a and "64 ViT-MLP" in a and "kept in HBM" in a
    b = bench.workload_label(argparse.Namespace(**dict(base, rank=8, batch=16, fp8_frozen=True)), 64, False)
    assert "configs[4]" in b and "r=8 alpha=16" in b and b.endswith("e5m2 gradients)")
    c = bench.workload_label(argparse.Namespace(**dict(base, batch=4, model="tiny", match_twice=True)), 8, True)
    assert "variation of BASELINE configs[1]" in c and c.startswith("[TINY-WIDTH") and "twice" in c and "on (per block" in c
