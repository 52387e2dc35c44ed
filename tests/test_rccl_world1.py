"""The data-parallel exchange on the real backend: "nccl" = RCCL, world size 1 on the one GPU of the test box -- enough
to load RCCL, create the communicator and run the reducer's side-stream all-reduces.  Runs in a
subprocess so that the pytest process keeps no default process group."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_reducer_runs_on_rccl_with_world_size_1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_worker.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["backend"] == "nccl" and r["world"] == 1 and r["rccl_version"][0].isdigit()
    assert r["params_unchanged_by_broadcast"] and r["buckets"] >= 2
    # one message per bucket, launched in index order from the hooks; the "used" flags ride in the last one (globally unused
    # parameters keep grad None, as under torch DDP)
    assert r["allreduce_calls"] == r["buckets"] and r["all_on_side_stream_async"]
    assert r["launch_log"] == [[b, "hook"] for b in range(r["buckets"])]
    assert r["flags"] and all(f == 1.0 for f in r["flags"])
    assert r["elements_reduced"] == r["flat_elements"]
    assert r["grads_equal"] and r["scalar"] == 5.0
    # launch positions on the GPU's clock: every bucket is launched from a hook, in index order, and all but the last START before the
    # backward has ended (negative = earlier than the end-of-backward event) -- the exchange rides behind the remaining backward
    assert r["deep_buckets"] >= 3 and r["deep_launch_log"] == [[b, "hook"] for b in range(r["deep_buckets"])]
    starts = r["deep_bucket_start_ms_after_backward_end"]
    assert len(starts) == r["deep_buckets"] and all(s < 0 for s in starts[:-1]), starts
    assert starts == sorted(starts), starts
