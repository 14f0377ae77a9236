"""The exchange forms of tsfresh_amd/distributed.py over RCCL on the one GPU a test box has (VERDICT r4 "Next" #10): the
all-gather form runs at a world of one through bench.py (tests/test_bench_launch.py); this is the point-to-point form --
what ranks whose shards differ in height use -- through RCCL's self send / receive.  World sizes 2 and 4 of both forms run
over gloo in tests/test_distributed_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_point_to_point_exchange_over_rccl_on_one_gpu(gpu):
    from bench import free_port
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "rccl_loopback_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    doc = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert doc["same"] and not doc["untouched_cells"], doc
    # ShardPipeline.run with the exchange forced at a world of one: 8 chunks, both forms, the staging ring reused
    for form in ("all_gather", "p2p"):
        d = doc["pipeline"][form]
        assert d["equal"] and d["exchanges"] >= 8 and d["ring_reuses"] == 4, (form, d)
    assert doc["pipeline"]["p2p"]["loopback_bytes_equal"], doc
