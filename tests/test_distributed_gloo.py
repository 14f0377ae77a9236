"""world_size-2 / -4 gloo tests of the id-sharding + exchange path (runs on CPU; the per-shard extraction is the emulated
kernel build, the collective code is exactly what the GPU path uses)."""
import os
import subprocess
import sys

import numpy as np

from tsfresh_amd.distributed import shard_bounds

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%(here)r, ".."))
sys.path.insert(0, %(here)r)
import torch, torch.distributed as dist
from tsfresh_amd.distributed import chunk_cuts, exchange_rows, extract_sharded, finish_exchange, shard_bounds
from tsfresh_amd.feature_extraction.settings import MinimalFCParameters
from emul_lib import emul_extract
dist.init_process_group(backend="gloo")
world, rank = dist.get_world_size(), dist.get_rank()
rng = np.random.default_rng(0)
# ragged: a few long series among many short ones -> sum(len^2)-balanced shards of very different heights
lens = rng.integers(1, 60, size=61)
lens[[3, 9]] = 400
values = rng.standard_normal(int(lens.sum()))
offsets = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=offsets[1:])
heights = np.diff(shard_bounds(lens, world))
assert heights.max() > 2 * max(heights.min(), 1) or world == 1, heights  # the case padding-to-the-tallest would hurt
params = MinimalFCParameters()
fn = lambda v, o: emul_extract(params, v, o)[1]
ref = fn(values, offsets)
for n_chunks in (1, 3):
    full = extract_sharded(fn, values, offsets, 10, dist=dist, n_chunks=n_chunks)
    assert full.shape == ref.shape and np.array_equal(full, ref), (full.shape, ref.shape)
# the equal-height form: all_gather_into_tensor through a staging block + scatter (what bench.py --gpus N runs)
rows, n_cols, n_chunks = 12, 7, 4
starts = [r * rows for r in range(world + 1)]
full = torch.full((world * rows, n_cols), float("nan"), dtype=torch.float64)
full[starts[rank]:starts[rank + 1]] = torch.arange(rows * n_cols, dtype=torch.float64).reshape(rows, n_cols) + 1000 * rank
cuts = chunk_cuts(rows, n_chunks)
stage = torch.empty((world * (cuts[1] - cuts[0] + 1), n_cols), dtype=torch.float64)
for c in range(n_chunks):
    lo_hi = [(cuts[c], cuts[c + 1])] * world
    finish_exchange(full, starts, rank, exchange_rows(full, starts, rank, lo_hi, dist, stage))
want = torch.cat([torch.arange(rows * n_cols, dtype=torch.float64).reshape(rows, n_cols) + 1000 * r for r in range(world)])
assert torch.equal(full, want)
if rank == 0:
    print("GLOO_OK", ref.shape, world, list(heights))
dist.destroy_process_group()
'''


def test_shard_bounds_balance_and_cover():
    lens = np.array([10, 10, 10, 100, 10, 10, 10, 10])
    b = shard_bounds(lens, 2)
    assert b[0] == 0 and b[-1] == len(lens) and np.all(np.diff(b) >= 0)
    # the long series dominates sum(len^2): it must not share a shard with everything else
    cost = lens.astype(float) ** 2
    parts = [cost[b[i]:b[i + 1]].sum() for i in range(2)]
    assert max(parts) <= 0.999 * cost.sum()
    assert list(shard_bounds(np.ones(8), 4)) == [0, 2, 4, 6, 8]
    assert list(shard_bounds([], 3)) == [0, 0, 0, 0]


def _run_ranks(tmp_path, nproc):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"here": HERE})
    import socket
    import emul_lib
    emul_lib.load()  # build the emulation library once, before the ranks start
    # loopback only, a free port per attempt; gloo's rendezvous occasionally mis-counts its peers on a busy box
    # ("connected to 1 peer ranks. Expected ... 1"), which is a property of the transport, not of the code under test
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo")
    out = None
    for attempt in range(3):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                             capture_output=True, text=True, env=env, timeout=600)
        if out.returncode == 0 or "peer ranks" not in (out.stdout + out.stderr):
            break
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_OK (61, 10) %d" % nproc in out.stdout, out.stdout[-500:]


def test_two_rank_gloo_exchange_matches_single_process(tmp_path):
    _run_ranks(tmp_path, 2)


def test_four_rank_gloo_exchange_of_unequal_ragged_shards(tmp_path):
    """World size 4, sum(len^2)-balanced shards of unequal height (point-to-point exchange, no padding), 1 and 3 row
    chunks, plus the equal-height all-gather + scatter form."""
    _run_ranks(tmp_path, 4)
