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
# ---- ShardPipeline.run ITSELF (what bench.py --gpus N times): two lanes, eight chunks, the staging ring reused, both forms ----
from tsfresh_amd.distributed import ShardPipeline
class HostPlan:
    """extract_device on host tensors: the emulation build stands in for the kernels; pointers are mapped back to rows"""
    def __init__(self, values_t, offsets_t, full_t):
        self.v, self.o, self.full = values_t, offsets_t, full_t
    def set_length_hint(self, lo, hi): pass
    def close(self): pass
    def extract_device(self, vptr, dtype, optr, n, out_ptr, ld, stream):
        c0 = (optr - self.o.data_ptr()) // 8
        off = self.o.numpy()[c0:c0 + n + 1]
        block = emul_extract(params, self.v.numpy()[off[0]:off[-1]], off - off[0])[1]
        r0 = (out_ptr - self.full.data_ptr()) // (8 * ld)
        self.full[r0:r0 + n] = torch.from_numpy(np.ascontiguousarray(block))
def run_pipeline(all_lens, n_chunks, p2p_only=False):
    all_off = np.zeros(len(all_lens) + 1, dtype=np.int64); np.cumsum(all_lens, out=all_off[1:])
    all_vals = np.random.default_rng(5).standard_normal(int(all_off[-1]))
    want = fn(all_vals, all_off)
    bounds = shard_bounds(all_lens, world)
    counts = [int(c) for c in np.diff(bounds)]
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    v = torch.from_numpy(all_vals[all_off[lo]:all_off[hi]].copy())
    o = torch.from_numpy((all_off[lo:hi + 1] - all_off[lo]).copy())
    full = torch.full((len(all_lens), 10), -5.0, dtype=torch.float64)
    pipe = ShardPipeline(None, 10, None, dist=dist, n_chunks=n_chunks, plan_factory=lambda: HostPlan(v, o, full))
    pipe.p2p_only = p2p_only
    for _ in range(2):   # a second run reuses the staging blocks of the first
        pipe.run(v, o, counts, full, 1)
        assert np.array_equal(full.numpy(), want), (rank, np.argwhere(full.numpy() != want)[:4])
        full.fill_(-5.0) if _ == 0 else None
    stats = (pipe.exchanges_issued, pipe.ring_reuses, counts)
    pipe.close()
    return stats
issued, reuses, counts = run_pipeline(np.full(16 * world, 20), 8)            # equal heights: all-gather + ring
assert len(set(counts)) == 1 and issued == 8 and reuses == 4, (issued, reuses, counts)
issued, reuses, counts = run_pipeline(lens, 8)                                # very unequal heights (an empty shard at world 4)
assert len(set(counts)) > 1, (issued, reuses, counts)
lens2 = np.random.default_rng(9).integers(10, 60, size=240); lens2[[5, 77, 150, 201]] = 150
issued, reuses, counts = run_pipeline(lens2, 8)                               # sum(len^2)-balanced, unequal: point to point, ring reused
assert len(set(counts)) > 1 and min(counts) >= 8 and reuses >= 1, (issued, reuses, counts)
issued, reuses, counts = run_pipeline(np.full(16 * world, 20), 8, p2p_only=True)
assert issued > 8 or world == 2, (issued, world)
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
