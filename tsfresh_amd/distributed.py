"""Multi-GPU extraction: ids sharded across ranks, feature matrix reassembled with one all-gather.

Every (id, kind) series is independent (the reference already maps them independently:
tsfresh/utilities/distribution.py:24-44), so the only exchange step is reassembling the `[n_ids x n_cols]` matrix.
One process per GPU, `torch.distributed` (backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests).  The
shards are contiguous blocks of the id-sorted series list, balanced by sum(len^2) when ragged (the O(L^2) entropy
sweep dominates the cost), and the collective is a single variable-size all-gather of float64 rows.
"""
import numpy as np


def shard_bounds(lengths, world_size, cost_power=2.0):
    """Contiguous shards of the series list with ~equal sum(len ** cost_power).  -> int64 [world_size + 1]."""
    lengths = np.asarray(lengths, dtype=np.float64)
    n = len(lengths)
    if n == 0:
        return np.zeros(world_size + 1, dtype=np.int64)
    cost = np.cumsum(lengths ** cost_power)
    targets = cost[-1] * np.arange(1, world_size) / world_size
    cuts = np.searchsorted(cost, targets, side="left") + 1
    bounds = np.concatenate([[0], np.minimum(cuts, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def all_gather_rows(local, counts, dist, device=None):
    """All-gather row blocks of different heights.  `local`: torch tensor [n_local, n_cols]; counts: rows per rank."""
    import torch
    world = dist.get_world_size()
    n_cols = local.shape[1]
    maxrows = int(max(counts)) if len(counts) else 0
    pad = torch.zeros((maxrows, n_cols), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    gathered = torch.empty((world * maxrows, n_cols), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, pad)
    parts = [gathered[r * maxrows: r * maxrows + int(counts[r])] for r in range(world)]
    return torch.cat(parts, dim=0)


def extract_sharded(extract_fn, values, offsets, n_cols, dist=None, torch_device=None):
    """Run `extract_fn(values_shard, offsets_shard) -> ndarray [n_local, n_cols]` on this rank's shard and return
    the full matrix (every rank gets it).  With dist=None (single process) this is just extract_fn."""
    import torch
    offsets = np.asarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    if dist is None or dist.get_world_size() == 1:
        return extract_fn(values, offsets)
    world, rank = dist.get_world_size(), dist.get_rank()
    bounds = shard_bounds(np.diff(offsets), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sub_off = offsets[lo:hi + 1]
    local = extract_fn(values[sub_off[0]:sub_off[-1]], sub_off - sub_off[0]) if hi > lo else np.empty((0, n_cols))
    dev = torch_device if torch_device is not None else "cpu"
    t = torch.as_tensor(np.ascontiguousarray(local), dtype=torch.float64, device=dev).reshape(-1, n_cols)
    full = all_gather_rows(t, np.diff(bounds), dist)
    assert full.shape[0] == n
    return full.cpu().numpy()
