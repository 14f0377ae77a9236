"""Multi-GPU extraction: ids sharded across GPUs, the feature matrix reassembled in id order.

Every (id, kind) series is independent (the reference already maps them independently:
tsfresh/utilities/distribution.py:24-44), so the only exchange step is reassembling the `[n_ids x n_cols]` matrix.
Shards are contiguous blocks of the id-sorted series list, balanced by sum(len^2) when ragged (the O(L^2) entropy
sweep dominates the cost), so their HEIGHTS differ; nothing here pads to the tallest shard.

Two front doors:

* one process per GPU (`torch.distributed`; backend "nccl" is RCCL on ROCm, "gloo" on CPU for the tests):
  `ShardPipeline` extracts this rank's shard in row chunks on two alternating launch streams and exchanges every
  finished chunk while the next one is being extracted -- `bench.py --gpus N` times exactly this object;
  `extract_sharded` is the host-array convenience around it.
* one process, several GPUs: `extract_on_devices` (reached through `extract_features(..., devices=[...])`): one plan
  and one host thread per device (the C-ABI call releases the GIL), each device's chunked H2D / kernels / D2H pipeline
  writing straight into its row slice of ONE page-locked result matrix.

The full matrix is rank-major: rows [starts[r], starts[r + 1]) belong to rank r.  The exchange of a chunk is
  * equal heights on every rank: one `all_gather_into_tensor` into a staging block + a device-side scatter into the rank
    slices (626 MB of copies at HBM speed cost < 0.5 ms; the collective itself is RCCL's all-gather over xGMI);
  * unequal heights: grouped point-to-point `isend` / `irecv` straight between the rank slices -- on the fully
    connected xGMI mesh every block crosses exactly one link, no padding, no staging.
"""
import collections
import threading

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


def chunk_cuts(n_rows, n_chunks):
    """Row cuts of a shard of `n_rows` rows in `n_chunks` chunks -- the same formula on every rank, so each rank knows
    every other rank's chunk boundaries from the shard heights alone."""
    n_chunks = max(1, int(n_chunks))
    return [int(n_rows) * c // n_chunks for c in range(n_chunks + 1)]


def exchange_rows(full, starts, rank, lo_hi, dist, stage=None, loopback=None, force=False):
    """Exchange one row chunk of every rank's slice of `full` (torch tensor [sum(counts), n_cols], rank-major).

    loopback: test hook for a box with one GPU -- this rank's own rows ALSO travel through the grouped isend / irecv form,
    from `mine` to `loopback[: rows]` over RCCL's self send / receive, so the point-to-point path (row slices of a larger
    tensor as P2POp buffers, batch_isend_irecv, the wait in finish_exchange) executes at a world size of one
    (tests/test_distributed_rccl.py).

    lo_hi[r] = (lo, hi): the rows of rank r's shard that belong to this chunk (shard-relative).  This rank's own rows
    must already be in place (or enqueued on the current stream).  Returns the list of async work handles; the caller
    waits on them.  Equal chunk heights -> all_gather_into_tensor through `stage` ([world * rows, n_cols]) + scatter
    (enqueued after the wait by `finish_exchange`); otherwise grouped isend / irecv between the slices themselves."""
    world = dist.get_world_size()
    heights = [hi - lo for lo, hi in lo_hi]
    mine = full[starts[rank] + lo_hi[rank][0]: starts[rank] + lo_hi[rank][1]]
    if loopback is not None and heights[rank] > 0:
        ops = [dist.P2POp(dist.isend, mine, rank), dist.P2POp(dist.irecv, loopback[: heights[rank]], rank)]
        return [("p2p", w, None, None) for w in dist.batch_isend_irecv(ops)]
    if world == 1 and not force:
        return []
    if stage is not None and len(set(heights)) == 1:
        if heights[0] == 0:
            return []
        work = dist.all_gather_into_tensor(stage[: world * heights[0]], mine, async_op=True)
        return [("gathered", work, stage, lo_hi)]
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        if heights[rank] > 0:
            ops.append(dist.P2POp(dist.isend, mine, peer))
        if heights[peer] > 0:
            ops.append(dist.P2POp(dist.irecv, full[starts[peer] + lo_hi[peer][0]: starts[peer] + lo_hi[peer][1]], peer))
    return [("p2p", w, None, None) for w in dist.batch_isend_irecv(ops)] if ops else []


def finish_exchange(full, starts, rank, handles):
    """Wait for the handles of `exchange_rows`; scatter staged all-gather blocks into the rank slices."""
    for kind, work, stage, lo_hi in handles:
        work.wait()
        if kind == "gathered":
            rows = lo_hi[0][1] - lo_hi[0][0]
            for r, (lo, hi) in enumerate(lo_hi):
                if r != rank:
                    full[starts[r] + lo: starts[r] + hi].copy_(stage[r * rows:(r + 1) * rows], non_blocking=True)


class _HostStream:
    """Stands in for a torch.cuda.Stream where the pipeline runs on host tensors (the gloo tests of ShardPipeline: world
    sizes 2 and 4 on a box without GPUs): host work is synchronous, so ordering between 'streams' is program order."""
    cuda_stream = 0

    def wait_stream(self, other):
        pass


class _NoContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class ShardPipeline:
    """One rank's side of the sharded extraction on a GPU (torch tensors, device-resident).

    The shard is extracted in `n_chunks` row chunks that alternate between two launch streams, each with its own native
    plan (a plan's device scratch belongs to one stream at a time): the thinning tail of one chunk's kernels overlaps
    the start of the next chunk's, and the exchange of a finished chunk (ordered after that chunk's kernels only) runs
    on the collective's own stream while the next chunk is being extracted.  Only the last chunk's exchange is exposed.
    """

    def __init__(self, specs, n_cols, device_index, dist=None, n_chunks=None, length_hint=None, plan_factory=None,
                 force_exchange=False):
        """plan_factory: callable -> an object with `extract_device(values_ptr, dtype, offsets_ptr, n, out_ptr, ld, stream)` and
        `close()`; with device_index None the pipeline runs on HOST tensors (no streams): how tests/test_distributed_gloo.py
        drives THIS class -- lanes, chunk cuts, the staging ring, the order of finish_exchange -- at world sizes 2 and 4 with
        the emulation build standing in for the kernels.  force_exchange: issue the exchange of every chunk at a world of one
        too (the all-gather of one rank / RCCL's self send-receive): the ring and the waits execute on a one-GPU box."""
        import torch
        self.torch = torch
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.n_cols = int(n_cols)
        self.host = device_index is None
        self.force_exchange = bool(force_exchange)
        self.n_chunks = int(n_chunks) if n_chunks else (8 if dist is not None else 1)
        if plan_factory is None:
            from tsfresh_amd import _native
            specs = list(specs)
            plan_factory = lambda: _native.Plan(specs, device=device_index)   # noqa: E731
        self.plans = [plan_factory()]
        if self.n_chunks > 1:
            self.plans.append(plan_factory())
        if length_hint is not None:
            for p in self.plans:
                p.set_length_hint(*length_hint)
        if self.host:
            self.dev = torch.device("cpu")
            self.main = _HostStream()
            self.lanes = [(self.plans[0], self.main)] if self.n_chunks == 1 else [(p, _HostStream()) for p in self.plans]
        else:
            self.dev = torch.device("cuda", device_index)
            self.main = torch.cuda.current_stream(self.dev)
            self.lanes = [(self.plans[0], self.main)] if self.n_chunks == 1 else \
                [(p, torch.cuda.Stream(device=self.dev)) for p in self.plans]
        self._stage = {}
        self.p2p_only = False         # tests: the grouped isend / irecv form also where the chunk heights agree
        self.exchanges_issued = 0     # handles of the last run (tests)
        self.ring_reuses = 0          # times a staging slot was drained because its block was about to be reused

    def _on(self, stream):
        return _NoContext() if self.host else self.torch.cuda.stream(stream)

    def close(self):
        for p in self.plans:
            p.close()
        self.plans = []

    def _stage_for(self, slot, rows, dtype):
        cur = self._stage.get(slot)
        if cur is None or cur.shape[0] < self.world * int(rows):
            cur = self._stage[slot] = self.torch.empty((self.world * int(rows), self.n_cols), device=self.dev, dtype=dtype)
        return cur

    def run(self, values, offsets, counts, full, dtype_code):
        """values / offsets: this rank's shard (device tensors; offsets int64, relative to `values`);
        counts: rows of every rank's shard; full: [sum(counts), n_cols] float64 device tensor that receives every
        rank's rows (rank-major).  Enqueues everything and waits for the exchange; returns `full`."""
        dist = self.dist
        self.exchanges_issued = self.ring_reuses = 0
        counts = [int(c) for c in counts]
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        n_chunks = max(1, min(self.n_chunks, max(min(counts), 1)))  # the same on every rank
        cuts = [chunk_cuts(c, n_chunks) for c in counts]
        mine = full[int(starts[self.rank]): int(starts[self.rank + 1])]
        starts_l = [int(s) for s in starts]
        n_lanes = len(self.lanes)
        pending = {}  # (lane, ring slot) -> (stream, handles): at most two exchanges in flight per lane
        for pl, st in self.lanes:
            if st is not self.main:
                st.wait_stream(self.main)
        for c in range(n_chunks):
            c0, c1 = cuts[self.rank][c], cuts[self.rank][c + 1]
            lane = c % n_lanes
            pl, st = self.lanes[lane]
            if c1 > c0:
                # offsets stay relative to the start of `values`: a chunk is the same buffer with a later offsets pointer
                pl.extract_device(values.data_ptr(), dtype_code, offsets.data_ptr() + 8 * c0, c1 - c0,
                                  mine.data_ptr() + 8 * self.n_cols * c0, self.n_cols, st.cuda_stream)
            if dist is not None and (self.world > 1 or self.force_exchange):
                slot = (lane, (c // n_lanes) % 2)
                if slot in pending:  # the staging block of this slot is about to be reused: scatter its rows first
                    pst, hs = pending.pop(slot)
                    self.ring_reuses += 1
                    with self._on(pst):
                        finish_exchange(full, starts_l, self.rank, hs)
                lo_hi = [(cuts[r][c], cuts[r][c + 1]) for r in range(self.world)]
                heights = {hi - lo for lo, hi in lo_hi}
                stage = self._stage_for(slot, max(heights), full.dtype) if len(heights) == 1 and not self.p2p_only else None
                loop = None
                if self.world == 1 and self.p2p_only:   # the point-to-point form at a world of one: RCCL's self send / receive
                    loop = self._stage_for(("loop",) + slot, max(heights), full.dtype)
                with self._on(st):  # the collective orders itself after this chunk's kernels only
                    hs = exchange_rows(full, starts_l, self.rank, lo_hi, dist, stage, loopback=loop, force=self.force_exchange)
                    self.exchanges_issued += len(hs)
                    pending[slot] = (st, hs)
        for pst, hs in pending.values():
            with self._on(pst):
                finish_exchange(full, starts_l, self.rank, hs)
        for pl, st in self.lanes:
            if st is not self.main:
                self.main.wait_stream(st)
        return full


def extract_sharded(extract_fn, values, offsets, n_cols, dist=None, torch_device=None, n_chunks=4):
    """Host-array front door of the process-per-GPU form: run `extract_fn(values_shard, offsets_shard) -> ndarray
    [n_local, n_cols]` chunk by chunk on this rank's sum(len^2)-balanced shard and return the full matrix in id order
    (every rank gets it).  The exchange of a chunk overlaps the extraction of the next one; shards of different height
    travel point to point without padding.  With dist=None (single process) this is just extract_fn."""
    import torch
    offsets = np.asarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    if dist is None or dist.get_world_size() == 1:
        return extract_fn(values, offsets)
    world, rank = dist.get_world_size(), dist.get_rank()
    bounds = shard_bounds(np.diff(offsets), world)
    counts = np.diff(bounds)
    starts = [int(b) for b in bounds]
    dev = torch_device if torch_device is not None else "cpu"
    full = torch.empty((n, n_cols), dtype=torch.float64, device=dev)
    n_chunks = max(1, min(int(n_chunks), int(counts.max()) if len(counts) else 1))
    cuts = [chunk_cuts(c, n_chunks) for c in counts]
    handles = []
    for c in range(n_chunks):
        lo, hi = int(bounds[rank]) + cuts[rank][c], int(bounds[rank]) + cuts[rank][c + 1]
        if hi > lo:
            sub_off = offsets[lo:hi + 1]
            block = extract_fn(values[sub_off[0]:sub_off[-1]], sub_off - sub_off[0])
            full[lo:hi] = torch.as_tensor(np.ascontiguousarray(block), dtype=torch.float64).reshape(-1, n_cols).to(dev)
        lo_hi = [(cuts[r][c], cuts[r][c + 1]) for r in range(world)]
        handles.append(exchange_rows(full, starts, rank, lo_hi, dist, None))
    for hs in handles:
        finish_exchange(full, starts, rank, hs)
    return full.cpu().numpy()


# Plans of extract_on_devices, kept across calls: (shard slot, device, specs) -> _PlanEntry.  The workers are fresh
# threads on every call, so the per-thread cache of feature_extraction/extraction.py never hits for them: round 2 built
# len(devices) plans per call (8 ms each) and never released them -- device memory grew with every call.  A slot's
# plan is driven by one thread at a time (the entry's lock, under which it is also CREATED: the global lock is never held
# across the ~8 ms of a plan build, so the devices build theirs concurrently).  An entry counts its users; the LRU only
# evicts -- and closes -- entries nobody holds, so a worker can never be handed a plan that is closed before it runs.
class _PlanEntry:
    __slots__ = ("plan", "lock", "users")

    def __init__(self):
        self.plan, self.lock, self.users = None, threading.Lock(), 0


_DEVICE_PLANS = collections.OrderedDict()
_DEVICE_PLANS_LOCK = threading.Lock()
_DEVICE_PLANS_MAX = 32


def _device_plan_acquire(slot, device, specs):
    """-> the entry of (slot, device, specs) with its user count raised; pair with _device_plan_release."""
    key = (int(slot), int(device), tuple((cid, tuple(float(v) for v in p)) for cid, p in specs))
    evicted = []
    with _DEVICE_PLANS_LOCK:
        entry = _DEVICE_PLANS.get(key)
        if entry is None:
            entry = _DEVICE_PLANS[key] = _PlanEntry()
        else:
            _DEVICE_PLANS.move_to_end(key)
        entry.users += 1
        if len(_DEVICE_PLANS) > _DEVICE_PLANS_MAX:
            for old_key in [k for k, e in _DEVICE_PLANS.items() if e.users == 0]:
                if len(_DEVICE_PLANS) <= _DEVICE_PLANS_MAX:
                    break
                evicted.append(_DEVICE_PLANS.pop(old_key))   # unreachable from here on, and nobody is inside it
    for old in evicted:
        with old.lock:
            if old.plan is not None:
                old.plan.close()
                old.plan = None
    return entry


def _device_plan_release(entry):
    with _DEVICE_PLANS_LOCK:
        entry.users -= 1


def clear_device_plans():
    """Release the idle plans extract_on_devices keeps (and the device memory they hold)."""
    with _DEVICE_PLANS_LOCK:
        keys = [k for k, e in _DEVICE_PLANS.items() if e.users == 0]
        entries = [_DEVICE_PLANS.pop(k) for k in keys]
    for e in entries:
        with e.lock:
            if e.plan is not None:
                e.plan.close()
                e.plan = None


def extract_on_devices(specs, values, offsets, devices, times=None, n_cols=None, parts=None):
    """One process, several GPUs: the ragged batch is cut into sum(len^2)-balanced contiguous shards, one per device;
    every device runs its chunked host pipeline (tsfa_extract, TSFA_HOST) from its own host thread and writes its rows
    straight into its slice of ONE page-locked matrix.  -> float64 [n_series, n_cols] in series order.
    parts: [(sub-list of specs, their column indices)] for a settings object that needs several native plans
    (feature_extraction/extraction.py: _split_native_specs); a device then uploads its shard once and runs every part on it."""
    from tsfresh_amd import _native
    specs = list(specs)
    n_cols = len(specs) if n_cols is None else int(n_cols)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("devices must name at least one HIP device")
    out = _native._result_matrix(n, n_cols)
    if n == 0 or n_cols == 0:
        return out
    bounds = shard_bounds(np.diff(offsets), len(devices))
    errors = []

    def work(k, dev):
        lo, hi = int(bounds[k]), int(bounds[k + 1])
        if hi <= lo:
            return
        entry = None
        entries = []
        try:
            sub = offsets[lo:hi + 1]
            if parts:
                import contextlib
                entries = [(_device_plan_acquire(k, dev, sp), cols) for sp, cols in parts]
                with contextlib.ExitStack() as stack:
                    plans = []
                    for (e, cols), (sp, _) in zip(entries, parts):
                        stack.enter_context(e.lock)
                        if e.plan is None:
                            e.plan = _native.Plan(sp, device=dev)
                        plans.append((e.plan, cols))
                    dm = _native.DeviceMatrix(hi - lo, n_cols, dev)
                    try:
                        _native.extract_parts_into(plans, values[sub[0]:sub[-1]], sub - sub[0], dm,
                                                   times=None if times is None else times[sub[0]:sub[-1]])
                        out[lo:hi] = dm.to_host()
                    finally:
                        dm.free()
                return
            entry = _device_plan_acquire(k, dev, specs)
            with entry.lock:
                if entry.plan is None:
                    entry.plan = _native.Plan(specs, device=dev)
                entry.plan.extract_host(values[sub[0]:sub[-1]], sub - sub[0],
                                        times=None if times is None else times[sub[0]:sub[-1]], out=out[lo:hi])
        except BaseException as e:  # surfaced in the calling thread
            errors.append(e)
        finally:
            if entry is not None:
                _device_plan_release(entry)
            for e, _ in entries:
                _device_plan_release(e)

    threads = [threading.Thread(target=work, args=(k, d), name="tsfresh_amd-dev%d" % d) for k, d in enumerate(devices)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out
