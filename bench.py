#!/usr/bin/env python
"""Headline benchmark: series/sec of the feature-extraction hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (tsfa_extract: every kernel of the plan) over one batch of synthetic series
that is ALREADY RESIDENT IN HBM, writing the dense [n_series x n_cols] float64 feature matrix to HBM; with N > 1
each rank (one process per GPU) extracts its own id-shard in row chunks and the shards are reassembled on every rank
with RCCL all-gathers (north_star) -- one per chunk, overlapped with the extraction of the next chunk -- inside the
timed region.  Per-GPU work is fixed -> "scaling": "weak".

Workload at N = 1: BASELINE.json configs[2] -- 100k synthetic float32 series x len 1024,
ComprehensiveFCParameters (783 columns) -- the configuration the metric/target is quoted on.

The JSON line also carries
  roofline     : dominant kernel's algorithmic HBM bytes per launch / its HIP-event duration vs 8 TB/s
  cpu_baseline : the numpy oracle (a port of the reference's calculators) timed on a bounded sample of the same
                 workload on the host cores (warm pool, >= 32 series per worker; one run, --cpu-baseline-full: median of 3) -- a reported baseline,
                 not the target; `reference_estimate` scales it by the port/reference ratio measured in the build
                 container (profiles/r02_reference_cpu.json)
  parity_sample: 8 rows of the timed output against the oracle (tests/parity.py), outside the timed region
  e2e          : host-buffer and DataFrame -> DataFrame rates of the same plan (PCIe-inclusive; never `value`)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _cpu_baseline_worker(args):
    values, offsets, params_name = args
    import warnings
    from oracle.extract import oracle_matrix
    from tsfresh_amd.feature_extraction import settings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return oracle_matrix(values, offsets, getattr(settings, params_name)())


def _cpu_warm(_):
    sys.path.insert(0, ROOT)
    import oracle.extract  # noqa: F401
    import tsfresh_amd.feature_extraction.settings  # noqa: F401
    return 0


def lib_sha16():
    """Build id of the library this process loaded: the first 16 hex digits of sha256(libtsfresh_amd.so).  The counter
    scripts (profiles/pmc_hbm.sh, pmc_issue.sh) record the same figure next to what they measure."""
    import hashlib
    from tsfresh_amd import _native
    try:
        return hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _replayed_counters(fname, cfg):
    """-> (document, stale): a committed PMC document is replayed only when it was taken on THIS workload and on THIS
    build of the library; a document of another build is refused and reported as stale."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", fname)))
    except (OSError, ValueError):
        return None, False
    w = doc.get("workload", {})
    if any(w.get(k) != cfg.get(k) for k in ("n_series_per_gpu", "length", "n_cols")):
        return None, False
    if doc.get("lib_sha16") != lib_sha16():
        return None, True
    return doc, False


def measured_hbm_traffic(kernel, cfg):
    """-> (HBM bytes per launch of `kernel`, stale) from the committed PMC pass (profiles/pmc_hbm.sh ->
    profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same command, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when the file is absent, was taken on another workload
    or on another build of the library."""
    doc, stale = _replayed_counters("hbm_traffic.json", cfg)
    return (doc.get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch") if doc else None), stale


def measured_valu_issue(cfg):
    """-> (vector-instruction issue of the whole step, stale) from the committed PMC pass (profiles/pmc_issue.sh ->
    profiles/valu_issue.json: SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE per kernel, separate rocprofv3 --pmc
    runs of this command).  This -- not HBM bandwidth -- is the roof ComprehensiveFCParameters runs against (DESIGN.md
    section 3.1): `ms_at_full_issue` = sum over kernels of wave-instructions x the measured cycles per instruction of the
    kernel's mix / (1024 SIMDs x the measured shader clock)."""
    doc, stale = _replayed_counters("valu_issue.json", cfg)
    return (doc.get("step") if doc else None), stale


def physical_cores():
    """(physical cores, logical CPUs) of this box."""
    logical = os.cpu_count() or 1
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen), logical
    except OSError:
        pass
    return logical, logical


def cpu_baseline(pool, workers, length, params_name, seed, per_worker=32, repeats=3):
    """The CPU path beside the GPU number, on the host cores of this box (rank 0, N = 1 only).

    What is timed is oracle/ -- the numpy restatement of the reference's calculators ("port", all 75 calculators) --
    because /root/reference does not exist on the GPU box.  Protocol: one worker process per core,
    started and warmed OUTSIDE the clock, `per_worker` (>= 32) series each, `repeats` runs, median; BLAS threads = 1
    (docs/text/tsfresh_on_a_cluster.rst:216-231).  profiles/r02_reference_cpu.json holds the same protocol run in the
    build container on BOTH the real reference (tsfresh.extract_features + MultiprocessingDistributor,
    distribution.py:438; 70 of 75 calculators importable there) and this port: the port does 0.945x the reference's
    series/s/core, so `value / port_over_reference_per_core` estimates the reference on these cores."""
    import statistics
    walls = []
    for r in range(repeats):
        rng = np.random.default_rng(seed + r)
        jobs = []
        for _ in range(workers):
            x = rng.standard_normal((per_worker, length), dtype=np.float32).astype(np.float64)
            jobs.append((x.reshape(-1), np.arange(per_worker + 1, dtype=np.int64) * length, params_name))
        t0 = time.perf_counter()
        pool.map(_cpu_baseline_worker, jobs)
        walls.append(time.perf_counter() - t0)
    n = workers * per_worker
    wall = statistics.median(walls)
    phys, logical = physical_cores()
    doc = {"value": n / wall, "unit": "series/sec", "cores": workers, "kind": "port",
           "physical_cores": phys, "logical_cpus": logical,
           "series_per_sec_per_core": n / wall / workers,
           "sample": "%d series x len %d, %s, oracle/ (numpy port of the reference calculators, all 75), %d warm worker "
                     "processes x %d series, median of %d runs (%s s), pool start outside the clock" % (
                         n, length, params_name, workers, per_worker, repeats, ", ".join("%.1f" % w for w in walls))}
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json")))
        ratio = ref["port_over_reference_per_core"]
        doc["port_over_reference_per_core"] = ratio
        doc["reference_estimate"] = {
            "value": doc["value"] / ratio, "unit": "series/sec",
            "basis": "real tsfresh extract_features(n_jobs=cores) vs this port, same protocol, build container "
                     "(profiles/r02_reference_cpu.json: reference %.3f, port %.3f series/s/core; reference default "
                     "n_jobs = cores // 2 -> half this value)" % (
                         ref["reference"]["all_cores"]["series_per_sec_per_core"],
                         ref["port"]["all_cores"]["series_per_sec_per_core"])}
    except (OSError, ValueError, KeyError):
        pass
    return doc


def parity_sample(pool, rows_in, rows_out, names, params_name):
    """A few rows of the TIMED output against the oracle (outside the timed region): 'ok' or 'fail: ...'."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity import compare
    L = rows_in.shape[1]
    jobs = [(rows_in[i].astype(np.float64), np.array([0, L], dtype=np.int64), params_name) for i in range(len(rows_in))]
    res = pool.map(_cpu_baseline_worker, jobs) if pool is not None else [_cpu_baseline_worker(j) for j in jobs]
    onames = res[0][0]
    want = np.concatenate([r[1] for r in res], axis=0)
    kind_names = ["value__" + n for n in names]
    idx = [kind_names.index(n) for n in onames]
    bad = compare(onames, rows_out[:, idx], want, [r.astype(np.float64) for r in rows_in])
    return "ok" if not bad else "fail: %d of %d cells, first %s" % (len(bad), want.size, bad[:3])


def e2e_block(plan, fplan, params_cls, n, L, calls=4):
    """Boundary timings beside the HBM-resident `value`: the same plan on HOST buffers (PCIe-inclusive:
    chunked H2D / kernels / D2H pipeline of tsfa_extract(TSFA_HOST)) and DataFrame in -> DataFrame out through
    extract_features (packer + host path + frame construction).  Best of `calls - 1` after one warm call."""
    import warnings

    import pandas as pd
    from tsfresh_amd import extract_features
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, L), dtype=np.float32)
    offsets = np.arange(n + 1, dtype=np.int64) * L
    plan.set_length_hint(0, 0)
    res = {"n_series": n, "length": L}
    best = None
    for _ in range(calls):
        t0 = time.perf_counter()
        m = plan.extract_host(x.reshape(-1), offsets)
        dt = time.perf_counter() - t0
        best = dt if best is None or _ == 1 else min(best, dt)  # first call: allocations / page-locking
    res["host_buffers"] = {"seconds": best, "series_per_sec": n / best,
                           "bytes_h2d": int(x.nbytes), "bytes_d2h": int(m.nbytes)}
    df = pd.DataFrame({"id": np.repeat(np.arange(n), L), "time": np.tile(np.arange(L), n), "value": x.reshape(-1)})
    best = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(calls):
            t0 = time.perf_counter()
            f = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params_cls())
            dt = time.perf_counter() - t0
            best = dt if best is None or _ == 1 else min(best, dt)
    assert f.shape == (n, len(fplan))
    same = np.array_equal(np.nan_to_num(f.to_numpy()), np.nan_to_num(m))
    res["dataframe"] = {"seconds": best, "series_per_sec": n / best, "rows_in": int(n * L),
                        "equals_host_buffer_result": bool(same)}
    return res


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_command(n_ranks, argv, port):
    """argv and environment that run this file as `n_ranks` processes, one per GPU, on this node -- what the driver's
    launcher form does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py ...`).  The reference's counterpart is the worker pool of its MultiprocessingDistributor
    (tsfresh/utilities/distribution.py:438-494)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
           "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's P2P buffers across processes need it here
    env.setdefault("OMP_NUM_THREADS", "1")
    env["TSFA_BENCH_LAUNCHED"] = "1"
    return cmd, env


def strong_ragged_layout(total, lo, hi, world, n_chunks=8):
    """configs[4] as a strong-scaling job (`--strong --ragged LO:HI --n-series TOTAL --gpus N`): ONE list of series lengths for
    the whole job (seed 42, the same on every rank), cut into contiguous shards of ~equal sum(len^2)
    (tsfresh_amd.distributed.shard_bounds) -- so the shards differ in HEIGHT and their row chunks travel point to point.
    Pure numpy: `--plan-only` prints it without a GPU."""
    from tsfresh_amd.distributed import chunk_cuts, shard_bounds
    lens = np.random.default_rng(42).integers(lo, hi + 1, size=total, dtype=np.int64)
    bounds = shard_bounds(lens, world)
    counts = [int(c) for c in np.diff(bounds)]
    chunks = max(1, min(n_chunks, max(min(counts), 1)))
    equal = all(len({chunk_cuts(c, chunks)[k + 1] - chunk_cuts(c, chunks)[k] for c in counts}) == 1 for k in range(chunks))
    cost = [float((lens[bounds[r]:bounds[r + 1]].astype(np.float64) ** 2).sum()) for r in range(world)]
    return {"lens": lens, "bounds": bounds, "counts": counts, "row_chunks": chunks,
            "exchange": "all_gather" if equal else "p2p", "sum_len2": cost}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-series", type=int, default=100_000, help="series per GPU")
    ap.add_argument("--length", type=int, default=1024)
    ap.add_argument("--params", default="comprehensive", choices=["comprehensive", "efficient", "minimal"])
    ap.add_argument("--ragged", default="", help="LO:HI -> series lengths uniform on [LO, HI] (configs[4] shape); "
                                                 "--length is ignored")
    ap.add_argument("--walk", action="store_true", help="random walks (randn().cumsum(), the recipe of the reference's "
                    "tests/benchmark.py:22) instead of i.i.d. N(0,1): the data-dependent kernels (LZ parse, CWT peaks, pair "
                    "sweep) see long runs and smooth ridges")
    ap.add_argument("--offset", type=float, default=0.0, help="add OFFSET to every sample (|mean| >> spread: the "
                    "double-double second passes of the Langevin fit and of AR / ADF take every series)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --n-series is the TOTAL of the job, split evenly over the N ranks (configs[3]: "
                         "`--strong --n-series 1000000 --length 256 --gpus N`); the default is weak (series per GPU fixed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="three runs of the CPU protocol (median) and the reference's default n_jobs = cpu_count() // 2 as a "
                         "second leg (~4 min of wall clock) instead of one run (~1 min): profiles/r04_z_bench.json was taken so")
    ap.add_argument("--cpu-workers", type=int, default=0, help="worker processes of the CPU baseline (default: every physical core)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer / DataFrame boundary timings")
    ap.add_argument("--plan-option", action="append", default=[], metavar="NAME=VALUE",
                    help="tsfa_plan_set_option on every plan of the run (A/B of an alternative route to the same numbers, e.g. "
                         "seq_rows=0); recorded in config.plan_options.  The default line sets none")
    ap.add_argument("--plan-only", action="store_true",
                    help="print the shard layout of the job as JSON (rows and sum(len^2) per rank, exchange form) and exit: no GPU")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N > 1: row chunks per step; the all-gather of chunk c runs on RCCL's stream while chunk c + 1 is "
                         "being extracted (0 = 8 when N > 1, else 1)")
    args = ap.parse_args()

    if args.plan_only:
        if args.strong and args.ragged:
            lo, hi = (int(t) for t in args.ragged.split(":"))
            lay = strong_ragged_layout(args.n_series, lo, hi, args.gpus, args.chunks if args.chunks > 0 else 8)
            doc = {"gpus": args.gpus, "scaling": "strong", "shard_rows": lay["counts"], "sum_len2": lay["sum_len2"],
                   "exchange": lay["exchange"], "row_chunks": lay["row_chunks"]}
        else:
            per = args.n_series // args.gpus if args.strong else args.n_series
            doc = {"gpus": args.gpus, "scaling": "strong" if args.strong else "weak", "shard_rows": [per] * args.gpus,
                   "exchange": "all_gather", "row_chunks": args.chunks if args.chunks > 0 else 8}
        print(json.dumps(doc))
        return
    # `python bench.py --gpus N` (no launcher): start N ranks of this file and hand back their exit code; rank 0 of the
    # child job prints the one JSON line.  Under a launcher (WORLD_SIZE set) this is skipped.
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("TSFA_BENCH_SELF_LAUNCH")):
        import subprocess
        cmd, env = self_launch_command(args.gpus, sys.argv[1:], free_port())
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    from tsfresh_amd import _native
    from tsfresh_amd.feature_extraction import settings
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or _native.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the native path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("TSFA_BENCH_FORCE_DIST"):  # the env var exercises the N > 1 code path on one GPU
        import torch.distributed as dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist_mod.init_process_group(backend="nccl", device_id=dev)
        dist = dist_mod
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (use --nproc-per-node %d, or run "
                         "`python bench.py --gpus %d` without a launcher)" % (args.gpus, world, args.gpus, args.gpus))
    ranks_seen = None
    if dist is not None:
        # "did RCCL see N ranks": a sum of ones over the communicator, next to torch's own world size
        ones = torch.ones(1, device=dev, dtype=torch.int32)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)

    cls = {"comprehensive": settings.ComprehensiveFCParameters, "efficient": settings.EfficientFCParameters,
           "minimal": settings.MinimalFCParameters}[args.params]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fplan = compile_fc_parameters(cls())
    n_cols = len(fplan)

    # BASELINE.md 3.4's recipe: np.random.default_rng(seed).standard_normal((n, L), dtype=float32), seed = 42 (+ rank:
    # every rank its own shard), drawn on the host and copied to the device BEFORE the clock
    n, L = args.n_series, args.length
    layout = None
    if args.strong and args.ragged:
        lo, hi = (int(t) for t in args.ragged.split(":"))
        layout = strong_ragged_layout(args.n_series, lo, hi, world, args.chunks if args.chunks > 0 else 8)
        n = layout["counts"][rank]
    elif args.strong:
        if args.n_series % world:
            raise SystemExit("bench.py --strong: --n-series %d is not a multiple of the %d ranks" % (args.n_series, world))
        n = args.n_series // world
    rng = np.random.default_rng(42 + rank)
    if args.ragged:
        lo, hi = (int(t) for t in args.ragged.split(":"))
        lens = rng.integers(lo, hi + 1, size=n, dtype=np.int64)
        if layout is not None:   # this rank's slice of the job's ONE length list
            lens = layout["lens"][layout["bounds"][rank]:layout["bounds"][rank + 1]]
        h_offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=h_offsets[1:])
        total = int(h_offsets[-1])
        L = total // n  # mean length, for the byte accounting below
        h_values = rng.standard_normal(total, dtype=np.float32)
    else:
        h_values = rng.standard_normal((n, L), dtype=np.float32)  # i.i.d. N(0,1) float32 series
        h_offsets = np.arange(n + 1, dtype=np.int64) * L
        if args.walk:   # tests/benchmark.py:22 of the reference: randn().cumsum()
            h_values = np.cumsum(h_values.astype(np.float64), axis=1).astype(np.float32)
        h_values = h_values.reshape(-1)
    if args.offset:
        h_values = (h_values.astype(np.float64) + args.offset).astype(np.float32)
    values = torch.from_numpy(np.ascontiguousarray(h_values)).to(dev)
    offsets = torch.from_numpy(h_offsets).to(dev)
    del h_values
    # The product pipeline (tsfresh_amd/distributed.py: ShardPipeline): the shard is extracted in row chunks on two
    # alternating launch streams (a plan each); with N > 1 every finished chunk is exchanged (RCCL all-gather into a
    # staging block + device scatter into the rank-major matrix) while the next chunk is being extracted, so only the
    # last chunk's exchange is exposed.  Every rank ends up with all rows: full = [world x n, n_cols], rank-major.
    from tsfresh_amd.distributed import ShardPipeline
    n_chunks = args.chunks if args.chunks > 0 else (8 if dist is not None else 1)
    n_chunks = max(1, min(n_chunks, n))
    pipe = ShardPipeline(fplan.native_specs(_native.calc_id), n_cols, local_rank, dist=dist, n_chunks=n_chunks,
                         length_hint=None if args.ragged else (L, L))  # equal lengths, known up front: no length scan
    plan_options = {kv.split("=", 1)[0]: float(kv.split("=", 1)[1]) for kv in args.plan_option}
    for pl in pipe.plans:
        for name, value in plan_options.items():
            pl.set_option(name, value)
    plan = pipe.plans[0]
    counts = layout["counts"] if layout is not None else [n] * world
    row0 = [int(v) for v in np.concatenate([[0], np.cumsum(counts)])]
    full = torch.empty((row0[-1], n_cols), device=dev, dtype=torch.float64)
    out = full[row0[rank]:row0[rank + 1]]
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        pipe.run(values, offsets, counts, full, _native.TSFA_F32)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)  # inputs fully materialised in HBM before the first launch
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
    value = row0[-1] * args.steps / elapsed
    if dist is not None and world > 1:
        # every rank holds every rank's rows: a checksum of each rank's block (NaNs zeroed), all-reduced with MAX and
        # MIN, must agree on all ranks -- i.e. everybody received the same bytes for every block
        sums = torch.stack([torch.nan_to_num(full[row0[r]:row0[r + 1]]).sum() for r in range(world)])
        hi, lo = sums.clone(), sums.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        assert bool((hi == lo).all().item()), "exchange: ranks disagree on the gathered matrix"

    # ---- N > 1: what the exchange costs on top of the extraction (outside the timed region) ----
    multi = None
    if dist is not None and world > 1:
        try:   # diagnostics only: a failure here (the same on every rank) must not cost the bench line
            local = ShardPipeline(fplan.native_specs(_native.calc_id), n_cols, local_rank, dist=None, n_chunks=n_chunks,
                                  length_hint=None if args.ragged else (L, L))
            local.run(values, offsets, [n], out, _native.TSFA_F32)   # a world of one: this rank's rows into its own block
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                local.run(values, offsets, [n], out, _native.TSFA_F32)
            barrier()
            t_local = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(t_local, op=dist.ReduceOp.MAX)
            compute_ms = 1000.0 * float(t_local.item()) / max(args.steps, 1)
            local.close()
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                rccl = None
            gathered = (row0[-1] - n) * n_cols * 8
            multi = {"world": dist.get_world_size(), "ranks_seen_by_rccl": ranks_seen,
                     "backend": "nccl (RCCL %s)" % rccl, "row_chunks": n_chunks, "shard_rows": counts,
                     "exchange": layout["exchange"] if layout is not None else "all_gather",
                     "compute_only_ms_per_step": compute_ms, "exchange_ms_exposed": ms_per_step - compute_ms,
                     "bytes_received_per_rank_per_step": gathered,
                     "exchange_gbs_per_rank_if_fully_exposed": gathered / max((ms_per_step - compute_ms) * 1e-3, 1e-9) / 1e9}
        except Exception as e:  # noqa: BLE001
            multi = {"world": dist.get_world_size(), "ranks_seen_by_rccl": ranks_seen, "error": repr(e)}

    # ---- per-kernel HIP-event timings (events recorded on the launch stream), 2 profiled passes ----
    plan.set_profiling(True)
    kt = {}
    reps = 2
    for _ in range(reps):
        plan.extract_device(values.data_ptr(), _native.TSFA_F32, offsets.data_ptr(), n, out.data_ptr(), n_cols, stream)
        for name, ms in plan.last_timings():
            kt[name] = kt.get(name, 0.0) + ms / reps
    plan.set_profiling(False)
    # every column of the TIMED output: finite everywhere, except the columns that are NaN by definition
    nonfinite_cols = [fplan.names[j] for j in torch.nonzero(~torch.isfinite(out).all(dim=0)).flatten().tolist()]
    finite = all(nm.startswith("query_similarity_count") for nm in nonfinite_cols)

    if rank == 0:
        dom = max(kt, key=kt.get) if kt else None
        roof = None
        if dom:
            fam_cols = {"k_basic": 0, "k_sort": 0, "k_spectral": 0, "k_ar": 0, "k_entropy": 0, "k_cwtpeaks": 0,
                        "k_seq": 0, "k_trend": 0, "k_cwt_gemm": 0}
            from tsfresh_amd.feature_extraction.registry import CALCULATORS  # noqa: F401
            fam_of = {"sample_entropy": "k_entropy", "approximate_entropy": "k_entropy",
                      "cwt_coefficients": "k_cwt_gemm", "number_cwt_peaks": "k_cwtpeaks",
                      "lempel_ziv_complexity": "k_seq"}
            for nm in fplan.names:
                fam_cols[fam_of.get(nm.split("__")[0], "other")] = fam_cols.get(fam_of.get(nm.split("__")[0], "other"), 0) + 1
            cols_dom = fam_cols.get(dom, 0) or n_cols
            # algorithmic bytes per series = 4*L read (each sample once) + 8 bytes per column written
            alg_bytes = n * (4 * L + 8 * cols_dom)
            achieved = alg_bytes / (kt[dom] * 1e-3) / 1e9
            # the entropy family runs as k_entropy_bits for series up to 1024 samples (fam_entropy_bits.h), as k_entropy beyond
            kname = dom
            if dom == "k_entropy" and L <= 1024 and not args.ragged:
                kname = "k_entropy_bits"
            notes = {"k_entropy_bits": "sorted ranges + bit-matrix sweep of all template pairs: VALU / LDS-issue bound, not "
                                       "HBM-bound (DESIGN.md roofline section)",
                     "k_entropy": "O(L^2) template-pair sweep: VALU(fp64)-bound, not HBM-bound (DESIGN.md roofline section)"}
            wl = {"n_series_per_gpu": n, "length": L, "n_cols": n_cols}
            traffic, stale_t = measured_hbm_traffic(kname, wl)
            valu, stale_v = measured_valu_issue(wl)
            roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "traffic_source": "profiles/hbm_traffic.json (builder-measured: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes of this command, profiles/pmc_hbm.sh; replayed, not re-measured in this run; "
                                      "refused when its lib_sha16 is not the loaded library's)",
                    "lib_sha16": lib_sha16(),
                    "stale": bool(stale_t or stale_v),
                    "kernel_ms": kt[dom],
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "note": notes.get(kname, "compute-side bound: see DESIGN.md roofline section"),
                    "valu": valu}
        line = {
            "metric": "series/sec (ComprehensiveFCParameters, len=1024)" if args.params == "comprehensive" and L == 1024
                      else "series/sec (%s, %s)" % (args.params, ("ragged len %s" % args.ragged) if args.ragged else "len=%d" % L),
            "value": value, "unit": "series/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d series/GPU x len %d float32 %s%s, %sFCParameters (%d columns), "
                                   "inputs and outputs resident in HBM%s" % (
                                       n, L, "random walks (cumsum of N(0,1))" if args.walk else "i.i.d. N(0,1)",
                                       (" + %g" % args.offset) if args.offset else "", args.params.capitalize(), n_cols,
                                       ", id-sharded + RCCL all-gather of the feature matrix" if world > 1 else ""),
                       "n_series_per_gpu": n, "length": L, "n_cols": n_cols, "parallelism": "ids%d" % world,
                       "row_chunks_per_step": n_chunks},
            "kernel_ms": kt, "outputs_finite": finite, "roofline": roof,
        }
        if plan_options:
            line["config"]["plan_options"] = plan_options
        if multi is None and dist is not None:   # a world of one through the distributed code path
            multi = {"world": dist.get_world_size(), "ranks_seen_by_rccl": ranks_seen, "row_chunks": n_chunks}
        if multi is not None:
            multi["self_launched"] = bool(os.environ.get("TSFA_BENCH_LAUNCHED"))
            line["multi_gpu"] = multi
        line["nonfinite_columns"] = nonfinite_cols
        params_name = args.params.capitalize() + "FCParameters"
        pool = None
        if world == 1 and not args.no_cpu_baseline:
            import multiprocessing as mp
            for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
                os.environ[v] = "1"  # the reference's own advice (docs/text/tsfresh_on_a_cluster.rst:216-231)
            # one worker per PHYSICAL core (SMT siblings halve the per-core rate), at most 64: measured on the 128-core box of
            # round 5 (profiles/r05_z_bench.json vs r05_zz), 128 workers give the CPU a LOWER total -- 31.3 series/s in 131 s
            # against 37.4 in 55 s with 64 (memory bandwidth: the numpy port streams n x n / 2 intermediate arrays) -- so 64
            # is the CPU's best showing here, and it keeps the default run within minutes; --cpu-workers N overrides
            workers = args.cpu_workers if args.cpu_workers > 0 else min(physical_cores()[0], 64)
            pool = mp.get_context("spawn").Pool(workers)
            pool.map(_cpu_warm, range(4 * workers))
        if world == 1 and not args.ragged:
            # ~8 rows of the timed output against the oracle, outside the timed region
            pick = sorted(set([0, 1, 2, n // 3, n // 2, n - 3, n - 2, n - 1]))
            if L > 4096:   # the oracle's entropies hold n x n float64 matrices (the reference's own arithmetic)
                pick = [0, n - 1] if L <= 8192 else []
            pick = [i for i in pick if 0 <= i < n]
            rows_in = values.view(n, L)[pick].cpu().numpy()
            rows_out = out[pick].cpu().numpy()
            line["parity_sample"] = parity_sample(pool, rows_in, rows_out, fplan.names, params_name) if pick else "skipped (series too long for the oracle)"
        if world == 1 and not args.ragged and not args.no_e2e:
            line["e2e"] = e2e_block(plan, fplan, cls, min(n, 20_000), L)
            if n > 20_000 and n * L <= 110_000_000:
                # the headline shape itself: configs[2] as a long frame is 102 M rows
                line["e2e_headline"] = e2e_block(plan, fplan, cls, n, L, calls=3)
        if pool is not None:
            line["cpu_baseline"] = cpu_baseline(pool, workers, L, params_name, seed=42, repeats=3 if args.cpu_baseline_full else 1)
            pool.close()
            pool.join()
            # the reference's DEFAULT: n_jobs = cpu_count() // 2 worker processes (tsfresh/defaults.py:7), one run
            half = max(1, (os.cpu_count() or 2) // 2)
            if half != workers and args.cpu_baseline_full:
                pool2 = mp.get_context("spawn").Pool(half)
                pool2.map(_cpu_warm, range(4 * half))
                d2 = cpu_baseline(pool2, half, L, params_name, seed=43, per_worker=8, repeats=1)
                pool2.close()
                pool2.join()
                line["cpu_baseline"]["default_n_jobs"] = {"workers": half, "value": d2["value"], "unit": "series/sec",
                                                          "series_per_sec_per_core": d2["series_per_sec_per_core"],
                                                          "sample": d2["sample"]}
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    pipe.close()


if __name__ == "__main__":
    main()
