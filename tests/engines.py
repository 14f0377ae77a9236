"""Three ways to evaluate an FCParameters mapping on a ragged batch, with one signature:
    engine(fc_parameters, values, offsets) -> (column names, float64 matrix)

  oracle : oracle/ (numpy restatement of the reference)                          CPU
  emul   : tests/emul (g++ single-thread build of the kernel sources)            CPU, logic only
  hip    : tsfresh_amd._native.Plan -> libtsfresh_amd.so -> HIP kernels          GPU (the product)
"""
import numpy as np


def oracle_engine(fc_parameters, values, offsets, kind="value", times=None):
    from oracle.extract import oracle_matrix
    return oracle_matrix(np.asarray(values, dtype=np.float64), offsets, fc_parameters, kind=kind, times=times)


_PARALLEL_CACHE = {}


def _oracle_one(job):
    fc_parameters, x, kind = job
    return oracle_engine(fc_parameters, x, np.array([0, len(x)], dtype=np.int64), kind=kind)


def oracle_engine_parallel(fc_parameters, values, offsets, kind="value", times=None, workers=None):
    """oracle_engine with the series dealt over worker processes (longest first): the reference's own arithmetic is
    O(n^2) in memory and time for the entropies, so a handful of 4096 .. 8192-sample series is minutes on one core."""
    import hashlib
    import multiprocessing as mp
    import os
    assert times is None
    values = np.asarray(values, dtype=np.float64)
    n = len(offsets) - 1
    # the fixture pairs `long` and `long_nosimd` hold the same series: the oracle runs once per session for both
    key = (hashlib.sha1(values.tobytes() + np.asarray(offsets, dtype=np.int64).tobytes()).hexdigest(), kind,
           repr(sorted((k, repr(v)) for k, v in dict(fc_parameters).items())))
    if key in _PARALLEL_CACHE:
        names, out = _PARALLEL_CACHE[key]
        return list(names), out.copy()
    order = sorted(range(n), key=lambda i: -(offsets[i + 1] - offsets[i]))
    jobs = [(fc_parameters, values[offsets[i]:offsets[i + 1]], kind) for i in order]
    workers = workers or min(os.cpu_count() or 1, 8, max(n, 1))
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_oracle_one, jobs, chunksize=1)
    names = res[0][0]
    out = np.empty((n, len(names)))
    for i, (nm, row) in zip(order, res):
        assert list(nm) == list(names)
        out[i] = row[0]
    _PARALLEL_CACHE[key] = (list(names), out.copy())
    return names, out


def emul_engine(fc_parameters, values, offsets, kind="value", times=None):
    from emul_lib import emul_extract
    return emul_extract(fc_parameters, values, offsets, kind=kind, times=times)


def hip_engine(fc_parameters, values, offsets, kind="value", device=0, times=None, options=None):
    """times: float64 hours since each series' first timestamp (DatetimeIndex data) -> linear_trend_timewise
    options: {name: value} for tsfa_plan_set_option (an alternative route to the same numbers: the A/B tests)"""
    from tsfresh_amd import _native
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    fplan = compile_fc_parameters(fc_parameters, has_datetime_index=times is not None)
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=device)
    for name, value in (options or {}).items():
        plan.set_option(name, value)
    try:
        out = plan.extract_host(values, np.asarray(offsets, dtype=np.int64), times=times)
    finally:
        plan.close()
    return [kind + "__" + n for n in fplan.names], out
