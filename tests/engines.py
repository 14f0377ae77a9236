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


def emul_engine(fc_parameters, values, offsets, kind="value", times=None):
    from emul_lib import emul_extract
    return emul_extract(fc_parameters, values, offsets, kind=kind, times=times)


def hip_engine(fc_parameters, values, offsets, kind="value", device=0, times=None):
    """times: float64 hours since each series' first timestamp (DatetimeIndex data) -> linear_trend_timewise"""
    from tsfresh_amd import _native
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    fplan = compile_fc_parameters(fc_parameters, has_datetime_index=times is not None)
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=device)
    try:
        out = plan.extract_host(values, np.asarray(offsets, dtype=np.int64), times=times)
    finally:
        plan.close()
    return [kind + "__" + n for n in fplan.names], out
