"""Three ways to evaluate an FCParameters mapping on a ragged batch, with one signature:
    engine(fc_parameters, values, offsets) -> (column names, float64 matrix)

  oracle : oracle/ (numpy restatement of the reference)                          CPU
  emul   : tests/emul (g++ single-thread build of the kernel sources)            CPU, logic only
  hip    : tsfresh_amd._native.Plan -> libtsfresh_amd.so -> HIP kernels          GPU (the product)
"""
import numpy as np


def oracle_engine(fc_parameters, values, offsets, kind="value"):
    from oracle.extract import oracle_matrix
    return oracle_matrix(np.asarray(values, dtype=np.float64), offsets, fc_parameters, kind=kind)


def emul_engine(fc_parameters, values, offsets, kind="value"):
    from emul_lib import emul_extract
    return emul_extract(fc_parameters, values, offsets, kind=kind)


def hip_engine(fc_parameters, values, offsets, kind="value", device=0):
    from tsfresh_amd import _native
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    fplan = compile_fc_parameters(fc_parameters)
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=device)
    try:
        out = plan.extract_host(values, np.asarray(offsets, dtype=np.int64))
    finally:
        plan.close()
    return [kind + "__" + n for n in fplan.names], out
