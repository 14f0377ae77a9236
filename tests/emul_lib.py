"""Test helper: build + bind tests/emul/libtsfa_emul.so, the single-thread g++ build of the kernel sources.

TEST INFRASTRUCTURE ONLY.  The product (`tsfresh_amd`) never imports this; it exists because the build container
has no GPU and GPU minutes are rationed, so kernel *logic* is first checked here against the oracle.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emul", "emul.cpp")
LIB = os.path.join(HERE, "emul", "libtsfa_emul.so")
CSRC = os.path.join(HERE, "..", "tsfresh_amd", "csrc")


class _Spec(ctypes.Structure):
    _fields_ = [("calc", ctypes.c_int32), ("reserved", ctypes.c_int32), ("p", ctypes.c_double * 4)]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def load():
    if _stale():
        tmp = "%s.%d.tmp" % (LIB, os.getpid())  # atomic: several processes (the gloo test's ranks) may build at once
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DTSFA_EMUL",
                               SRC, "-o", tmp])
        os.replace(tmp, LIB)
    lib = ctypes.CDLL(LIB)
    lib.tsfa_emul_calc_id.argtypes = [ctypes.c_char_p]
    lib.tsfa_emul_calc_id.restype = ctypes.c_int
    lib.tsfa_emul_extract.argtypes = [ctypes.POINTER(_Spec), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int]
    lib.tsfa_emul_extract.restype = ctypes.c_int
    lib.tsfa_emul_extract_timed.argtypes = [ctypes.POINTER(_Spec), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_char_p, ctypes.c_int]
    lib.tsfa_emul_extract_timed.restype = ctypes.c_int
    lib.tsfa_emul_set_pool.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    lib.tsfa_emul_set_pool.restype = None
    return lib


def emul_extract_specs(specs, values, offsets, times=None):
    """specs: [(calculator id of the EMULATION library, p[4])] -> float64 matrix [n_series x len(specs)]."""
    lib = load()
    arr = (_Spec * max(len(specs), 1))()
    pool = []
    for i, (cid, p) in enumerate(specs):
        arr[i].calc = cid
        for k in range(4):
            arr[i].p[k] = p[k]
        if len(p) > 4:   # an array-valued parameter (query_similarity_count): p[4:] goes to the pool, p[2] = its offset
            arr[i].p[2] = float(len(pool))
            pool.extend(float(v) for v in p[4:])
    pool_arr = np.ascontiguousarray(pool, dtype=np.float64)
    lib.tsfa_emul_set_pool(pool_arr.ctypes.data, len(pool_arr))
    values = np.ascontiguousarray(values, dtype=np.float64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    out = np.empty((n, len(specs)))
    err = ctypes.create_string_buffer(512)
    if times is not None:
        times = np.ascontiguousarray(times, dtype=np.float64)
    rc = lib.tsfa_emul_extract_timed(arr, len(specs), values.ctypes.data, None if times is None else times.ctypes.data,
                                     offsets.ctypes.data, n, out.ctypes.data, len(specs), err, 512)
    if rc != 0:
        raise RuntimeError("emul: %d %s" % (rc, err.value.decode()))
    return out


def emul_extract(fc_parameters, values, offsets, kind="value", times=None):
    """-> (column names, float64 matrix) using the same plan compiler as the product."""
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    lib = load()
    plan = compile_fc_parameters(fc_parameters, has_datetime_index=times is not None)
    specs = plan.native_specs(lambda name: lib.tsfa_emul_calc_id(name.encode()))
    return [kind + "__" + nm for nm in plan.names], emul_extract_specs(specs, values, offsets, times=times)


class EmulPlan:
    """Stands in for tsfresh_amd._native.Plan in CPU tests of the DataFrame-level code (tests/test_frames.py): the same
    `extract_host` contract, the g++ build of the kernel sources behind it."""

    def __init__(self, fplan):
        lib = load()
        self.specs = fplan.native_specs(lambda name: lib.tsfa_emul_calc_id(name.encode()))

    def extract_host(self, values, offsets, times=None):
        return emul_extract_specs(self.specs, values, offsets, times=times)
