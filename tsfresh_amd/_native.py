"""ctypes binding of libtsfresh_amd.so (include/tsfresh_amd.h).

The library is built in-tree by `tsfresh_amd/csrc/Makefile` (or `__graft_entry__.build()`).  Loading fails loudly
when it is missing, and computing fails loudly when no HIP device is visible: there is no CPU fallback.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSFA_LIB: diagnostics override (the phase-clock build of profiles/phase_ticks.py); the default is the in-tree library
LIB_PATH = os.environ.get("TSFA_LIB") or os.path.join(_HERE, "libtsfresh_amd.so")

TSFA_OK = 0
TSFA_ERR_INVALID = -1
TSFA_ERR_UNSUPPORTED = -2
TSFA_ERR_NO_DEVICE = -3
TSFA_ERR_HIP = -4
TSFA_ERR_TOO_LONG = -5
TSFA_F32, TSFA_F64, TSFA_I64, TSFA_I32 = 0, 1, 2, 3
TSFA_PACK_UNSORTED, TSFA_PACK_VALUE_NAN = 1, 2
TSFA_HOST, TSFA_DEVICE = 0, 1

# every symbol include/tsfresh_amd.h declares
EXPORTS = (
    "tsfa_version", "tsfa_device_count", "tsfa_last_error", "tsfa_calc_id", "tsfa_calc_name", "tsfa_calc_count",
    "tsfa_plan_create", "tsfa_plan_create_with_data", "tsfa_plan_n_cols", "tsfa_plan_destroy", "tsfa_extract", "tsfa_extract_timed",
    "tsfa_extract_windows",
    "tsfa_plan_set_profiling",
    "tsfa_plan_set_option",
    "tsfa_plan_last_timings",
    "tsfa_plan_set_length_hint",
    "tsfa_host_alloc",
    "tsfa_host_free",
    "tsfa_pack_scan",
    "tsfa_pack_offsets",
    "tsfa_impute",
    "tsfa_relevance_classes",
    "tsfa_relevance_classes_ks",
    "tsfa_relevance_real",
    "tsfa_ks_outer_prob",
    "tsfa_device_alloc",
    "tsfa_device_free",
    "tsfa_device_copy",
    "tsfa_gather_columns",
    "tsfa_scatter_columns",
)


class FeatureSpec(ctypes.Structure):
    _fields_ = [("calc", ctypes.c_int32), ("reserved", ctypes.c_int32), ("p", ctypes.c_double * 4)]


class RelevanceCol(ctypes.Structure):
    _fields_ = [("n_unique", ctypes.c_int64), ("v_lo", ctypes.c_double), ("v_hi", ctypes.c_double),
                ("tie_term", ctypes.c_double)]


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("tsfresh_amd native error {}: {}".format(code, message))
        self.code = code


_lib = None


def load():
    """Load (once) and return the ctypes handle; raises ImportError if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "{} is missing: build it with `make -C tsfresh_amd/csrc` (hipcc --offload-arch=gfx950). "
            "tsfresh_amd has no CPU fallback.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    lib.tsfa_version.restype = ctypes.c_int
    lib.tsfa_device_count.restype = ctypes.c_int
    lib.tsfa_last_error.restype = ctypes.c_char_p
    lib.tsfa_calc_id.argtypes = [ctypes.c_char_p]
    lib.tsfa_calc_id.restype = ctypes.c_int
    lib.tsfa_calc_name.argtypes = [ctypes.c_int]
    lib.tsfa_calc_name.restype = ctypes.c_char_p
    lib.tsfa_calc_count.restype = ctypes.c_int
    lib.tsfa_plan_create.argtypes = [ctypes.POINTER(FeatureSpec), ctypes.c_int32, ctypes.c_int32,
                                     ctypes.POINTER(ctypes.c_void_p)]
    lib.tsfa_plan_create.restype = ctypes.c_int
    lib.tsfa_plan_create_with_data.argtypes = [ctypes.POINTER(FeatureSpec), ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    lib.tsfa_plan_create_with_data.restype = ctypes.c_int
    lib.tsfa_plan_n_cols.argtypes = [ctypes.c_void_p]
    lib.tsfa_plan_n_cols.restype = ctypes.c_int32
    lib.tsfa_plan_destroy.argtypes = [ctypes.c_void_p]
    lib.tsfa_plan_destroy.restype = None
    lib.tsfa_extract.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
    lib.tsfa_extract.restype = ctypes.c_int
    lib.tsfa_extract_timed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                       ctypes.c_void_p]
    lib.tsfa_extract_timed.restype = ctypes.c_int
    lib.tsfa_extract_windows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
    lib.tsfa_extract_windows.restype = ctypes.c_int
    lib.tsfa_plan_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.tsfa_plan_set_profiling.restype = ctypes.c_int
    lib.tsfa_plan_last_timings.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p),
                                           ctypes.POINTER(ctypes.c_float), ctypes.c_int32]
    lib.tsfa_plan_last_timings.restype = ctypes.c_int32
    lib.tsfa_pack_scan.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                   ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32),
                                   ctypes.POINTER(ctypes.c_int64)]
    lib.tsfa_pack_scan.restype = ctypes.c_int
    lib.tsfa_pack_offsets.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    lib.tsfa_pack_offsets.restype = ctypes.c_int
    lib.tsfa_host_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    lib.tsfa_host_alloc.restype = ctypes.c_int
    lib.tsfa_host_free.argtypes = [ctypes.c_void_p]
    lib.tsfa_host_free.restype = ctypes.c_int
    lib.tsfa_plan_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
    lib.tsfa_plan_set_option.restype = ctypes.c_int
    _lib = lib
    return lib


def _check(lib, rc):
    if rc != TSFA_OK:
        raise NativeError(rc, lib.tsfa_last_error().decode("utf-8", "replace"))


def set_library_option(name, value):
    """Library-wide option (tsfa_plan_set_option with a NULL plan), e.g. "relevance_batch"."""
    lib = load()
    _check(lib, lib.tsfa_plan_set_option(None, name.encode("ascii"), float(value)))


def device_count():
    return int(load().tsfa_device_count())


def calc_id(name):
    return int(load().tsfa_calc_id(name.encode("ascii")))


# Page-locking memory is slow (~0.1 ms per MB), so freed blocks are kept for reuse, by power-of-two size class, up to
# TSFRESH_AMD_PINNED_POOL_MB (default 4096) in total; a result matrix handed to the caller returns here when the
# caller's DataFrame dies.
_PIN_POOL = {}
_PIN_POOL_BYTES = [0]
_PIN_LOCK = __import__("threading").RLock()   # re-entrant: a cyclic GC pass under the lock may finalize another _PinnedBlock


def _pin_pool_cap():
    return int(os.environ.get("TSFRESH_AMD_PINNED_POOL_MB", "4096")) << 20


class _PinnedBlock:
    """Owner of one tsfa_host_alloc allocation; numpy arrays built on it keep it alive through their base chain."""

    def __init__(self, nbytes):
        lib = load()
        size = 1 << max(20, int(nbytes - 1).bit_length()) if nbytes > (1 << 20) else (1 << 20)
        with _PIN_LOCK:
            free = _PIN_POOL.get(size)
            ptr = free.pop() if free else None
            if ptr is not None:
                _PIN_POOL_BYTES[0] -= size
        if ptr is None:
            p = ctypes.c_void_p()
            _check(lib, lib.tsfa_host_alloc(ctypes.byref(p), size))
            ptr = p.value
        self._lib, self.ptr, self.nbytes, self.size = lib, ptr, int(nbytes), size

    def __del__(self):
        try:
            ptr, self.ptr = getattr(self, "ptr", None), None
            if not ptr:
                return
            with _PIN_LOCK:
                if _PIN_POOL_BYTES[0] + self.size <= _pin_pool_cap():
                    _PIN_POOL.setdefault(self.size, []).append(ptr)
                    _PIN_POOL_BYTES[0] += self.size
                    return
            self._lib.tsfa_host_free(ctypes.c_void_p(ptr))
        except Exception:  # interpreter shutdown
            pass


def release_pinned_pool():
    """Give the pooled page-locked blocks back to the system."""
    lib = load()
    with _PIN_LOCK:
        for size, ptrs in _PIN_POOL.items():
            for ptr in ptrs:
                lib.tsfa_host_free(ctypes.c_void_p(ptr))
        _PIN_POOL.clear()
        _PIN_POOL_BYTES[0] = 0


def pinned_empty(shape, dtype):
    """np.empty(shape, dtype) in page-locked host memory (tsfa_host_alloc): the copy engines reach it directly, so the
    chunked H2D / kernels / D2H pipeline of tsfa_extract(TSFA_HOST) runs at PCIe rate.  Returned to the pool with the
    last view."""
    dtype = np.dtype(dtype)
    shape = tuple(int(d) for d in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    block = _PinnedBlock(n * dtype.itemsize)
    buf = (ctypes.c_char * max(block.nbytes, 1)).from_address(block.ptr)
    buf._tsfa_owner = block  # array -> memoryview -> buf -> block: the allocation lives as long as any view of it
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


_PACK_TYPES = {np.dtype(np.float32): TSFA_F32, np.dtype(np.float64): TSFA_F64, np.dtype(np.int64): TSFA_I64,
               np.dtype(np.int32): TSFA_I32}


def pack_scan(ids, sort_values, values):
    """One multi-threaded pass over the rows of a long frame (tsfa_pack_scan): -> (flags, offsets or None).
    Returns None when a column has a layout / element type the native scan does not take (the caller uses numpy)."""
    cols = []
    for a in (ids, sort_values, values):
        if a is None:
            cols.append((None, 0))
            continue
        if isinstance(a, np.ndarray) and a.dtype.kind in "mM" and a.dtype.itemsize == 8:
            a = a.view(np.int64)  # datetime64 / timedelta64 sort keys compare like their integer ticks (no NaT: checked)
        if not isinstance(a, np.ndarray) or a.ndim != 1 or not a.flags.c_contiguous or a.dtype not in _PACK_TYPES:
            return None
        cols.append((a.ctypes.data_as(ctypes.c_void_p), _PACK_TYPES[a.dtype]))
    lib = load()
    flags, groups = ctypes.c_int32(0), ctypes.c_int64(0)
    n = len(ids)
    _check(lib, lib.tsfa_pack_scan(cols[0][0], cols[0][1], cols[1][0], cols[1][1], cols[2][0], cols[2][1], n,
                                   ctypes.byref(flags), ctypes.byref(groups)))
    if flags.value & TSFA_PACK_UNSORTED:
        return flags.value, None
    offsets = np.empty(groups.value + 1, dtype=np.int64)
    _check(lib, lib.tsfa_pack_offsets(offsets.ctypes.data_as(ctypes.c_void_p), groups.value, n))
    return flags.value, offsets


def _result_matrix(n_rows, n_cols):
    """The feature matrix of a host-side extraction: page-locked when it is large enough for the copy-out to matter."""
    if n_rows * n_cols * 8 >= (4 << 20) and os.environ.get("TSFRESH_AMD_PINNED", "1") != "0":
        return pinned_empty((n_rows, n_cols), np.float64)
    return np.empty((n_rows, n_cols), dtype=np.float64)


class Plan:
    """Owns a `tsfa_plan*`: the compiled list of output columns of one kind on one device."""

    def __init__(self, specs, device=0):
        """specs: iterable of (calc_id, (p0, p1, p2, p3)).  A parameter tuple longer than four carries an ARRAY-valued
        parameter in p[4:] (query_similarity_count's query: registry._query_similarity_encode); those samples go to the plan's
        float64 pool and p[2] becomes their offset (tsfa_plan_create_with_data)."""
        lib = load()
        specs = list(specs)
        arr = (FeatureSpec * max(len(specs), 1))()
        pool = []
        for i, (cid, p) in enumerate(specs):
            arr[i].calc = int(cid)
            arr[i].reserved = 0
            for k in range(4):
                arr[i].p[k] = float(p[k])
            if len(p) > 4:
                arr[i].p[2] = float(len(pool))
                pool.extend(float(v) for v in p[4:])
        handle = ctypes.c_void_p()
        if pool:
            pool_arr = np.ascontiguousarray(pool, dtype=np.float64)
            _check(lib, lib.tsfa_plan_create_with_data(arr, len(specs), pool_arr.ctypes.data, len(pool_arr), int(device),
                                                       ctypes.byref(handle)))
        else:
            _check(lib, lib.tsfa_plan_create(arr, len(specs), int(device), ctypes.byref(handle)))
        self._lib = lib
        self._h = handle
        self.n_cols = len(specs)
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tsfa_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter exit the HIP runtime may already be gone: leak rather than call into it
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    def set_option(self, name, value=1.0):
        """A launch option of this plan (include/tsfresh_amd.h: tsfa_plan_set_option): an alternative route to the same
        numbers, or a diagnostic pre-fill.  Environment variables do not reach the kernels."""
        _check(self._lib, self._lib.tsfa_plan_set_option(self._h, name.encode("ascii"), float(value)))

    def set_profiling(self, enable=True):
        _check(self._lib, self._lib.tsfa_plan_set_profiling(self._h, 1 if enable else 0))

    def set_length_hint(self, min_len, max_len):
        """Promise the length range of every following batch (skips the per-call length scan and its host sync)."""
        self._lib.tsfa_plan_set_length_hint.restype = ctypes.c_int32
        self._lib.tsfa_plan_set_length_hint.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        _check(self._lib, self._lib.tsfa_plan_set_length_hint(self._h, int(min_len), int(max_len)))

    def last_timings(self):
        cap = 32
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        n = self._lib.tsfa_plan_last_timings(self._h, names, ms, cap)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def extract_host(self, values, offsets, times=None, out=None):
        """values: 1-D float32/float64 ndarray; offsets: int64 ndarray (n_series + 1) -> float64 [n_series, n_cols].
        times: float64 ndarray laid out like `values` (hours since each series' first timestamp) for plans that
        hold linear_trend_timewise columns.  out: optional C-contiguous-row float64 [n_series, >= n_cols] target (a row
        slice of a larger matrix); default: a new page-locked matrix."""
        values = np.ascontiguousarray(values)
        if values.dtype == np.float32:
            dt = TSFA_F32
        else:
            values = np.ascontiguousarray(values, dtype=np.float64)
            dt = TSFA_F64
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n_series = offsets.shape[0] - 1
        if out is None:
            out = _result_matrix(n_series, self.n_cols)
        elif (out.dtype != np.float64 or out.ndim != 2 or out.shape[0] != n_series or out.shape[1] < self.n_cols
              or out.strides[1] != 8 or out.strides[0] % 8):
            raise ValueError("out must be a float64 [n_series, >= n_cols] matrix with contiguous rows")
        ld = out.strides[0] // 8 if n_series else self.n_cols
        if n_series == 0 or self.n_cols == 0:
            return out
        if values.size == 0:
            raise ValueError("every series must hold at least one sample")
        tptr = None
        if times is not None:
            times = np.ascontiguousarray(times, dtype=np.float64)
            if times.shape != values.shape:
                raise ValueError("times must have one entry per sample")
            tptr = times.ctypes.data_as(ctypes.c_void_p)
        _check(self._lib, self._lib.tsfa_extract_timed(
            self._h, values.ctypes.data_as(ctypes.c_void_p), dt, tptr, offsets.ctypes.data_as(ctypes.c_void_p),
            n_series, out.ctypes.data_as(ctypes.c_void_p), ld, TSFA_HOST, None))
        return out

    def extract_windows_host(self, values, starts, ends, times=None):
        """Window views of one shared buffer: series s = values[starts[s]:ends[s]] (views may overlap) ->
        float64 [n_windows, n_cols].  The rolled (forecasting) layout without materialising the windows."""
        values = np.ascontiguousarray(values)
        if values.dtype == np.float32:
            dt = TSFA_F32
        else:
            values = np.ascontiguousarray(values, dtype=np.float64)
            dt = TSFA_F64
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        ends = np.ascontiguousarray(ends, dtype=np.int64)
        if starts.shape != ends.shape or starts.ndim != 1:
            raise ValueError("starts and ends must be 1-D arrays of the same length")
        n = starts.shape[0]
        out = _result_matrix(n, self.n_cols)
        if n == 0 or self.n_cols == 0:
            return out
        if starts.min() < 0 or ends.max() > values.shape[0]:
            raise ValueError("a window reaches outside the value buffer")
        tptr = None
        if times is not None:
            times = np.ascontiguousarray(times, dtype=np.float64)
            tptr = times.ctypes.data_as(ctypes.c_void_p)
        _check(self._lib, self._lib.tsfa_extract_windows(
            self._h, values.ctypes.data_as(ctypes.c_void_p), dt, tptr, starts.ctypes.data_as(ctypes.c_void_p),
            ends.ctypes.data_as(ctypes.c_void_p), n, out.ctypes.data_as(ctypes.c_void_p), self.n_cols, TSFA_HOST, None))
        return out

    def extract_into(self, values, offsets, matrix, col0=0, times=None):
        """Host arrays in, columns [col0, col0 + n_cols) of the DeviceMatrix `matrix` out: the samples are copied to the
        device once and the feature rows stay there (tsfa_extract / tsfa_extract_timed with TSFA_DEVICE pointers)."""
        values = np.ascontiguousarray(values)
        if values.dtype == np.float32:
            dt = TSFA_F32
        else:
            values = np.ascontiguousarray(values, dtype=np.float64)
            dt = TSFA_F64
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n_series = offsets.shape[0] - 1
        if matrix.shape[0] != n_series or col0 < 0 or col0 + self.n_cols > matrix.shape[1]:
            raise ValueError("the device matrix does not hold [n_series, col0 + n_cols] cells")
        if n_series == 0 or self.n_cols == 0:
            return
        if values.size == 0:
            raise ValueError("every series must hold at least one sample")
        bufs = [_DeviceBuffer(self._lib, values, matrix.device), _DeviceBuffer(self._lib, offsets, matrix.device)]
        try:
            tptr = None
            if times is not None:
                times = np.ascontiguousarray(times, dtype=np.float64)
                if times.shape != values.shape:
                    raise ValueError("times must have one entry per sample")
                bufs.append(_DeviceBuffer(self._lib, times, matrix.device))
                tptr = ctypes.c_void_p(bufs[2].ptr)
            _check(self._lib, self._lib.tsfa_extract_timed(
                self._h, ctypes.c_void_p(bufs[0].ptr), dt, tptr, ctypes.c_void_p(bufs[1].ptr), n_series,
                ctypes.c_void_p(matrix.ptr + 8 * col0), matrix.ld, TSFA_DEVICE, None))
        finally:
            for bf in bufs:
                bf.free()

    def extract_device(self, values_ptr, dtype, offsets_ptr, n_series, out_ptr, ld_out, stream=None):
        """Raw device-pointer entry (ints): used with torch tensors (`.data_ptr()`) by bench.py / the sharded path."""
        _check(self._lib, self._lib.tsfa_extract(
            self._h, ctypes.c_void_p(values_ptr), int(dtype), ctypes.c_void_p(offsets_ptr), int(n_series),
            ctypes.c_void_p(out_ptr), int(ld_out), TSFA_DEVICE, ctypes.c_void_p(stream) if stream else None))


class DeviceMatrix:
    """Row-major float64 [n_rows, n_cols] in the HBM of one device (tsfa_device_alloc): the feature matrix of the chain
    extract -> impute -> select when it never leaves the GPU (convenience.extract_relevant_features(device_resident=True)).
    Owns its memory; `to_host(cols)` fetches the named column indices (tsfa_gather_columns) or everything."""

    def __init__(self, n_rows, n_cols, device=0):
        self._lib = load()
        self.shape = (int(n_rows), int(n_cols))
        self.device = int(device)
        self._ptr = ctypes.c_void_p()
        _bind_device_api(self._lib)
        _check(self._lib, self._lib.tsfa_device_alloc(ctypes.byref(self._ptr), max(1, 8 * self.shape[0] * self.shape[1]),
                                                      self.device))

    @property
    def ptr(self):
        return self._ptr.value

    @property
    def ld(self):
        return self.shape[1]

    def free(self):
        if getattr(self, "_ptr", None) is not None and self._ptr.value:
            self._lib.tsfa_device_free(self._ptr, self.device)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        import sys
        if sys is not None and not sys.is_finalizing():
            try:
                self.free()
            except Exception:
                pass

    def to_host(self, cols=None):
        n, m = self.shape
        if cols is None:
            out = np.empty((n, m), dtype=np.float64)
            _check(self._lib, self._lib.tsfa_device_copy(out.ctypes.data_as(ctypes.c_void_p), self._ptr, out.nbytes, 0,
                                                         self.device))
            return out
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        out = np.empty((n, cols.shape[0]), dtype=np.float64)
        _check(self._lib, self._lib.tsfa_gather_columns(self._ptr, n, m, cols.ctypes.data_as(ctypes.c_void_p),
                                                        cols.shape[0], out.ctypes.data_as(ctypes.c_void_p), self.device))
        return out


def extract_parts_into(parts, values, offsets, matrix, col0=0, times=None):
    """Several native plans over ONE upload of the samples: parts = [(Plan, column indices in the caller's order)].  Every
    part's block is extracted into a transient device matrix and scattered (tsfa_scatter_columns) into columns
    col0 + cols of `matrix` (a DeviceMatrix): the composite plans of feature_extraction/extraction.py (several ADF lag
    selections, more than 128 CWT columns) without an upload per part, and with the matrix staying in HBM."""
    lib = load()
    _bind_device_api(lib)
    values = np.ascontiguousarray(values)
    if values.dtype == np.float32:
        dt = TSFA_F32
    else:
        values = np.ascontiguousarray(values, dtype=np.float64)
        dt = TSFA_F64
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_series = offsets.shape[0] - 1
    if n_series == 0:
        return
    if values.size == 0:
        raise ValueError("every series must hold at least one sample")
    device = matrix.device
    bufs = [_DeviceBuffer(lib, values, device), _DeviceBuffer(lib, offsets, device)]
    tmp = None
    try:
        tptr = None
        if times is not None:
            times = np.ascontiguousarray(times, dtype=np.float64)
            bufs.append(_DeviceBuffer(lib, times, device))
            tptr = ctypes.c_void_p(bufs[2].ptr)
        width = max(plan.n_cols for plan, _ in parts)
        tmp = DeviceMatrix(n_series, width, device)
        for plan, cols in parts:
            if plan.n_cols == 0:
                continue
            _check(lib, lib.tsfa_extract_timed(plan._h, ctypes.c_void_p(bufs[0].ptr), dt, tptr, ctypes.c_void_p(bufs[1].ptr),
                                               n_series, ctypes.c_void_p(tmp.ptr), plan.n_cols, TSFA_DEVICE, None))
            idx = np.ascontiguousarray(np.asarray(cols, dtype=np.int64) + int(col0), dtype=np.int32)
            _check(lib, lib.tsfa_scatter_columns(ctypes.c_void_p(matrix.ptr), matrix.ld, idx.ctypes.data_as(ctypes.c_void_p),
                                                 ctypes.c_void_p(tmp.ptr), n_series, plan.n_cols, device))
    finally:
        for bf in bufs:
            bf.free()
        if tmp is not None:
            tmp.free()


def _bind_device_api(lib):
    if getattr(lib, "_tsfa_device_api_bound", False):
        return
    lib.tsfa_device_alloc.restype = ctypes.c_int32
    lib.tsfa_device_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_int32]
    lib.tsfa_device_free.restype = ctypes.c_int32
    lib.tsfa_device_free.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.tsfa_device_copy.restype = ctypes.c_int32
    lib.tsfa_device_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32]
    lib.tsfa_gather_columns.restype = ctypes.c_int32
    lib.tsfa_gather_columns.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_int32]
    lib.tsfa_scatter_columns.restype = ctypes.c_int32
    lib.tsfa_scatter_columns.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int32]
    lib._tsfa_device_api_bound = True


class _DeviceBuffer:
    """A transient device copy of a host array (inputs of Plan.extract_into)."""

    def __init__(self, lib, array, device):
        self._lib, self.device = lib, device
        self._ptr = ctypes.c_void_p()
        _bind_device_api(lib)
        _check(lib, lib.tsfa_device_alloc(ctypes.byref(self._ptr), max(1, array.nbytes), device))
        _check(lib, lib.tsfa_device_copy(self._ptr, array.ctypes.data_as(ctypes.c_void_p), array.nbytes, 1, device))

    @property
    def ptr(self):
        return self._ptr.value

    def free(self):
        if self._ptr.value:
            self._lib.tsfa_device_free(self._ptr, self.device)
            self._ptr = ctypes.c_void_p()


def _matrix_args(X):
    """(pointer, n_rows, n_cols, ld, space, keepalive) of a host ndarray or a DeviceMatrix."""
    if isinstance(X, DeviceMatrix):
        return ctypes.c_void_p(X.ptr), X.shape[0], X.shape[1], X.ld, TSFA_DEVICE, X
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("X must be two-dimensional")
    return X.ctypes.data_as(ctypes.c_void_p), X.shape[0], X.shape[1], X.shape[1], TSFA_HOST, X


def relevance_classes(X, y_codes, n_classes, device=0, with_ks=False):
    """Per-column relevance statistics of the row-major float64 matrix X against class codes (tsfa_relevance_classes).
    -> (n_unique int64[m], v_lo[m], v_hi[m], tie_term[m], rank_sums[m, C], hi_counts[m, C]) (+ ks_d[m, C] with_ks)."""
    lib = load()
    xptr, n, m, ld, space, _keep = _matrix_args(X)
    y_codes = np.ascontiguousarray(y_codes, dtype=np.int32)
    if y_codes.shape != (n,):
        raise ValueError("one class code per row")
    cols = (RelevanceCol * max(m, 1))()
    rank_sums = np.zeros((m, n_classes), dtype=np.float64)
    hi_counts = np.zeros((m, n_classes), dtype=np.int64)
    lib.tsfa_relevance_classes.restype = ctypes.c_int32
    lib.tsfa_relevance_classes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]
    if with_ks:
        ks_d = np.zeros((m, n_classes), dtype=np.float64)
        lib.tsfa_relevance_classes_ks.restype = ctypes.c_int32
        lib.tsfa_relevance_classes_ks.argtypes = lib.tsfa_relevance_classes.argtypes + [ctypes.c_void_p]
        _check(lib, lib.tsfa_relevance_classes_ks(xptr, n, m, ld, space,
                                                  y_codes.ctypes.data_as(ctypes.c_void_p), int(n_classes), int(device),
                                                  ctypes.cast(cols, ctypes.c_void_p), rank_sums.ctypes.data_as(ctypes.c_void_p),
                                                  hi_counts.ctypes.data_as(ctypes.c_void_p), ks_d.ctypes.data_as(ctypes.c_void_p)))
    else:
        _check(lib, lib.tsfa_relevance_classes(xptr, n, m, ld, space,
                                               y_codes.ctypes.data_as(ctypes.c_void_p), int(n_classes), int(device),
                                               ctypes.cast(cols, ctypes.c_void_p), rank_sums.ctypes.data_as(ctypes.c_void_p),
                                               hi_counts.ctypes.data_as(ctypes.c_void_p)))
    rec = np.frombuffer(cols, dtype=[("n_unique", "<i8"), ("v_lo", "<f8"), ("v_hi", "<f8"), ("tie_term", "<f8")], count=m)
    res = (rec["n_unique"].copy(), rec["v_lo"].copy(), rec["v_hi"].copy(), rec["tie_term"].copy(), rank_sums, hi_counts)
    return res + (ks_d,) if with_ks else res


_REAL_COL_DTYPE = [("n_unique", "<i8"), ("v_lo", "<f8"), ("v_hi", "<f8"), ("dis", "<i8"), ("xtie", "<i8"), ("ntie", "<i8"),
                   ("x0", "<f8"), ("x1", "<f8"), ("n_hi", "<i8"), ("ks_d", "<f8")]


def relevance_real(X, y, device=0):
    """Per-column statistics of the row-major float64 matrix X against a real-valued target (tsfa_relevance_real).
    -> (structured array with the fields of tsfa_relevance_real_col, dense ranks of y)."""
    lib = load()
    xptr, n, m, ld, space, _keep = _matrix_args(X)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if y.shape != (n,):
        raise ValueError("one target value per row")
    # the target is one vector: its dense ranks / sorted order are prepared here once for all columns
    uniq, inverse = np.unique(y, return_inverse=True)
    y_rank = np.ascontiguousarray(inverse, dtype=np.int32)
    y_perm = np.ascontiguousarray(np.argsort(y, kind="stable"), dtype=np.int32)
    ys = y[y_perm]
    y_end = np.ones(n, dtype=np.uint8)
    if n > 1:
        y_end[:-1] = ys[1:] != ys[:-1]
    cols = np.zeros(max(m, 1), dtype=_REAL_COL_DTYPE)
    lib.tsfa_relevance_real.restype = ctypes.c_int32
    lib.tsfa_relevance_real.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    _check(lib, lib.tsfa_relevance_real(xptr, n, m, ld, space,
                                        y_rank.ctypes.data_as(ctypes.c_void_p), y_perm.ctypes.data_as(ctypes.c_void_p),
                                        y_end.ctypes.data_as(ctypes.c_void_p), int(device),
                                        cols.ctypes.data_as(ctypes.c_void_p)))
    return cols[:m], y_rank


def impute_matrix(X, device=0):
    """In-place impute of the C-contiguous-row float64 matrix X on the device (tsfa_impute).
    -> (col_max, col_min, col_median of the finite values, number of finite cells per column)."""
    lib = load()
    if isinstance(X, DeviceMatrix):
        xptr, (n, m), ld, space = ctypes.c_void_p(X.ptr), X.shape, X.ld, TSFA_DEVICE
        device = X.device
    else:
        if not (isinstance(X, np.ndarray) and X.dtype == np.float64 and X.ndim == 2 and X.flags.writeable
                and (X.shape[1] == 0 or X.strides[1] == 8) and X.strides[0] % 8 == 0 and X.strides[0] >= 8 * X.shape[1]):
            raise ValueError("impute_matrix needs a writeable float64 matrix with contiguous rows")
        n, m = X.shape
        xptr, ld, space = X.ctypes.data_as(ctypes.c_void_p), (X.strides[0] // 8 if n else m), TSFA_HOST
    cmax, cmin, cmed = (np.zeros(m) for _ in range(3))
    cnt = np.zeros(m, dtype=np.int32)
    lib.tsfa_impute.restype = ctypes.c_int32
    lib.tsfa_impute.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    _check(lib, lib.tsfa_impute(xptr, n, m, ld, space, int(device),
                                cmax.ctypes.data_as(ctypes.c_void_p), cmin.ctypes.data_as(ctypes.c_void_p),
                                cmed.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
    return cmax, cmin, cmed, cnt


def ks_outer_prob(m, n, g, h):
    lib = load()
    lib.tsfa_ks_outer_prob.restype = ctypes.c_double
    lib.tsfa_ks_outer_prob.argtypes = [ctypes.c_int64] * 4
    return float(lib.tsfa_ks_outer_prob(int(m), int(n), int(g), int(h)))
