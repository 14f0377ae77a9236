"""`extract_features`: the reference's public entry point, with the hot path on the GPU.

Signature and return contract follow tsfresh/feature_extraction/extraction.py:30-190 (`extract_features`) and
:193-305 (`_do_extraction`): a pandas container in, a float64 DataFrame out whose index holds the ids (original
dtype, sorted) and whose columns are named ``"{kind}__{calculator}__{parameters}"``.  What changes is the inside:

    reference                                              here
    ------------------------------------------------------------------------------------------------
    to_tsdata -> iterable of pd.Series  (data.py:447)      pack_timeseries -> ragged buffers (data.py)
    distributor.map_reduce(_do_extraction_on_chunk, ...)   Plan.extract_host -> tsfa_extract (C-ABI, HIP)
    data.pivot(list of (id, name, value))  (data.py:86)    the dense matrix IS the pivot

Arguments that only steer the reference's CPU distributors (`n_jobs`, `chunksize`, `disable_progressbar`,
`show_warnings`) are accepted and ignored.  `distributor` may be a `tsfresh_amd.utilities.distribution.GPUDistributor`
(only its device is used) or any other `DistributorBaseClass` (ignored with a warning); an object that is no distributor
raises the reference's ValueError.
"""
import collections
import os
import threading
import warnings

import numpy as np
import pandas as pd

from tsfresh_amd import _native
from tsfresh_amd.feature_extraction.data import pack_timeseries
from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
from tsfresh_amd.feature_extraction.reference_errors import check_reference_data_errors
from tsfresh_amd.feature_extraction.settings import ComprehensiveFCParameters


def _default_device():
    if "TSFRESH_AMD_DEVICE" in os.environ:
        return int(os.environ["TSFRESH_AMD_DEVICE"])
    if "LOCAL_RANK" in os.environ:  # one process per GPU under torch.distributed.run
        return int(os.environ["LOCAL_RANK"]) % max(_native.device_count(), 1)
    return 0


# Native plans (device-side spec tables, twiddles, staging buffers) are kept across calls: creating one costs ~8 ms,
# a 20 000-series extraction ~25 ms.  One plan may be driven by one host thread at a time (ctypes releases the GIL inside
# tsfa_extract), so every thread keeps its OWN small LRU: no thread can evict -- and destroy -- a plan another thread
# is running, and no lock is held across the native call.
_PLAN_CACHE_SIZE = 6
_BATCH_KINDS_MAX_SAMPLES = 4_000_000   # kinds sharing a plan are concatenated into one native call up to this size
_TLS = threading.local()
# sample_entropy / approximate_entropy are O(n^2): up to this length the bit-matrix sweeps serve them (fam_entropy_bits.h in LDS
# to 4096 samples: 12 ms per 100 000 series of 1024; fam_entropy_hbits.h with the per-sample arrays in HBM to 17 408: 50 us per
# series of 16 384 with the GPU full); beyond it the float64 pair sweep of the long-series build takes over -- the call returns
# the right values, for any length, but these two calculators are then nearly all of its time and the caller should know
ENTROPY_FAST_MAX_LEN = 17408
_QUADRATIC = ("sample_entropy", "approximate_entropy")


_LONG_ENTROPY_WARNED = False


def _warn_long_entropy(fc_parameters, pk, show_warnings=False):
    """A run-time warning, not a calculator's: with show_warnings=False (the reference's default, which silences what the
    calculators emit) it is raised ONCE per process instead of on every call (round-4 ADVICE); show_warnings=True: every call."""
    global _LONG_ENTROPY_WARNED
    if _LONG_ENTROPY_WARNED and not show_warnings:
        return
    try:
        quadratic = any(name in fc_parameters for name in _QUADRATIC)
    except TypeError:
        quadratic = False
    if pk.n_series == 0 or not quadratic:
        return
    longest = int(np.diff(pk.offsets).max())
    if longest > ENTROPY_FAST_MAX_LEN:
        _LONG_ENTROPY_WARNED = True
        warnings.warn("kind {!r}: series of up to {} samples with sample_entropy / approximate_entropy in the settings: these "
                      "calculators are O(n^2) and beyond {} samples leave the bit-matrix sweeps for a float64 pair sweep (several "
                      "milliseconds per series at 32 768 samples on a full MI355X, four times that per doubling: nearly all of the "
                      "extraction).  EfficientFCParameters() leaves them out, as the reference recommends for long series.".format(pk.kind, longest, ENTROPY_FAST_MAX_LEN),
                      UserWarning, stacklevel=3)


def _thread_cache():
    cache = getattr(_TLS, "plans", None)
    if cache is None:
        cache = _TLS.plans = collections.OrderedDict()   # dies with the thread; the plans close in Plan.__del__
    return cache


_CWT_COLUMNS_PER_PLAN = 128   # tsfa_api.cpp: "more than 128 cwt_coefficients columns in one plan"


class _CompositePlan:
    """Several native plans behind one plan's `extract_host` / `extract_windows_host`: the columns of a settings object that
    one native plan cannot hold together: `augmented_dickey_fuller` with more than one `autolag` value (fc.py:499-545
    evaluates every dict of the list on its own; the kernels hold ONE lag-search fit per series, so `tsfa_plan_create` takes
    one value per plan) and more than 128 `cwt_coefficients` columns (the filter bank of one plan).  The first part carries
    every other column + the first lag selection's + the first 128 CWT columns, the others one lag selection / 128 CWT
    columns each; results are scattered into the caller's column order."""

    def __init__(self, parts, n_cols):
        self.parts = parts          # [(native plan, column indices in the full matrix)]
        self.n_cols = int(n_cols)

    def _gather(self, n_rows, run):
        out = _native._result_matrix(n_rows, self.n_cols)
        for plan, cols in self.parts:
            out[:, cols] = run(plan)
        return out

    def extract_host(self, values, offsets, times=None, out=None):
        # one upload of the samples for all parts; the parts' columns meet in a device matrix that comes back in one copy
        n_rows = len(offsets) - 1
        if not all(isinstance(pl, _native.Plan) for pl, _ in self.parts):   # (duck-typed parts: the CPU tests' emulation plans)
            res = self._gather(n_rows, lambda pl: pl.extract_host(values, offsets, times=times))
            if out is not None:
                out[:, :self.n_cols] = res
                return out
            return res
        dm = _native.DeviceMatrix(n_rows, self.n_cols, self.parts[0][0].device)
        try:
            self.extract_into(values, offsets, dm, times=times)
            res = dm.to_host()
        finally:
            dm.free()
        if out is not None:
            out[:, :self.n_cols] = res
            return out
        return res

    def extract_windows_host(self, values, starts, ends, times=None):
        return self._gather(len(starts), lambda pl: pl.extract_windows_host(values, starts, ends, times=times))

    def extract_into(self, values, offsets, matrix, col0=0, times=None):
        """Columns [col0, col0 + n_cols) of the DeviceMatrix `matrix` (device_resident=True: the feature matrix never leaves
        HBM): every part extracts into a transient block that tsfa_scatter_columns puts in the caller's column order."""
        _native.extract_parts_into(self.parts, values, offsets, matrix, col0=col0, times=times)


def _split_native_specs(specs):
    """-> [(sub-list of specs, their column indices)]: one part unless augmented_dickey_fuller columns name several lag
    selections (p[1]; attr code 3 -- a column that is NaN whatever the fit -- belongs to any part) or the plan holds more
    cwt_coefficients columns than one native plan takes."""
    adf = _native.calc_id("augmented_dickey_fuller")
    modes = []
    for cid, p in specs:
        if cid == adf and int(p[0]) != 3 and float(p[1]) not in modes:
            modes.append(float(p[1]))
    parts = [([], []) for _ in (modes or [0.0])]
    cwt = _native.calc_id("cwt_coefficients")
    n_cwt, overflow = 0, []
    for j, (cid, p) in enumerate(specs):
        if cid == cwt:
            # the MFMA filter bank of one plan holds _CWT_COLUMNS_PER_PLAN columns (tsfa_plan_create refuses more): the rest
            # goes to plans of its own (15 coefficients x 10 widths is already 150 columns)
            n_cwt += 1
            if n_cwt > _CWT_COLUMNS_PER_PLAN:
                k = (n_cwt - 1) // _CWT_COLUMNS_PER_PLAN - 1
                while len(overflow) <= k:
                    overflow.append(([], []))
                overflow[k][0].append((cid, p))
                overflow[k][1].append(j)
                continue
        k = modes.index(float(p[1])) if (cid == adf and int(p[0]) != 3) else 0
        parts[k][0].append((cid, p))
        parts[k][1].append(j)
    return [pt for pt in parts + overflow if pt[0]]


def _acquire_plan(fplan, device, pins=None):
    specs = list(fplan.native_specs(_native.calc_id))
    parts = _split_native_specs(specs)
    if len(parts) == 1:
        return _acquire_plan_specs(specs, device, pins)
    # every part is pinned while the others are acquired (a composite of more parts than the cache holds must not close its
    # own first part: round-5 ADVICE); a caller without a pin set gets a local one for the duration of the acquisition
    local = pins if pins is not None else set()
    return _CompositePlan([(_acquire_plan_specs(sub, device, local), cols) for sub, cols in parts], len(specs))


def _trim_cache(cache, pins=()):
    """Close the oldest plans beyond _PLAN_CACHE_SIZE, never one whose key is in `pins` (the plans the running call
    still holds: a frame of more than _PLAN_CACHE_SIZE kinds with distinct settings keeps them all until it returns)."""
    if len(cache) <= _PLAN_CACHE_SIZE:
        return
    for key in [k for k in cache if k not in pins]:
        if len(cache) <= _PLAN_CACHE_SIZE:
            break
        cache.pop(key).close()  # this thread's own, idle plan


def _acquire_plan_specs(specs, device, pins=None):
    """The calling thread's cached native plan for `specs` on `device`.  `pins`: a set the caller owns for the duration
    of ONE extract call; every key acquired through it is exempt from eviction until the caller trims the cache."""
    specs = list(specs)
    key = (int(device), tuple((cid, tuple(float(v) for v in p)) for cid, p in specs))
    cache = _thread_cache()
    if pins is not None:
        pins.add(key)
    plan = cache.get(key)
    if plan is not None:
        cache.move_to_end(key)
        return plan
    plan = _native.Plan(specs, device=device)
    cache[key] = plan
    _trim_cache(cache, pins if pins is not None else (key,))
    return plan


def clear_plan_cache():
    """Release the cached native plans of the CALLING thread (and the device memory they hold).  Plans cached by other
    threads are released when those threads call this, or with the interpreter."""
    cache = _thread_cache()
    while cache:
        _, old = cache.popitem(last=False)
        old.close()


def extract_features(
    timeseries_container,
    default_fc_parameters=None,
    kind_to_fc_parameters=None,
    column_id=None,
    column_sort=None,
    column_kind=None,
    column_value=None,
    chunksize=None,
    n_jobs=None,
    show_warnings=False,
    disable_progressbar=True,
    impute_function=None,
    profile=False,
    profiling_filename=None,
    profiling_sorting=None,
    distributor=None,
    pivot=True,
    device=None,
    devices=None,
):
    """Extract features from a pandas container on one MI355X (or, with `devices`, on several).

    :param timeseries_container: long/wide `pd.DataFrame` or dict of DataFrames (reference data formats).
    :param default_fc_parameters: calculator name -> list of parameter dicts (or None); default
        `ComprehensiveFCParameters()` (extraction.py:149-152).
    :param kind_to_fc_parameters: kind -> FCParameters overriding the default for that kind.
    :param column_id, column_sort, column_kind, column_value: as in the reference.
    :param impute_function: called on the result DataFrame (extraction.py:181-182).
    :param pivot: False returns the flat list of `(id, column name, value)` tuples (extraction.py:301-302).
    :param device: HIP device ordinal (default: $TSFRESH_AMD_DEVICE, else $LOCAL_RANK, else 0).
    :param devices: list of HIP device ordinals: the series of every kind are cut into sum(len^2)-balanced contiguous
        shards, one per device, extracted concurrently from this process (one plan + host thread per device) into one
        page-locked matrix (`tsfresh_amd.distributed.extract_on_devices`).  The reference's counterpart is
        `n_jobs` / a MultiprocessingDistributor over CPU cores (extraction.py:262-275).
    :return: `pd.DataFrame` of dtype float64.
    """
    if default_fc_parameters is None and kind_to_fc_parameters is None:
        default_fc_parameters = ComprehensiveFCParameters()
    elif default_fc_parameters is None and kind_to_fc_parameters is not None:
        default_fc_parameters = {}
    if distributor is not None:
        # extraction.py:285-286 only type-checks the object; any DistributorBaseClass is a valid argument there.  The
        # series go to the GPU whatever it is: a GPUDistributor lends its device, another distributor is thanked and ignored
        from tsfresh_amd.utilities.distribution import GPUDistributor, is_distributor
        if not is_distributor(distributor):
            raise ValueError("the passed distributor is not an DistributorBaseClass object")
        if isinstance(distributor, GPUDistributor):
            if device is None:
                device = distributor.device
        else:
            warnings.warn("distributor {}: tsfresh_amd extracts on the GPU and does not map chunks over it; the argument "
                          "is ignored".format(type(distributor).__name__), UserWarning, stacklevel=2)
    if profile:
        warnings.warn("profile=True (cProfile of the Python calculators) has no meaning for the GPU path; "
                      "use rocprofv3 or Plan.set_profiling instead", stacklevel=2)

    packed, id_dtype, has_dt_index = pack_timeseries(
        timeseries_container, column_id=column_id, column_kind=column_kind, column_value=column_value,
        column_sort=column_sort)
    if devices is not None:
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one HIP device")
        if device is None:
            device = devices[0]
    if device is None:
        device = _default_device()
    for pk in packed:   # outside the filter below: this one is about run time, not about a calculator's domain
        _warn_long_entropy(kind_to_fc_parameters[pk.kind] if kind_to_fc_parameters and pk.kind in kind_to_fc_parameters
                           else default_fc_parameters, pk, show_warnings)

    with warnings.catch_warnings():
        if not show_warnings:
            warnings.simplefilter("ignore")
        else:
            warnings.simplefilter("default")

        blocks = []  # (PackedKind, column names, matrix)
        plan_cache = {}
        pins = set()  # native plans this call holds: not evictable before it returns
        jobs = []    # (PackedKind, FeaturePlan, native plan or None)
        for pk in packed:
            if kind_to_fc_parameters and pk.kind in kind_to_fc_parameters:
                fc_parameters = kind_to_fc_parameters[pk.kind]
            else:
                fc_parameters = default_fc_parameters
            kind_has_dt = pk.times is not None  # extraction.py:349-358 checks the index of each series
            key = (id(fc_parameters), kind_has_dt)
            if key not in plan_cache:
                fplan = compile_fc_parameters(fc_parameters, has_datetime_index=kind_has_dt)
                multi_dev = devices is not None and len(devices) > 1   # extract_on_devices keeps its own plans
                nplan = _acquire_plan(fplan, device, pins) if fplan.names and not multi_dev else None
                plan_cache[key] = (fplan, nplan)
            fplan, nplan = plan_cache[key]
            if len(fplan) == 0:
                continue
            jobs.append((pk, fplan, nplan))

        # Kinds that share a plan and a sample dtype travel in ONE native call while the batch is small: a frame of 6
        # kinds x 88 series x 15 samples (BASELINE configs[0]) is one upload / launch set / download instead of six.
        # Beyond _BATCH_KINDS_MAX_SAMPLES the concatenation copy costs more than the launches it saves.
        matrices = {}
        multi = devices is not None and len(devices) > 1
        groups = collections.OrderedDict()
        for j, (pk, fplan, nplan) in enumerate(jobs):
            if nplan is not None and not multi:
                groups.setdefault((id(nplan), pk.values.dtype.str, pk.times is not None), []).append(j)
        for members in groups.values():
            total = sum(len(jobs[j][0].values) for j in members)
            if len(members) < 2 or total > _BATCH_KINDS_MAX_SAMPLES:
                continue
            pks = [jobs[j][0] for j in members]
            nplan = jobs[members[0]][2]
            values = np.concatenate([pk.values for pk in pks])
            times = np.concatenate([pk.times for pk in pks]) if pks[0].times is not None else None
            offs, base = [np.zeros(1, dtype=np.int64)], 0
            for pk in pks:
                offs.append(np.asarray(pk.offsets[1:], dtype=np.int64) + base)
                base += int(pk.offsets[-1])
            big = nplan.extract_host(values, np.concatenate(offs), times=times)
            r0 = 0
            for j, pk in zip(members, pks):
                matrices[j] = big[r0:r0 + pk.n_series]
                r0 += pk.n_series

        for j, (pk, fplan, nplan) in enumerate(jobs):
            if j in matrices:
                matrix = matrices[j]
            elif not fplan.names:
                matrix = np.empty((pk.n_series, 0))
            elif multi:
                from tsfresh_amd.distributed import extract_on_devices
                specs = list(fplan.native_specs(_native.calc_id))
                parts = _split_native_specs(specs)   # (one part unless the settings need several native plans)
                # (several native plans: every device uploads its shard ONCE and runs all the parts on it)
                matrix = extract_on_devices(specs, pk.values, pk.offsets, devices, times=pk.times,
                                            parts=parts if len(parts) > 1 else None)
            else:
                matrix = nplan.extract_host(pk.values, pk.offsets, times=pk.times)
            # the reference's exceptions that depend on the samples (an infinite value under binned_entropy / ar_coefficient)
            check_reference_data_errors(fplan.specs, matrix, pk.values, pk.offsets[:-1], pk.offsets[1:])
            # user-defined calculators (callable keys): per series on the host, spliced in at their dict position
            names, matrix = fplan.finish(matrix, lambda i, pk=pk: pk.values[pk.offsets[i]:pk.offsets[i + 1]], pk.n_series)
            blocks.append((pk, [pk.kind + "__" + name for name in names], matrix))
        _trim_cache(_thread_cache())

    return _assemble(blocks, id_dtype, pivot, impute_function)


def _assemble(blocks, id_dtype, pivot, impute_function):
    """(PackedKind-like with .ids, column names, matrix) blocks -> the reference's result container
    (data.py:86-121 pivot / extraction.py:301-302 tuples)."""
    if not pivot:
        result = []
        for pk, names, matrix in blocks:
            for r, sample_id in enumerate(pk.ids):
                row = matrix[r]
                result.extend((sample_id, name, row[c]) for c, name in enumerate(names))
        return result

    # assemble: union of ids (sorted), one column block per kind
    if not blocks:
        return pd.DataFrame()
    all_ids = blocks[0][0].ids
    same_ids = all(len(b[0].ids) == len(all_ids) and np.array_equal(b[0].ids, all_ids) for b in blocks[1:])
    if same_ids:
        index = pd.Index(all_ids)
        data = np.concatenate([b[2] for b in blocks], axis=1) if len(blocks) > 1 else blocks[0][2]
    else:
        index = pd.Index(all_ids)
        for b in blocks[1:]:
            index = index.union(pd.Index(b[0].ids))
        index = index.sort_values()
        n_cols = sum(b[2].shape[1] for b in blocks)
        data = np.full((len(index), n_cols), np.nan)
        c0 = 0
        for pk, names, matrix in blocks:
            rows = index.get_indexer(pd.Index(pk.ids))
            data[rows, c0:c0 + matrix.shape[1]] = matrix
            c0 += matrix.shape[1]
    columns = [name for b in blocks for name in b[1]]
    result = pd.DataFrame(data, index=index, columns=columns, dtype=float)
    try:
        result.index = result.index.astype(id_dtype)  # data.py:115-116
    except (TypeError, ValueError):
        pass
    if not result.index.is_monotonic_increasing:  # ids come out of the packer sorted: no 125 MB copy per 20 k rows
        result = result.sort_index()
    if impute_function is not None:
        impute_function(result)
    return result


class _WindowBlock:
    def __init__(self, kind, ids):
        self.kind = kind
        self.ids = ids


def extract_rolled_features(timeseries_container, column_id=None, column_sort=None, column_kind=None, column_value=None,
                            rolling_direction=1, max_timeshift=None, min_timeshift=0, default_fc_parameters=None,
                            kind_to_fc_parameters=None, impute_function=None, show_warnings=False, pivot=True,
                            device=None):
    """`extract_features(roll_time_series(container, ...), column_id="id", ...)` without building the rolled frame.

    The reference's forecasting workflow first copies every window into a new DataFrame
    (tsfresh/utilities/dataframe_functions.py:340-372, :601) and then extracts features from it.  Here every series is
    packed and uploaded ONCE; the windows are `(start, end)` views into that buffer
    (`tsfresh_amd.utilities.dataframe_functions.roll_views` restates the reference's index arithmetic) and go to the
    kernels through `tsfa_extract_windows`.  Same rows, same ``(id, shift)`` index, same columns as the two-step form.
    """
    from tsfresh_amd.utilities.dataframe_functions import roll_views
    if default_fc_parameters is None and kind_to_fc_parameters is None:
        default_fc_parameters = ComprehensiveFCParameters()
    elif default_fc_parameters is None:
        default_fc_parameters = {}
    if isinstance(timeseries_container, pd.DataFrame) and len(timeseries_container) <= 1:
        raise ValueError("Your time series container has zero or one rows!. Can not perform rolling.")
    packed, id_dtype, _ = pack_timeseries(timeseries_container, column_id=column_id, column_kind=column_kind,
                                          column_value=column_value, column_sort=column_sort)
    if device is None:
        device = _default_device()
    # prediction_steps is the longest series over ALL ids and kinds of ONE frame (dataframe_functions.py:546); a dict
    # container is rolled entry by entry (:430-445), each entry with the longest series of its own frame
    per_kind_steps = isinstance(timeseries_container, dict)
    steps_all = max(int(np.diff(pk.offsets).max()) for pk in packed if pk.n_series)
    blocks, plan_cache, pins = [], {}, set()
    with warnings.catch_warnings():
        warnings.simplefilter("default" if show_warnings else "ignore")
        for pk in packed:
            fc_parameters = kind_to_fc_parameters[pk.kind] if kind_to_fc_parameters and pk.kind in kind_to_fc_parameters \
                else default_fc_parameters
            kind_has_dt = pk.times is not None
            key = (id(fc_parameters), kind_has_dt)
            if key not in plan_cache:
                fplan = compile_fc_parameters(fc_parameters, has_datetime_index=kind_has_dt)
                if fplan.host_calls:
                    from tsfresh_amd.feature_extraction.registry import UnsupportedFeature
                    raise UnsupportedFeature("custom (callable) calculators are evaluated per series on the host: roll the "
                                             "frame with roll_time_series and call extract_features on it")
                plan_cache[key] = (fplan, _acquire_plan(fplan, device, pins) if len(fplan) else None)
            fplan, nplan = plan_cache[key]
            if nplan is None:
                continue
            lengths = np.diff(pk.offsets)
            steps = int(lengths.max()) if per_kind_steps else steps_all
            # roll_views sizes its shifts from the longest series it is given: append a phantom of `steps` samples
            gi, frm, until, ts = roll_views(np.concatenate([lengths, [steps]]), rolling_direction, max_timeshift, min_timeshift)
            keep = gi < len(lengths)
            gi, frm, until, ts = gi[keep], frm[keep], until[keep], ts[keep]
            starts, ends = pk.offsets[gi] + frm, pk.offsets[gi] + until
            if pk.sort is not None:
                shift_val = pk.sort[ends - 1] if rolling_direction > 0 else pk.sort[starts]
                if shift_val.dtype.kind in "mM":  # as roll_time_series: pandas Timestamps in the window ids
                    shift_val = pd.Series(shift_val).tolist()
            else:
                shift_val = ts - 1
            ids = np.empty(len(gi), dtype=object)
            base_ids = pk.ids[gi]
            for i in range(len(gi)):
                ids[i] = (base_ids[i], shift_val[i])
            matrix = nplan.extract_windows_host(pk.values, starts, ends, times=pk.times)
            check_reference_data_errors(fplan.specs, matrix, pk.values, starts, ends)
            order = sorted(range(len(ids)), key=lambda i: ids[i])
            blocks.append((_WindowBlock(pk.kind, ids[order]), [pk.kind + "__" + n for n in fplan.names], matrix[order]))
        _trim_cache(_thread_cache())
    return _assemble(blocks, np.dtype(object), pivot, impute_function)
