"""Input adapters: pandas containers -> ragged buffers (one per kind) for the C-ABI.

The reference wraps its three input formats in iterables of single `pd.Series`
(tsfresh/feature_extraction/data.py: WideTsFrameAdapter :181, LongTsFrameAdapter :233, TsDictAdapter :294,
to_tsdata :447) and then walks them one series at a time.  Here the same formats, checks and error messages
produce, per kind, ONE contiguous value buffer plus an offsets array: the `[n_ids x n_kinds x max_len]` ragged
layout the kernels consume.  Everything is vectorised (factorize + lexsort); no per-series Python.
"""
import numpy as np
import pandas as pd

from tsfresh_amd import _native

_NATIVE_SCAN_MIN_ROWS = 1 << 18  # below this the numpy passes are as fast as spawning the scan threads


class PackedKind:
    """All series of one kind: `values[offsets[i]:offsets[i+1]]` is the series of `ids[i]` (ids sorted)."""

    def __init__(self, kind, ids, values, offsets, times=None, sort=None):
        self.kind = kind
        self.ids = ids
        self.values = values
        self.offsets = offsets
        self.times = times  # float64 hours since each series' first timestamp (DatetimeIndex input only)
        self.sort = sort    # the sort column in packed order (None without column_sort); names rolled windows

    @property
    def n_series(self):
        return len(self.offsets) - 1


def _check_colname(*columns):
    # data.py:124-145
    for col in columns:
        if str(col).endswith("_"):
            raise ValueError("Dict keys are not allowed to end with '_': {}".format(col))
        if "__" in str(col):
            raise ValueError("Dict keys are not allowed to contain '__': {}".format(col))


def _check_nan(df, *columns, defer=()):
    # data.py:148-167.  Integer / boolean columns cannot hold NaN (no 20 M-row isnull() pass for an int64 id column);
    # columns in `defer` are checked by _pack (in the same native pass that finds the group boundaries).
    for col in columns:
        if col not in df.columns:
            raise ValueError("Column not found: {}".format(col))
        if col in defer or df[col].dtype.kind in "iub":
            continue
        if df[col].isnull().any():
            raise ValueError("Column must not contain NaN values: {}".format(col))


def _raise_if_nan(values, name):
    if name is not None and values.dtype.kind == "f" and bool(np.isnan(values).any()):
        raise ValueError("Column must not contain NaN values: {}".format(name))


def _get_value_columns(df, *other_columns):
    # data.py:170-178
    value_columns = [col for col in df.columns if col not in other_columns]
    if len(value_columns) == 0:
        raise ValueError("Could not guess the value column! Please hand it to the function as an argument.")
    return value_columns


def _as_values(column):
    arr = np.asarray(column)
    if arr.dtype == np.float32 or arr.dtype == np.float64:
        return arr   # read only from here on: no copy
    return arr.astype(np.float64)


def _hours_since_first(index, order, offsets):
    """Per sample: hours since the first timestamp of its series, with the arithmetic of the reference's
    linear_trend_timewise (feature_calculators.py:2291-2296): (ix - ix[0]).total_seconds() / 3600.0."""
    ix = index[order]
    counts = np.diff(offsets)
    first = ix[np.repeat(offsets[:-1], counts)]
    return np.ascontiguousarray(np.asarray((ix - first).total_seconds() / float(3600)), dtype=np.float64)


def _pack_presorted(kind, ids, values, sort_values, index, nan_name=None):
    """The usual layout of a long frame -- numeric ids already non-decreasing, every group already in sort order --
    needs no hashing and no permutation: the group boundaries are where the id changes (three linear passes over the
    rows instead of `factorize` + `bincount` + the element-wise sortedness test: 4x less packing time on 20 M rows).
    Returns None when the layout is anything else (the general path then sorts)."""
    if ids.dtype.kind not in "iuf" or len(ids) < 2:
        return None
    if len(ids) >= _NATIVE_SCAN_MIN_ROWS:
        # one multi-threaded native pass: layout proof + group boundaries + the NaN check of the value column
        # (a float32 / float64 column as it is; float16 / longdouble -- which CAN hold a NaN -- converted first, so that
        # the scan sees them; integer and bool columns cannot hold one and are only converted -- a full copy -- once the
        # layout is proven, so an unsorted integer frame is not converted twice)
        raw = np.asarray(values)
        if raw.dtype.kind == "f" and raw.dtype.itemsize not in (4, 8):
            raw = _as_values(raw)
        scanned = raw if raw.dtype.kind == "f" and raw.dtype.itemsize in (4, 8) else None
        sv = None if sort_values is None else np.asarray(sort_values)
        res = _native.pack_scan(ids, sv, scanned)
        if res is not None:
            flags, offsets = res
            if flags & _native.TSFA_PACK_VALUE_NAN and nan_name is not None:
                raise ValueError("Column must not contain NaN values: {}".format(nan_name))
            if offsets is None:
                return None
            vals = scanned if scanned is not None else _as_values(raw)
            times = _hours_since_first(index, slice(None), offsets) if index is not None else None
            return PackedKind(str(kind), ids[offsets[:-1]], np.ascontiguousarray(vals), offsets, times, sv)
    if not bool(np.all(ids[1:] >= ids[:-1])):  # also False for NaN ids
        return None
    cuts = np.flatnonzero(ids[1:] != ids[:-1]) + 1
    sv = None
    if sort_values is not None:
        sv = np.asarray(sort_values)
        try:
            bad = np.flatnonzero(sv[1:] < sv[:-1]) + 1  # descents are only allowed where a new id starts
        except TypeError:
            return None
        if len(bad) > len(cuts) or not bool(np.all(np.isin(bad, cuts, assume_unique=True))):
            return None
    offsets = np.empty(len(cuts) + 2, dtype=np.int64)
    offsets[0] = 0
    offsets[1:-1] = cuts
    offsets[-1] = len(ids)
    uniques = ids[offsets[:-1]]
    times = _hours_since_first(index, slice(None), offsets) if index is not None else None
    vals = np.ascontiguousarray(_as_values(values))
    _raise_if_nan(vals, nan_name)
    return PackedKind(str(kind), uniques, vals, offsets, times, sv)


def _pack(kind, ids, values, sort_values, index=None, nan_name=None):
    """Group `values` by `ids` (ascending), each group ordered by `sort_values` (stable).  `index`: the frame's
    DatetimeIndex (row-aligned with `values`) or None.  nan_name: the value column's name if its NaN check
    (data.py:148-167) has been left to this function."""
    ids = np.asarray(ids)
    fast = _pack_presorted(kind, ids, values, sort_values, index, nan_name)
    if fast is not None:
        return fast
    _raise_if_nan(_as_values(values), nan_name)
    codes, uniques = pd.factorize(ids, sort=True)
    order = None
    if len(codes) > 1 and np.all(codes[1:] >= codes[:-1]):
        # already grouped by id in ascending order (the usual layout of a long frame): if every group is also in sort
        # order the permutation is the identity, and the 20M-row lexsort (90 % of the packing time) is skipped
        if sort_values is None:
            order = slice(None)
        else:
            try:
                sv = np.asarray(sort_values)
                if np.all((sv[1:] >= sv[:-1]) | (codes[1:] != codes[:-1])):
                    order = slice(None)
            except TypeError:  # sort values that do not compare element-wise
                order = None
    if order is None:
        if sort_values is not None:
            order = np.lexsort((np.asarray(sort_values), codes))
        else:
            order = np.argsort(codes, kind="stable")
    counts = np.bincount(codes, minlength=len(uniques))
    offsets = np.zeros(len(uniques) + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    times = _hours_since_first(index, order, offsets) if index is not None and len(codes) else None
    return PackedKind(str(kind), np.asarray(uniques), np.ascontiguousarray(_as_values(values)[order]), offsets, times,
                      None if sort_values is None else np.asarray(sort_values)[order])


def _arrow_to_frame(table):
    """A pyarrow Table / RecordBatch as a DataFrame, column by column through numpy (no Python objects for primitive
    columns; pandas still consolidates the columns into its own blocks): the checks and the packing below then see an
    ordinary long / wide frame."""
    cols = {}
    for name in table.schema.names:
        col = table.column(name)
        if hasattr(col, "combine_chunks"):
            col = col.combine_chunks()
        try:
            cols[name] = col.to_numpy(zero_copy_only=True)
        except Exception:  # strings, nulls, chunk boundaries: Arrow has to materialise a copy
            cols[name] = col.to_numpy(zero_copy_only=False)
    return pd.DataFrame(cols, copy=False)


def _pack_arrow_wide(table, column_id, column_kind, column_value, column_sort):
    """Wide-format pyarrow Table / RecordBatch with primitive, null-free columns: the Arrow buffers go to the packer as
    numpy views (zero copy) -- no pandas frame, no block consolidation.  None -> the caller converts to a DataFrame
    and takes the general route (strings, nulls, chunked columns that need a copy, the long format)."""
    if column_id is None or column_kind is not None:
        return None
    names = list(table.schema.names)
    if column_id not in names or (column_sort is not None and column_sort not in names):
        return None
    value_columns = [column_value] if column_value is not None else [c for c in names if c not in (column_id, column_sort)]
    if not value_columns or any(c not in names for c in value_columns):
        return None
    arrays = {}
    for name in [column_id] + ([column_sort] if column_sort is not None else []) + value_columns:
        col = table.column(name)
        if getattr(col, "null_count", 0):
            return None
        if hasattr(col, "num_chunks"):
            if col.num_chunks != 1:
                return None
            col = col.chunk(0)
        try:
            arrays[name] = col.to_numpy(zero_copy_only=True)
        except Exception:
            return None
        if arrays[name].dtype.kind not in "iuf":
            return None
    _check_colname(*value_columns)
    for name in [column_id] + ([column_sort] if column_sort is not None else []):
        _raise_if_nan(arrays[name], name)
    ids = arrays[column_id]
    sort_all = arrays[column_sort] if column_sort is not None else None
    packed = [_pack(c, ids, arrays[c], sort_all, None, nan_name=c) for c in value_columns]
    return packed, ids.dtype, False


def pack_timeseries(container, column_id=None, column_kind=None, column_value=None, column_sort=None):
    """-> (list[PackedKind] in output-column order, dtype of the id column, has_datetime_index)."""
    if type(container).__module__.startswith("pyarrow") and hasattr(container, "schema"):
        direct = _pack_arrow_wide(container, column_id, column_kind, column_value, column_sort)
        if direct is not None:
            return direct
        container = _arrow_to_frame(container)
    if isinstance(container, pd.DataFrame):
        df = container
        if column_id is None:
            raise ValueError("A value for column_id needs to be supplied")
        if column_kind is not None:
            # long format (data.py:233-291)
            if column_value is None:
                possible = _get_value_columns(df, column_id, column_sort, column_kind)
                if len(possible) != 1:
                    raise ValueError(
                        "Could not guess the value column, as the number of unused columns os not equal to 1."
                        "These columns where currently unused: {}"
                        "Please hand it to the function as an argument.".format(",".join(map(str, possible))))
                column_value = possible[0]
            _check_nan(df, column_id, column_kind, column_value)
            if column_sort is not None:
                _check_nan(df, column_sort)
            kinds = df[column_kind].to_numpy()
            dt_index = df.index if isinstance(df.index, pd.DatetimeIndex) else None
            kcodes, kuniq = pd.factorize(kinds, sort=True)
            packed = []
            ids_all = df[column_id].to_numpy()
            vals_all = df[column_value].to_numpy()
            sort_all = df[column_sort].to_numpy() if column_sort is not None else None
            for k, kind in enumerate(kuniq):
                sel = np.nonzero(kcodes == k)[0]
                packed.append(_pack(kind, ids_all[sel], vals_all[sel], None if sort_all is None else sort_all[sel],
                                    None if dt_index is None else dt_index[sel]))
            return packed, df[column_id].dtype, isinstance(df.index, pd.DatetimeIndex)
        # wide format (data.py:181-230)
        _check_nan(df, column_id)
        value_columns = [column_value] if column_value is not None else _get_value_columns(df, column_id, column_sort)
        deferred = [c for c in value_columns if c in df.columns and df[c].dtype.kind == "f"]
        _check_nan(df, *value_columns, defer=deferred)
        _check_colname(*value_columns)
        if column_sort is not None:
            _check_nan(df, column_sort)
        ids_all = df[column_id].to_numpy()
        sort_all = df[column_sort].to_numpy() if column_sort is not None else None
        dt_index = df.index if isinstance(df.index, pd.DatetimeIndex) else None
        packed = [_pack(col, ids_all, df[col].to_numpy(), sort_all, dt_index, nan_name=col if col in deferred else None)
                  for col in value_columns]
        return packed, df[column_id].dtype, isinstance(df.index, pd.DatetimeIndex)
    if isinstance(container, dict):
        # dict of frames, one per kind (data.py:294-338)
        _check_colname(*list(container.keys()))
        for frame in container.values():
            _check_nan(frame, column_id, column_value)
        if column_sort is not None:
            for frame in container.values():
                _check_nan(frame, column_sort)
        packed, id_dtype, has_dt = [], None, False
        for kind, frame in container.items():
            sort_vals = frame[column_sort].to_numpy() if column_sort is not None else None
            packed.append(_pack(kind, frame[column_id].to_numpy(), frame[column_value].to_numpy(), sort_vals,
                                frame.index if isinstance(frame.index, pd.DatetimeIndex) else None))
            id_dtype = frame[column_id].dtype
            has_dt = has_dt or isinstance(frame.index, pd.DatetimeIndex)
        return packed, id_dtype, has_dt
    raise ValueError("df must be a DataFrame or a dict of DataFrames. "
                     "See https://tsfresh.readthedocs.io/en/latest/text/data_formats.html")
