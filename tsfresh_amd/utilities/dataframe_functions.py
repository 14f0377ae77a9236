"""Rolling ("forecasting") windows.

`roll_time_series` keeps the reference's contract (tsfresh/utilities/dataframe_functions.py:377-603): every id is cut
into sub-windows, each of which becomes a new time series with the id ``(id, shift)``.  The reference builds the
result by copying every window into a new DataFrame (`_roll_out_time_series` :294-372, one `groupby.apply` per
shift); here the windows are first described as `(series, first row, end row)` views (`roll_views`), and

  * `roll_time_series` materialises them with one vectorised gather -- same rows, same order, same ``id`` tuples;
  * `tsfresh_amd.extract_rolled_features` hands the views straight to the GPU (`tsfa_extract_windows`): the samples
    are uploaded once and every window is a `(start, end)` pair into that one buffer, which removes the O(windows x
    length) host memory of the forecasting workflow (BASELINE.json configs[4]).
"""
import warnings

import numpy as np
import pandas as pd


def roll_views(lengths, rolling_direction=1, max_timeshift=None, min_timeshift=0):
    """Window views of series with the given lengths.

    Restates the index arithmetic of `_roll_out_time_series` (dataframe_functions.py:340-358) for all shifts at once.
    -> (series index, first row, end row (exclusive), timeshift) as int64 arrays, ordered by (series, timeshift).
    """
    if rolling_direction == 0:
        raise ValueError("Rolling direction of 0 is not possible")
    if max_timeshift is not None and max_timeshift <= 0:
        raise ValueError("max_timeshift needs to be positive!")
    if min_timeshift < 0:
        raise ValueError("min_timeshift needs to be positive or zero!")
    lengths = np.asarray(lengths, dtype=np.int64)
    if lengths.size == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, z, z, z
    amount = abs(int(rolling_direction))
    steps = int(lengths.max())                       # prediction_steps (:546)
    mts = int(max_timeshift or steps)                # :548
    if rolling_direction > 0:
        shifts = np.arange(steps, 0, -amount, dtype=np.int64)[::-1]   # :550-551
    else:
        shifts = np.arange(1, steps + 1, amount, dtype=np.int64)      # :553
    sidx = np.repeat(np.arange(lengths.size, dtype=np.int64), shifts.size)
    ts = np.tile(shifts, lengths.size)
    ln = lengths[sidx]
    if rolling_direction > 0:
        until = ts                                   # :343
        frm = np.maximum(until - mts - 1, 0)         # :344
        ok = until <= ln                             # :346
    else:
        frm = np.maximum(ts - 1, 0)                  # :349
        until = np.minimum(frm + mts + 1, ln)        # :350-352 (iloc clips)
        ok = frm < ln
    ok &= (until - frm) >= (min_timeshift + 1)       # :354
    return sidx[ok], frm[ok], until[ok], ts[ok]


def _sorted_groups(df, column_id, column_sort, column_kind):
    """Row order of `df.sort_values(column_sort)` grouped by (kind, id) -> (row order, group starts, group lengths,
    group keys as a list of arrays)."""
    n = len(df)
    order = np.arange(n)
    if column_sort is not None:
        order = np.argsort(df[column_sort].to_numpy(), kind="stable")
    keys = []
    if column_kind is not None:
        keys.append(pd.factorize(df[column_kind].to_numpy()[order], sort=True)[0])
    keys.append(pd.factorize(df[column_id].to_numpy()[order], sort=True)[0])
    grp = np.lexsort(tuple(reversed(keys)))          # stable: keeps the sort order inside a group
    order = order[grp]
    code = np.zeros(n, dtype=np.int64)
    for k in keys:
        code = code * (int(k.max()) + 1 if n else 1) + k[grp]
    change = np.nonzero(np.diff(code))[0] + 1
    starts = np.concatenate([[0], change]).astype(np.int64)
    lengths = np.diff(np.concatenate([starts, [n]])).astype(np.int64)
    return order, starts, lengths


def roll_time_series(df_or_dict, column_id, column_sort=None, column_kind=None, rolling_direction=1,
                     max_timeshift=None, min_timeshift=0, chunksize=None, n_jobs=None, show_warnings=False,
                     disable_progressbar=True, distributor=None):
    """Reference-compatible `roll_time_series` (dataframe_functions.py:377): returns the rolled DataFrame (or dict of
    DataFrames) with the new ``id`` column of ``(id, shift)`` tuples, sorted by ``["id", column_sort or "sort"]``.
    `chunksize`, `n_jobs`, `disable_progressbar`, `distributor` only steer the reference's CPU distributors and are
    ignored."""
    if rolling_direction == 0:
        raise ValueError("Rolling direction of 0 is not possible")
    if max_timeshift is not None and max_timeshift <= 0:
        raise ValueError("max_timeshift needs to be positive!")
    if min_timeshift < 0:
        raise ValueError("min_timeshift needs to be positive or zero!")
    if isinstance(df_or_dict, dict):
        if column_kind is not None:
            raise ValueError("You passed in a dictionary and gave a column name for the kind. Both are not possible.")
        return {key: roll_time_series(df_or_dict[key], column_id=column_id, column_sort=column_sort,
                                      column_kind=column_kind, rolling_direction=rolling_direction,
                                      max_timeshift=max_timeshift, min_timeshift=min_timeshift)
                for key in df_or_dict}
    df = df_or_dict
    if len(df) <= 1:
        raise ValueError("Your time series container has zero or one rows!. Can not perform rolling.")
    if column_id is None:
        raise ValueError("You have to set the column_id which contains the ids of the different time series")
    if column_id not in df:
        raise AttributeError("The given column for the id is not present in the data.")
    if column_sort is not None:
        if df[column_sort].isnull().any():
            raise ValueError("You have NaN values in your sort column.")
    else:
        df = df.copy()
        df["sort"] = range(df.shape[0])             # :543
    order, starts, lengths = _sorted_groups(df, column_id, column_sort, column_kind)
    if column_sort is not None and df[column_sort].dtype != object:
        sv = df[column_sort].to_numpy()[order]
        d = sv[:-1] - sv[1:]
        inner = np.ones(len(sv) - 1, dtype=bool)
        inner[starts[1:] - 1] = False               # differences across group borders do not count
        if inner.any() and d[inner].min() != d[inner].max():
            warnings.warn("Your time stamps are not uniformly sampled, which makes rolling "
                          "nonsensical in some domains.")
    gi, frm, until, ts = roll_views(lengths, rolling_direction, max_timeshift, min_timeshift)
    wlen = until - frm
    # rows of all windows: group start + first row + 0..len-1
    first = starts[gi] + frm
    rows = np.repeat(first, wlen) + (np.arange(int(wlen.sum())) - np.repeat(np.cumsum(wlen) - wlen, wlen))
    out = df.iloc[order[rows]].copy()
    sort_col = column_sort or "sort"
    if column_sort is not None:
        sv = df[column_sort].to_numpy()[order]
        shift_val = sv[first + wlen - 1] if rolling_direction > 0 else sv[first]   # :363-366
        if shift_val.dtype.kind in "mM":             # pandas hands out Timestamps / Timedeltas, not numpy scalars
            shift_val = pd.Series(shift_val).tolist()
    else:
        shift_val = ts - 1                           # :368
    ids = df[column_id].to_numpy()[order][first]
    if ids.dtype.kind in "mM":
        ids = pd.Series(ids).tolist()
    new_ids = np.empty(len(first), dtype=object)
    for i in range(len(first)):
        new_ids[i] = (ids[i], shift_val[i])
    out["id"] = np.repeat(new_ids, wlen)            # :370
    # the reference concatenates shift by shift (pd.concat(..., ignore_index=True) :601): give every row the position
    # it has there, so that even the index of the result is the reference's
    by_shift = np.lexsort((gi, ts))
    base = np.empty(len(first), dtype=np.int64)
    base[by_shift] = np.cumsum(wlen[by_shift]) - wlen[by_shift]
    out.index = np.repeat(base, wlen) + (np.arange(int(wlen.sum())) - np.repeat(np.cumsum(wlen) - wlen, wlen))
    return out.sort_values(by=["id", sort_col])


# ---------------------------------------------------------------------------------------------------------------
# Imputation of the feature matrix (reference: dataframe_functions.py:49-214).  Same names, same
# in-place contract, same warnings and errors; the replacement itself is three vectorised numpy passes over the
# float64 block instead of three row-replicated DataFrames and `DataFrame.where`.
# ---------------------------------------------------------------------------------------------------------------
def check_for_nans_in_columns(df, columns=None):
    """Raise the reference's ValueError if `df[columns]` holds a NaN (dataframe_functions.py:20)."""
    if columns is None:
        columns = df.columns
    sub = df.loc[:, columns]
    if pd.isnull(sub).any().any():
        raise ValueError("Columns {} of DataFrame must not contain NaN values".format(
            sub.columns[pd.isnull(sub).sum() > 0].tolist()))


def get_range_values_per_column(df):
    """-> (col_to_max, col_to_min, col_to_median) over the finite values of every column; 0 for a column without
    any finite value, with the reference's RuntimeWarning (dataframe_functions.py:176)."""
    data = np.asarray(df.values, dtype=np.float64)
    finite = np.isfinite(data)
    dead = ~finite.any(axis=0) if data.shape[0] else np.zeros(data.shape[1], dtype=bool)
    if np.any(dead):
        warnings.warn("The columns {} did not have any finite values. Filling with zeros.".format(
            df.iloc[:, np.where(dead)[0]].columns.values), RuntimeWarning)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mx = np.where(dead, 0.0, np.max(np.where(finite, data, -np.inf), axis=0))
        mn = np.where(dead, 0.0, np.min(np.where(finite, data, np.inf), axis=0))
        med = np.where(dead, 0.0, np.nanmedian(np.where(finite, data, np.nan), axis=0))
    cols = df.columns
    return dict(zip(cols, mx)), dict(zip(cols, mn)), dict(zip(cols, med))


def impute_dataframe_range(df_impute, col_to_max, col_to_min, col_to_median):
    """+inf -> col_to_max, -inf -> col_to_min, NaN -> col_to_median, in place (dataframe_functions.py:113)."""
    if len(df_impute) == 0:
        return df_impute
    columns = df_impute.columns
    if (not set(columns) <= set(col_to_median.keys()) or not set(columns) <= set(col_to_max.keys())
            or not set(columns) <= set(col_to_min.keys())):
        raise ValueError("Some of the dictionaries col_to_median, col_to_max, col_to_min contains more or less keys "
                         "than the column names in df")
    if (np.any(~np.isfinite(list(col_to_median.values()))) or np.any(~np.isfinite(list(col_to_min.values())))
            or np.any(~np.isfinite(list(col_to_max.values())))):
        raise ValueError("Some of the dictionaries col_to_median, col_to_max, col_to_min contains non finite values "
                         "to replace")
    data = np.array(df_impute.values, dtype=np.float64)
    mx = np.array([col_to_max[c] for c in columns], dtype=np.float64)
    mn = np.array([col_to_min[c] for c in columns], dtype=np.float64)
    med = np.array([col_to_median[c] for c in columns], dtype=np.float64)
    data = np.where(data == np.inf, mx, data)
    data = np.where(data == -np.inf, mn, data)
    data = np.where(np.isnan(data), med, data)
    df_impute[:] = data
    return df_impute


def impute_dataframe_zero(df_impute):
    """NaN, -inf, +inf -> 0, in place (dataframe_functions.py:91)."""
    if len(df_impute) == 0:
        return df_impute
    data = np.array(df_impute.values, dtype=np.float64)
    data[~np.isfinite(data)] = 0.0
    df_impute[:] = data
    return df_impute


_DEVICE_IMPUTE_MIN_CELLS = 1 << 16


def impute(df_impute, device=None):
    """-inf -> column min, +inf -> column max, NaN -> column median (over the finite values), in place; a column
    without finite values becomes 0 (dataframe_functions.py:49).  The usual `impute_function` of extract_features.

    A float64 frame of at least 65 536 cells whose values are one contiguous block (what extract_features returns) is
    imputed on the GPU: `tsfa_impute` sorts every column once in HBM (maximum / minimum / median of the finite values)
    and patches the cells in place -- the reference's np.ma passes (dataframe_functions.py:142-180) take seconds on a
    100 000 x 783 matrix.  Smaller or mixed-dtype frames use the numpy restatement below."""
    if len(df_impute) == 0:
        return df_impute
    if df_impute.shape[0] * df_impute.shape[1] >= _DEVICE_IMPUTE_MIN_CELLS and all(d == np.float64 for d in df_impute.dtypes):
        from tsfresh_amd import _native
        vals = df_impute.values
        if (_native.device_count() > 0 and vals.dtype == np.float64 and vals.flags.c_contiguous and vals.flags.writeable
                and np.shares_memory(vals, df_impute.values)):
            from tsfresh_amd.feature_extraction.extraction import _default_device
            _, _, _, cnt = _native.impute_matrix(vals, device=_default_device() if device is None else device)
            if (cnt == 0).any():  # the reference's warning for columns without a finite value (:162-170)
                warnings.warn("The columns {} did not have any finite values. Filling with zeros.".format(
                    df_impute.iloc[:, np.where(cnt == 0)[0]].columns.values), RuntimeWarning)
            return df_impute
    col_to_max, col_to_min, col_to_median = get_range_values_per_column(df_impute)
    return impute_dataframe_range(df_impute, col_to_max, col_to_min, col_to_median)


def make_forecasting_frame(x, kind, max_timeshift, rolling_direction, min_timeshift=0):
    """The reference's forecasting container (tsfresh/utilities/dataframe_functions.py:606): every time stamp of the
    single series `x` gets the window of its last `max_timeshift` predecessors as a series of its own (id = ("id",
    time stamp)), and `y` holds the value to predict for each of them.  Built on `roll_time_series` of this package;
    for extraction the windows need not be materialised at all -- see `extract_rolled_features`."""
    n = len(x)
    t = x.index if isinstance(x, pd.Series) else range(n)
    df = pd.DataFrame({"id": ["id"] * n, "time": t, "value": x, "kind": kind})
    df_shift = roll_time_series(df, column_id="id", column_sort="time", column_kind="kind",
                                rolling_direction=rolling_direction, max_timeshift=max_timeshift,
                                min_timeshift=min_timeshift)
    # drop, in every window, the row that is to be predicted (its last one)
    last_of_window = ~df_shift.duplicated(subset=["id"], keep="last") if len(df_shift) else pd.Series([], dtype=bool)
    df_shift = df_shift[~last_of_window]
    # targets: every value but the first, named like the windows, restricted to the windows that exist
    y = df["value"][1:]
    y.index = map(lambda i: ("id", i), y.index)
    valid_ids = set(df_shift["id"].unique())
    y = y[y.index.isin(valid_ids)]
    return df_shift, y
