"""ORACLE (test infrastructure only): the per-series dispatcher of the hot path.

Restates tsfresh/feature_extraction/extraction.py:308-386 (`_do_extraction_on_chunk`): walk the FCParameters dict,
call simple calculators once per parameter dict and combiners once with the whole list, and name every value
``"{kind}__{calculator}[__{parameters}]"`` (parameters formatted as utilities/string_manipulation.py:47-74 does).
"""
import warnings

import numpy as np

from oracle.calculators import COMBINERS, SeriesOracle


def _param_suffix(param):
    # utilities/string_manipulation.py:47 convert_to_output_format
    return "__".join(
        str(k) + "_" + ('"' + str(param[k]) + '"' if isinstance(param[k], str) else str(param[k]))
        for k in sorted(param.keys()))


def oracle_series(x, fc_parameters, kind="value", times=None):
    """-> list[(column name, float value)] for one series, in the reference's emission order.
    times: hours since the series' first timestamp when the data has a DatetimeIndex, else None."""
    so = SeriesOracle(x, times)
    skip = () if times is not None else ("linear_trend_timewise",)
    out = []
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for name, param_list in fc_parameters.items():
            if name in skip:  # needs a DatetimeIndex; the reference skips it with a warning otherwise
                continue
            func = getattr(so, name)
            if name in COMBINERS:
                items = func(param_list)
            elif param_list:
                items = [(_param_suffix(p), func(**p)) for p in param_list]
            else:
                items = [("", func())]
            for key, value in items:
                col = str(kind) + "__" + name + ("__" + str(key) if key else "")
                out.append((col, float(value)))
    return out


def oracle_matrix(values, offsets, fc_parameters, kind="value", times=None):
    """-> (column names, float64 matrix [n_series, n_cols]) for a ragged batch."""
    names, rows = None, []
    for s in range(len(offsets) - 1):
        items = oracle_series(values[offsets[s]:offsets[s + 1]], fc_parameters, kind=kind,
                              times=None if times is None else times[offsets[s]:offsets[s + 1]])
        cols, seen, vals = [], {}, []
        for c, v in items:
            if c in seen:
                vals[seen[c]] = v
            else:
                seen[c] = len(cols)
                cols.append(c)
                vals.append(v)
        if names is None:
            names = cols
        rows.append(vals)
    return names, np.asarray(rows, dtype=np.float64)
