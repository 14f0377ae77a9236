"""The reference's second plug point: a distributor object.

`tsfresh.extract_features(..., distributor=GPUDistributor())` makes the REFERENCE's own `_do_extraction`
(tsfresh/feature_extraction/extraction.py:193-305) hand its `TsData` iterable to the GPU instead of mapping
`_do_extraction_on_chunk` over a process pool: `map_reduce` (utilities/distribution.py:74-104) receives the iterable
of `Timeseries(id, kind, data)` chunks plus the FCParameters in `function_kwargs`, and has to return the flat list of
`(id, "kind__calculator__params", value)` tuples that `data.pivot` (data.py:86) expects.

When tsfresh is importable the class derives from its `DistributorBaseClass` (the reference checks `isinstance`,
extraction.py:285); here, where it is not, it is a plain class with the same methods.  `tsfresh_amd.extract_features`
accepts an instance too and only takes the device from it.
"""
import warnings

import numpy as np
import pandas as pd

try:  # pragma: no cover - tsfresh is not installed in the build image
    from tsfresh.utilities.distribution import DistributorBaseClass as _Base
except Exception:  # ImportError or a broken optional dependency of tsfresh
    _Base = object


def is_distributor(obj):
    """What extraction.py:285 asks: an instance of the reference's DistributorBaseClass where tsfresh is importable; where it is
    not, any object with the class's interface (`map_reduce` + `close`, utilities/distribution.py:64-104)."""
    if _Base is not object and isinstance(obj, _Base):
        return True
    return isinstance(obj, GPUDistributor) or (callable(getattr(obj, "map_reduce", None)) and callable(getattr(obj, "close", None)))


class GPUDistributor(_Base):
    """`map_reduce` on one MI355X.  `device`: HIP ordinal (default: the package default, see extract_features)."""

    def __init__(self, device=None):
        self.device = device

    def _extract_matrix(self, fc_parameters, values, offsets, times, has_dt):
        """-> (column names without the kind prefix, float64 [n_series, n_cols]) through the native plan cache."""
        from tsfresh_amd import _native  # noqa: F401  (fails loudly without the library: there is no CPU path)
        from tsfresh_amd.feature_extraction.extraction import _acquire_plan, _default_device
        from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
        fplan = compile_fc_parameters(fc_parameters, has_datetime_index=has_dt)
        n_series = len(offsets) - 1
        if len(fplan) == 0:
            return [], np.empty((n_series, 0))
        if fplan.names:
            device = self.device if self.device is not None else _default_device()
            plan = _acquire_plan(fplan, device)  # cached per thread: no 8 ms plan build per call
            matrix = plan.extract_host(values, offsets, times=times)
            from tsfresh_amd.feature_extraction.reference_errors import check_reference_data_errors
            check_reference_data_errors(fplan.specs, matrix, values, offsets[:-1], offsets[1:])
        else:
            matrix = np.empty((n_series, 0))
        names, matrix = fplan.finish(matrix, lambda i: values[offsets[i]:offsets[i + 1]], n_series)
        return list(names), matrix

    def map_reduce(self, map_function=None, data=None, function_kwargs=None, chunk_size=None, data_length=None):
        kwargs = function_kwargs or {}
        default_fc = kwargs.get("default_fc_parameters")
        kind_to_fc = kwargs.get("kind_to_fc_parameters")
        show_warnings = kwargs.get("show_warnings", False)
        # group the chunks by kind (the ORIGINAL key: kind_to_fc_parameters is looked up with it, extraction.py:333-336)
        # and, within a kind, by whether the series carries a DatetimeIndex -- the reference decides per series whether
        # linear_trend_timewise can be computed (extraction.py:349-358), so a kind with mixed indices gets the timewise
        # columns for the series that have timestamps and not for the others
        by_kind = {}
        for chunk in data:
            sample_id, kind, series = chunk[0], chunk[1], chunk[2]
            has_dt = isinstance(series, pd.Series) and isinstance(series.index, pd.DatetimeIndex)
            by_kind.setdefault(kind, {}).setdefault(has_dt, []).append((sample_id, series))
        result = []
        with warnings.catch_warnings():
            warnings.simplefilter("default" if show_warnings else "ignore")
            for kind, groups in by_kind.items():
                if kind_to_fc and kind in kind_to_fc:
                    fc_parameters = kind_to_fc[kind]
                elif kind_to_fc and str(kind) in kind_to_fc:
                    fc_parameters = kind_to_fc[str(kind)]
                else:
                    fc_parameters = default_fc
                for has_dt, items in groups.items():
                    arrays = [np.asarray(s) for _, s in items]
                    if any(len(a) == 0 for a in arrays):
                        raise ValueError("every series must hold at least one sample")
                    dtype = np.float32 if all(a.dtype == np.float32 for a in arrays) else np.float64
                    values = np.concatenate([a.astype(dtype, copy=False) for a in arrays])
                    offsets = np.zeros(len(arrays) + 1, dtype=np.int64)
                    np.cumsum([len(a) for a in arrays], out=offsets[1:])
                    times = None
                    if has_dt:  # feature_calculators.py:2291-2296
                        times = np.concatenate([np.asarray((s.index - s.index[0]).total_seconds() / float(3600))
                                                for _, s in items]).astype(np.float64)
                    names, matrix = self._extract_matrix(fc_parameters, values, offsets, times, has_dt)
                    names = [str(kind) + "__" + n for n in names]
                    for r, (sample_id, _) in enumerate(items):
                        row = matrix[r]
                        result.extend((sample_id, name, row[c]) for c, name in enumerate(names))
        return result

    def close(self):
        pass
