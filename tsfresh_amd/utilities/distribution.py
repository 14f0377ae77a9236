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


class GPUDistributor(_Base):
    """`map_reduce` on one MI355X.  `device`: HIP ordinal (default: the package default, see extract_features)."""

    def __init__(self, device=None):
        self.device = device

    def map_reduce(self, map_function=None, data=None, function_kwargs=None, chunk_size=None, data_length=None):
        from tsfresh_amd import _native
        from tsfresh_amd.feature_extraction.extraction import _default_device
        from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
        kwargs = function_kwargs or {}
        default_fc = kwargs.get("default_fc_parameters")
        kind_to_fc = kwargs.get("kind_to_fc_parameters")
        show_warnings = kwargs.get("show_warnings", False)
        device = self.device if self.device is not None else _default_device()
        # group the chunks by kind, keeping the order in which kinds and ids arrive
        by_kind = {}
        for chunk in data:
            sample_id, kind, series = chunk[0], chunk[1], chunk[2]
            by_kind.setdefault(str(kind), []).append((sample_id, series))
        result = []
        with warnings.catch_warnings():
            warnings.simplefilter("default" if show_warnings else "ignore")
            for kind, items in by_kind.items():
                fc_parameters = kind_to_fc[kind] if kind_to_fc and kind in kind_to_fc else default_fc
                arrays = [np.asarray(s) for _, s in items]
                if any(len(a) == 0 for a in arrays):
                    raise ValueError("every series must hold at least one sample")
                dtype = np.float32 if all(a.dtype == np.float32 for a in arrays) else np.float64
                values = np.concatenate([a.astype(dtype, copy=False) for a in arrays])
                offsets = np.zeros(len(arrays) + 1, dtype=np.int64)
                np.cumsum([len(a) for a in arrays], out=offsets[1:])
                has_dt = all(isinstance(s, pd.Series) and isinstance(s.index, pd.DatetimeIndex) for _, s in items)
                times = None
                if has_dt:  # feature_calculators.py:2291-2296
                    times = np.concatenate([np.asarray((s.index - s.index[0]).total_seconds() / float(3600))
                                            for _, s in items]).astype(np.float64)
                fplan = compile_fc_parameters(fc_parameters, has_datetime_index=has_dt)
                if len(fplan) == 0:
                    continue
                plan = _native.Plan(fplan.native_specs(_native.calc_id), device=device)
                try:
                    matrix = plan.extract_host(values, offsets, times=times)
                finally:
                    plan.close()
                names = [kind + "__" + n for n in fplan.names]
                for r, (sample_id, _) in enumerate(items):
                    row = matrix[r]
                    result.extend((sample_id, name, row[c]) for c, name in enumerate(names))
        return result

    def close(self):
        pass
