"""FCParameters dict -> flat list of output columns ("feature plan").

This replaces the triple Python loop of the reference's per-series dispatcher
(tsfresh/feature_extraction/extraction.py:338-378: for calculator / for parameter set / for (key, value)) by a
one-time compilation: every (calculator, parameter-dict) pair becomes one output column with the exact name the
reference would emit, plus the numeric spec handed to `tsfa_plan_create`.
"""
import warnings

from tsfresh_amd.feature_extraction.registry import CALCULATORS, UnsupportedFeature


class FeaturePlan:
    """Column names (without the kind prefix) and native specs of one FCParameters mapping."""

    def __init__(self, names, specs):
        self.names = names    # list[str]  "<calculator>[__<params>]"
        self.specs = specs    # list[(calculator name, (p0, p1, p2, p3))], aligned with names

    def __len__(self):
        return len(self.names)

    def native_specs(self, calc_id):
        """[(id, p)] with ids resolved through the C library's registry (tsfa_calc_id)."""
        out = []
        for name, p in self.specs:
            cid = calc_id(name)
            if cid < 0:
                raise UnsupportedFeature("{} is not known to libtsfresh_amd".format(name))
            out.append((cid, p))
        return out


def compile_fc_parameters(fc_parameters, has_datetime_index=False):
    """Compile one FCParameters mapping (settings.py) to a :class:`FeaturePlan`.

    Column order follows the reference: calculators in dict order, parameter sets in list order
    (extraction.py:339-378).  Raises UnsupportedFeature for anything that has no native kernel -- custom
    callables, matrix_profile, query_similarity_count with a query --
    instead of silently computing it on the CPU.
    """
    names, specs, seen = [], [], set()
    for key, param_list in fc_parameters.items():
        if callable(key):
            raise UnsupportedFeature(
                "custom calculator {!r}: user-defined Python calculators cannot run in the native path".format(
                    getattr(key, "__name__", key)))
        if key not in CALCULATORS:
            raise AttributeError("module 'feature_calculators' has no attribute {!r}".format(key))
        calc = CALCULATORS[key]
        if calc.index_type is not None:
            if not has_datetime_index:
                # same message as the reference (extraction.py:353-357)
                warnings.warn("{} requires the data to have a index of type {}. Results will "
                              "not be calculated".format(key, "<class 'pandas.core.indexes.datetimes.DatetimeIndex'>"))
                continue
        if not calc.native:
            raise UnsupportedFeature("{} has no native kernel".format(key))
        params = param_list if param_list else [None]
        for param in params:
            suffix = calc.key(param)
            name = key if not suffix else key + "__" + suffix
            if name in seen:  # a repeated parameter set maps to the same column (pivot keeps one)
                continue
            seen.add(name)
            names.append(name)
            specs.append((key, calc.encode(param) if param is not None else (0.0, 0.0, 0.0, 0.0)))
    return FeaturePlan(names, specs)
