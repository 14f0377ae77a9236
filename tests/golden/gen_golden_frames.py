"""Generate tests/golden/ref_frames_{main,conda}.json: WHOLE result frames of the real `tsfresh.extract_features`
(index values and dtype, column names in order, values), together with the input containers they were computed from.

    python tests/golden/gen_golden_frames.py                       # main interpreter, ref_frames_main.json
    /opt/conda/bin/python3.9 tests/golden/gen_golden_frames.py     # second interpreter, ref_frames_conda.json

Neither interpreter of the build container imports the whole reference (gen_golden_main.py / gen_golden_conda.py say
why), so the main run leaves out the five calculators that live in statsmodels / PyWavelets and the conda run holds the
cases that exist for them; everything else is the unmodified reference -- tsfresh.extract_features with its own
to_tsdata adapters, MapDistributor (n_jobs = 0) and pivot.

Cases:
  * the reference's own test inputs, imported from /root/reference/tests (never copied into the repo as source):
    tests/fixtures.py:28 create_test_data_sample with the calls of tests/units/feature_extraction/test_extraction.py
    :40-55 (exact integer results) and :207-237 (shuffled rows), the wide and dict containers of fixtures.py,
    load_driftbif(100, 10, classification=True, seed=42) with string ids as tests/integrations/test_feature_extraction.py
    :16-60 calls it;
  * BASELINE config 1's stand-in: 88 ids x 15 steps x 6 integer kinds (the shape and value range of the robot execution
    failures data, examples/robot_execution_failures.py:95,127-129 -- the data itself is a download), MinimalFCParameters;
  * string ids, tuple ids (the window ids of roll_time_series), unsorted ids without a sort column, float32 values,
    kind_to_fc_parameters, series of unequal length per kind, a kind missing for some ids.
"""
import json
import os
import sys
import types
import warnings

warnings.filterwarnings("ignore")
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CONDA = sys.version_info[:2] == (3, 9)

if CONDA:
    class _MachAr:
        def __init__(self, *a, **k):
            fi = np.finfo(float)
            self.eps, self.tiny, self.huge, self.epsneg, self.xmin, self.xmax = fi.eps, fi.tiny, fi.max, fi.epsneg, fi.tiny, fi.max
    if not hasattr(np, "MachAr"):
        np.MachAr = _MachAr
    for _n in ("Int64Index", "Float64Index", "UInt64Index"):
        if not hasattr(pd, _n):
            setattr(pd, _n, pd.Index)
    _st = types.ModuleType("stumpy")
    _st.core = types.SimpleNamespace()
    sys.modules["stumpy"] = _st
    sys.modules["dask"] = None
    sys.modules["distributed"] = None
else:
    class _Raiser(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)

            def _fail(*a, **k):
                raise RuntimeError("stubbed third-party module %s.%s was called" % (self.__name__, item))
            return _fail
    for mod in ("pywt", "stumpy", "statsmodels", "statsmodels.tools", "statsmodels.tools.sm_exceptions", "statsmodels.tsa",
                "statsmodels.tsa.ar_model", "statsmodels.tsa.stattools", "statsmodels.stats", "statsmodels.stats.multitest"):
        sys.modules[mod] = _Raiser(mod)
    sys.modules["statsmodels.tools.sm_exceptions"].MissingDataError = type("MissingDataError", (Exception,), {})
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

from tsfresh import extract_features  # noqa: E402
from tsfresh.feature_extraction import settings as ref_settings  # noqa: E402

from frame_codec import encode_container, encode_frame  # noqa: E402

THIRD_PARTY = ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller",
               "ar_coefficient")


def _params(name):
    """Named parameter sets (the JSON stores the NAME; tests rebuild them from tsfresh_amd's settings)."""
    if name == "minimal":
        return ref_settings.MinimalFCParameters()
    full = ref_settings.EfficientFCParameters() if name.startswith("efficient") else ref_settings.ComprehensiveFCParameters()
    if name.endswith("_no3p"):
        return {k: v for k, v in full.items() if k not in THIRD_PARTY}
    if name.endswith("_only3p"):
        return {k: v for k, v in full.items() if k in THIRD_PARTY}
    return full


def _reference_fixtures():
    sys.path.insert(0, "/root/reference")
    from tests.fixtures import DataTestCase
    t = DataTestCase()
    return t


def main_cases():
    rng = np.random.default_rng(20260925)
    cases = []
    fx = _reference_fixtures()
    sample = fx.create_test_data_sample()
    call = dict(column_id="id", column_sort="sort", column_kind="kind", column_value="val")
    cases.append(("reference_test_data_sample", sample, call, "comprehensive_no3p", None))
    cases.append(("reference_test_data_sample_shuffled", sample.sample(frac=1, random_state=3), call, "comprehensive_no3p", None))
    cases.append(("reference_test_data_sample_minimal", sample, call, "minimal", None))
    wide = fx.create_test_data_sample_wide()
    cases.append(("reference_test_data_sample_wide", wide, dict(column_id="id", column_sort="sort"), "minimal", None))
    one = fx.create_one_valued_time_series()
    cases.append(("reference_one_valued_series", one, call, "efficient_no3p", None))
    near = fx.create_test_data_nearly_numerical_indices()
    cases.append(("reference_nearly_numerical_indices", near, call, "minimal", None))
    # dict container: kind -> frame, as fixtures.py builds it from the sample
    d = {k: g[["id", "sort", "val"]].reset_index(drop=True) for k, g in sample.groupby("kind")}
    cases.append(("dict_container", d, dict(column_id="id", column_sort="sort", column_value="val"), "minimal", None))

    from tsfresh.examples.driftbif_simulation import load_driftbif
    df, _ = load_driftbif(100, 10, classification=True, seed=42)
    df["my_id"] = df["id"].astype("str")
    del df["id"]
    cases.append(("driftbif_kind_value", df, dict(column_id="my_id", column_sort="time", column_kind="dimension",
                                                   column_value="value"), "minimal", None))
    cases.append(("driftbif_kind_only", df, dict(column_id="my_id", column_sort="time", column_kind="dimension"),
                  "minimal", None))
    cases.append(("driftbif_no_kind", df.drop(columns=["dimension"]), dict(column_id="my_id", column_sort="time"),
                  "minimal", None))

    # BASELINE config 1's stand-in: 88 ids x 15 steps x 6 integer kinds, wide format
    n_ids, steps = 88, 15
    robot = pd.DataFrame({"id": np.repeat(np.arange(1, n_ids + 1), steps), "time": np.tile(np.arange(steps), n_ids)})
    for j, kind in enumerate(["F_x", "F_y", "F_z", "T_x", "T_y", "T_z"]):
        scale = [3, 3, 40, 12, 8, 2][j]
        robot[kind] = np.rint(scale * rng.standard_normal(n_ids * steps) + [-1, 1, 50, -10, -4, 0][j]).astype(np.int64)
    cases.append(("robot_like_88x15x6_minimal", robot, dict(column_id="id", column_sort="time"), "minimal", None))

    # ragged float frame: string ids in non-sorted order, two kinds of unequal length, one kind missing for one id
    rows = []
    for sid, na, nb in (("sensor_b", 40, 25), ("sensor_a", 33, 0), ("10", 18, 30), ("9", 50, 50)):
        for kind, n in (("temp", na), ("press", nb)):
            v = np.cumsum(rng.standard_normal(n)) + (300.0 if kind == "temp" else 0.0)
            rows.append(pd.DataFrame({"id": sid, "t": rng.permutation(n), "kind": kind, "value": v}))
    ragged = pd.concat(rows, ignore_index=True).sample(frac=1, random_state=1).reset_index(drop=True)
    cases.append(("ragged_string_ids_efficient", ragged, dict(column_id="id", column_sort="t", column_kind="kind",
                                                               column_value="value"), "efficient_no3p", None))
    k2fc = {"temp": {"mean": None, "quantile": [{"q": 0.1}, {"q": 0.9}], "fft_coefficient": [{"coeff": 1, "attr": "abs"}]},
            "press": {"maximum": None, "number_peaks": [{"n": 1}, {"n": 3}], "linear_trend": [{"attr": "slope"}]}}
    cases.append(("kind_to_fc_parameters", ragged, dict(column_id="id", column_sort="t", column_kind="kind",
                                                         column_value="value"), None, k2fc))
    # no sort column, integer ids in descending order, float32 values
    f32 = pd.DataFrame({"id": np.repeat([7, 3, 5], 64), "value": rng.standard_normal(192).astype(np.float32)})
    cases.append(("float32_no_sort_column", f32, dict(column_id="id"), "efficient_no3p", None))
    cases.append(float32_report_case("comprehensive_no3p"))
    # tuple ids: the frame roll_time_series makes of a small two-kind frame
    from tsfresh.utilities.dataframe_functions import roll_time_series
    small = pd.DataFrame({"id": np.repeat(["x", "y"], 6), "time": np.tile(np.arange(6), 2),
                          "a": rng.standard_normal(12), "b": rng.integers(0, 9, 12).astype(float)})
    rolled = roll_time_series(small, column_id="id", column_sort="time", max_timeshift=3, min_timeshift=1, n_jobs=0,
                              disable_progressbar=True)
    cases.append(("tuple_ids_rolled", rolled, dict(column_id="id", column_sort="time"), "minimal", None))
    return cases


def float32_report_case(params_name):
    """The BASELINE dtype fed AS IS (SURVEY H2 option a): float32 columns, on which the reference's numpy / pandas calls
    compute partly in float32.  Not a gate -- the gate is x.astype(float64) -- but the data of DESIGN's deviation table
    (profiles/float32_as_is_report.py)."""
    rng = np.random.default_rng(20260927)
    n = 1024
    rows = []
    for sid in range(4):
        rows.append(pd.DataFrame({"id": sid, "time": np.arange(n), "value": rng.standard_normal(n, dtype=np.float32)}))
    for sid in range(4, 8):
        rows.append(pd.DataFrame({"id": sid, "time": np.arange(n),
                                  "value": np.cumsum(rng.standard_normal(n, dtype=np.float32)).astype(np.float32)}))
    df = pd.concat(rows, ignore_index=True)
    assert df["value"].dtype == np.float32
    return ("float32_as_is_%s" % params_name, df, dict(column_id="id", column_sort="time"), params_name, None)


def conda_cases():
    rng = np.random.default_rng(20260926)
    rows = []
    for sid in (3, 1, 2):
        for kind, n in (("u", 120), ("v", 64)):
            v = np.cumsum(rng.standard_normal(n)) if kind == "u" else 5.0 + rng.standard_normal(n)
            rows.append(pd.DataFrame({"id": sid, "time": np.arange(n), "kind": kind, "value": v}))
    df = pd.concat(rows, ignore_index=True)
    call = dict(column_id="id", column_sort="time", column_kind="kind", column_value="value")
    return [("third_party_calculators_two_kinds", df, call, "comprehensive_only3p", None),
            float32_report_case("comprehensive_only3p")]


def main():
    cases = conda_cases() if CONDA else main_cases()
    out = []
    for name, container, call, params_name, k2fc in cases:
        kwargs = dict(call)
        if params_name is not None:
            kwargs["default_fc_parameters"] = _params(params_name)
        if k2fc is not None:
            kwargs["kind_to_fc_parameters"] = k2fc
        res = extract_features(container, n_jobs=0, disable_progressbar=True, **kwargs)
        out.append({"name": name, "call": call, "params": params_name, "kind_to_fc_parameters": k2fc,
                    "input": encode_container(container), "output": encode_frame(res)})
        print("%-40s -> %s" % (name, res.shape))
    path = os.path.join(HERE, "ref_frames_%s.json" % ("conda" if CONDA else "main"))
    with open(path, "w") as f:
        json.dump({"versions": {"python": sys.version.split()[0], "numpy": np.__version__, "pandas": pd.__version__},
                   "cases": out}, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
