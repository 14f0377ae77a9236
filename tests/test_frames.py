"""Frame-level goldens: `tsfresh_amd.extract_features(container, ...)` must return the frame the
REAL `tsfresh.extract_features` returned for the same container -- same index values and dtype, same column names, values
within the parity bar.  Fixtures: tests/golden/ref_frames_{main,conda}.json (gen_golden_frames.py; they hold the inputs
too).  The CPU tests run the DataFrame-level code of the product over the g++ build of the kernel sources (EmulPlan in
place of the native plan); the `-m gpu` twin runs the product as shipped."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from golden.frame_codec import decode_container, decode_frame
from parity import compare

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    cases = []
    for f in ("ref_frames_main.json", "ref_frames_conda.json"):
        with open(os.path.join(G, f)) as fh:
            cases += json.load(fh)["cases"]
    return cases


CASES = _load()
THIRD_PARTY = ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller",
               "ar_coefficient")


def _params(name):
    from tsfresh_amd.feature_extraction import settings
    if name is None:
        return None
    if name == "minimal":
        return settings.MinimalFCParameters()
    full = settings.EfficientFCParameters() if name.startswith("efficient") else settings.ComprehensiveFCParameters()
    if name.endswith("_no3p"):
        return {k: v for k, v in full.items() if k not in THIRD_PARTY}
    if name.endswith("_only3p"):
        return {k: v for k, v in full.items() if k in THIRD_PARTY}
    return full


def _series_by_kind(container, call):
    """{kind: {id: float64 samples in sort order}} -- a plain pandas restatement of the reference's data adapters
    (data.py:183-338), only used to hand parity.compare the series a row was computed from."""
    cid, csort, ckind, cval = (call.get(k) for k in ("column_id", "column_sort", "column_kind", "column_value"))
    out = {}

    def add(kind, df, value_col):
        d = df.sort_values(csort, kind="stable") if csort else df
        for sid, g in d.groupby(cid, sort=False):
            out.setdefault(str(kind), {})[sid] = g[value_col].to_numpy(dtype=np.float64)

    if isinstance(container, dict):
        for kind, df in container.items():
            add(kind, df, cval)
    elif ckind is not None:
        vcol = cval or [c for c in container.columns if c not in (cid, csort, ckind)][0]
        for kind, df in container.groupby(ckind, sort=False):
            add(kind, df, vcol)
    else:
        for c in container.columns:
            if c not in (cid, csort):
                add(c, container, c)
    return out


def _check(case, got):
    want = decode_frame(case["output"])
    assert list(got.index) == list(want.index), (case["name"], list(got.index)[:5], list(want.index)[:5])
    assert got.index.dtype == want.index.dtype, (case["name"], got.index.dtype, want.index.dtype)
    assert set(got.columns) == set(want.columns), (case["name"], sorted(set(got.columns) ^ set(want.columns))[:6])
    assert all(t == np.float64 for t in got.dtypes)
    series = _series_by_kind(decode_container(case["input"]), case["call"])
    kinds = sorted({c.split("__")[0] for c in want.columns})
    for kind in kinds:
        cols = [c for c in want.columns if c.split("__")[0] == kind]
        rows = [i for i in want.index if i in series.get(kind, {})]
        missing = [i for i in want.index if i not in series.get(kind, {})]
        if missing:   # an id without this kind: NaN for every column of the kind (data.py:86-121 pivot)
            assert np.isnan(got.loc[missing, cols].to_numpy()).all() and np.isnan(want.loc[missing, cols].to_numpy()).all()
        names = ["value__" + c.split("__", 1)[1] for c in cols]
        # float32 columns fed as is: the reference then computes partly in float32; the gate is its value for
        # x.astype(float64) and this comparison is the REPORT beside it (profiles/r03_float32_as_is.md: <= 4.1e-4)
        rtol = 2e-3 if "float32" in case["name"] else 1e-6
        bad = compare(names, got.loc[rows, cols].to_numpy(), want.loc[rows, cols].to_numpy(),
                      [series[kind][i] for i in rows], rtol=rtol)
        assert not bad, (case["name"], kind, bad[:6])


def _run(case):
    from tsfresh_amd import extract_features
    kwargs = dict(case["call"])
    if case["params"] is not None:
        kwargs["default_fc_parameters"] = _params(case["params"])
    if case["kind_to_fc_parameters"] is not None:
        kwargs["kind_to_fc_parameters"] = case["kind_to_fc_parameters"]
    return extract_features(decode_container(case["input"]), **kwargs)


@pytest.fixture
def emul_plans(monkeypatch):
    from emul_lib import EmulPlan
    from tsfresh_amd.feature_extraction import extraction
    monkeypatch.setattr(extraction, "_acquire_plan", lambda fplan, device, pins=None: EmulPlan(fplan))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_frame_equals_the_reference_frame_emulated(case, emul_plans):
    _check(case, _run(case))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_frame_equals_the_reference_frame(case, gpu):
    _check(case, _run(case))


def test_the_reference_s_own_assertions_hold_in_the_fixture():
    """tests/units/feature_extraction/test_extraction.py:40-55: the exact integers the reference asserts."""
    want = decode_frame(next(c for c in CASES if c["name"] == "reference_test_data_sample")["output"])
    assert list(want["a__maximum"]) == [71, 77] and list(want["a__sum_values"]) == [691, 1017]
    assert list(want["a__abs_energy"]) == [32211, 63167] and list(want["b__sum_values"]) == [757, 695]
    assert list(want["b__minimum"]) == [3, 1] and list(want["b__abs_energy"]) == [36619, 35483]
    assert list(want["b__mean"]) == [37.85, 34.75] and list(want["b__median"]) == [39.5, 28.0]
    drift = decode_frame(next(c for c in CASES if c["name"] == "driftbif_kind_value")["output"])
    assert drift.shape == (100, 20) and "1__mean" in drift.columns and "11" in drift.index
    assert abs(drift.loc["5", "1__mean"] - 5.516e-05) < 1e-4   # tests/integrations/test_feature_extraction.py:33-38


def test_shuffled_rows_give_the_same_frame(emul_plans):
    """test_extraction.py:207-237"""
    a = _run(next(c for c in CASES if c["name"] == "reference_test_data_sample"))
    b = _run(next(c for c in CASES if c["name"] == "reference_test_data_sample_shuffled"))
    pd.testing.assert_frame_equal(a, b[a.columns])


def test_more_distinct_per_kind_plans_than_the_cache_holds(monkeypatch):
    """A frame whose kinds compile to more distinct native plans than the per-thread LRU holds (the usual from_columns /
    extract_relevant_features output): every plan acquired by the running call stays open until the call has used it;
    the cache is trimmed afterwards.  (Round-3 ADVICE: the 7th acquire closed the 1st plan before it ran.)"""
    from tsfresh_amd import _native, extract_features
    from tsfresh_amd.feature_extraction import extraction

    log = {"open": 0, "closed": 0, "ran_closed": 0}

    class FakePlan:
        def __init__(self, specs, device=0):
            self.specs, self.closed = list(specs), False
            log["open"] += 1

        def extract_host(self, values, offsets, times=None):
            if self.closed:
                log["ran_closed"] += 1
                raise RuntimeError("plan is NULL")
            from emul_lib import emul_extract_specs
            return emul_extract_specs(self.specs, values, offsets, times=times)

        def close(self):
            if not self.closed:
                self.closed = True
                log["closed"] += 1

    from emul_lib import load
    lib = load()
    monkeypatch.setattr(_native, "Plan", FakePlan)
    monkeypatch.setattr(_native, "calc_id", lambda name: lib.tsfa_emul_calc_id(name.encode()))  # the emulation's table
    extraction.clear_plan_cache()
    n_kinds = extraction._PLAN_CACHE_SIZE + 3
    rng = np.random.default_rng(3)
    frames = []
    for k in range(n_kinds):
        frames.append(pd.DataFrame({"id": np.repeat(np.arange(4), 12), "time": np.tile(np.arange(12), 4),
                                    "kind": "k%d" % k, "value": rng.standard_normal(48)}))
    df = pd.concat(frames, ignore_index=True)
    k2fc = {"k%d" % k: {"quantile": [{"q": 0.1 + 0.05 * k}], "mean": None} for k in range(n_kinds)}
    try:
        got = extract_features(df, column_id="id", column_sort="time", column_kind="kind", column_value="value",
                               kind_to_fc_parameters=k2fc)
        assert log["ran_closed"] == 0 and log["open"] == n_kinds
        assert got.shape == (4, 2 * n_kinds)
        for k in range(n_kinds):
            x = df[df.kind == "k%d" % k].groupby("id")["value"]
            np.testing.assert_allclose(got["k%d__mean" % k].to_numpy(), x.mean().to_numpy(), rtol=1e-12)
            np.testing.assert_allclose(got["k%d__quantile__q_%s" % (k, 0.1 + 0.05 * k)].to_numpy(),
                                       x.quantile(0.1 + 0.05 * k).to_numpy(), rtol=1e-12)
        # trimmed back after the call: only the newest _PLAN_CACHE_SIZE plans stay open
        assert log["closed"] == 3 and len(extraction._thread_cache()) == extraction._PLAN_CACHE_SIZE
    finally:
        extraction.clear_plan_cache()


def test_long_series_with_entropy_columns_warn_about_their_run_time(emul_plans):
    """VERDICT r3 'long-series cliff': beyond 4096 samples the O(n^2) entropies leave the LDS sweep; the call still works
    but says so -- even with show_warnings=False, which only silences the calculators' own domain warnings."""
    import warnings as w
    from tsfresh_amd import extract_features
    from tsfresh_amd.feature_extraction import extraction
    n = extraction.ENTROPY_FAST_MAX_LEN + 1
    df = pd.DataFrame({"id": 0, "time": np.arange(n), "value": np.sin(np.arange(n) * 0.01)})
    extraction._LONG_ENTROPY_WARNED = False
    with pytest.warns(UserWarning, match="O\\(n\\^2\\)"):
        extract_features(df, column_id="id", column_sort="time", default_fc_parameters={"sample_entropy": None, "mean": None})
    with w.catch_warnings():       # once per process under the default show_warnings=False (round-4 ADVICE) ...
        w.simplefilter("error")
        extract_features(df, column_id="id", column_sort="time", default_fc_parameters={"sample_entropy": None, "mean": None})
    with pytest.warns(UserWarning, match="O\\(n\\^2\\)"):   # ... every call with show_warnings=True
        extract_features(df, column_id="id", column_sort="time", default_fc_parameters={"sample_entropy": None, "mean": None},
                         show_warnings=True)
    with w.catch_warnings():
        w.simplefilter("error")    # no warning without the quadratic calculators, nor at the limit itself
        extract_features(df, column_id="id", column_sort="time", default_fc_parameters={"mean": None, "median": None})
        extract_features(df.iloc[:-1], column_id="id", column_sort="time", default_fc_parameters={"sample_entropy": None})
