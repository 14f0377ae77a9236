"""augmented_dickey_fuller with the lag selections ComprehensiveFCParameters does not use -- autolag "BIC", "t-stat", None
(fc.py:499-545 passes the value through to statsmodels.adfuller) -- against statsmodels' own output
(tests/golden/ref_conda_*_adf.npz: `gen_golden_conda.py --params adf`, real statsmodels 0.12.2 under the second interpreter)
on the main / degenerate / offset / long series.  A plan holds ONE autolag value (the kernels keep one fit per series)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from engines import emul_engine, oracle_engine
from parity import compare

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = {"main": "ref_conda_adf.npz", "degenerate": "ref_conda_degenerate_adf.npz", "offset": "ref_conda_offset_adf.npz",
         "long": "ref_conda_long_adf.npz"}
# skipped cells (tests/parity.py R4 / R5, per autolag value): the degenerate set is made of exact ramps and constants
SKIP_BOUND = {"main": 0.05, "degenerate": 0.55, "offset": 0.07, "long": 0.0}
MODES = ["BIC", "t-stat", None]


def _params(mode):
    return {"augmented_dickey_fuller": [{"attr": a, "autolag": mode} for a in ("teststat", "pvalue", "usedlag")]}


def _check(engine, set_name, mode):
    g = np.load(os.path.join(G, FILES[set_name]))
    v, o = g["values"], g["offsets"]
    series = [v[o[i]:o[i + 1]] for i in range(len(o) - 1)]
    names, got = engine(_params(mode), v, o)
    ref_names = list(g["names"])
    want = g["matrix"][:, [ref_names.index(n) for n in names]]
    skipped = []
    bad = compare(names, got, want, series, skipped=skipped)
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:8])
    assert len(skipped) <= SKIP_BOUND[set_name] * want.size, (len(skipped), want.size)


@pytest.mark.parametrize("mode", MODES, ids=[str(m) for m in MODES])
@pytest.mark.parametrize("set_name", sorted(FILES))
@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_engine_matches_statsmodels(engine, set_name, mode):
    _check(engine, set_name, mode)


def test_every_lag_selection_compiles_by_default():
    """Round 5: the selections other than "AIC" ran on the device (profiles/r05_a_pytest_new.log); no gate is left."""
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    for mode in MODES + ["AIC", "aic", "bic"]:
        compile_fc_parameters(_params(mode))
    compile_fc_parameters({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": "AIC"}, {"attr": "pvalue"}]})


def test_autolag_values_statsmodels_rejects():
    """A string statsmodels does not know -> ValueError inside adfuller -> fc.py:523 returns NaNs: a NaN column, also
    BESIDE columns with a valid lag selection (round-4 ADVICE: it was refused as 'one autolag value per plan'); a
    non-string (5) -> statsmodels' TypeError, which the reference does not catch."""
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    x = np.cumsum(np.random.default_rng(3).standard_normal(200))
    params = {"augmented_dickey_fuller": [{"attr": "teststat", "autolag": "BIC"}, {"attr": "teststat", "autolag": "nonsense"},
                                          {"attr": "usedlag", "autolag": "BIC"}, {"attr": "pvalue", "autolag": "None"}]}
    names, got = emul_engine(params, x, np.array([0, 200], dtype=np.int64))
    _, want = oracle_engine({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": "BIC"}, {"attr": "usedlag", "autolag": "BIC"}]},
                            x, np.array([0, 200], dtype=np.int64))
    by = dict(zip(names, got[0]))
    assert np.isnan(by['value__augmented_dickey_fuller__attr_"teststat"__autolag_"nonsense"'])
    assert np.isnan(by['value__augmented_dickey_fuller__attr_"pvalue"__autolag_"None"'])
    assert abs(by['value__augmented_dickey_fuller__attr_"teststat"__autolag_"BIC"'] - want[0, 0]) <= 1e-6 * abs(want[0, 0])
    assert by['value__augmented_dickey_fuller__attr_"usedlag"__autolag_"BIC"'] == want[0, 1]
    with pytest.raises(TypeError):
        compile_fc_parameters({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": 5}]})


def test_one_autolag_value_per_plan():
    with pytest.raises(RuntimeError) as e:
        emul_engine({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": "BIC"}, {"attr": "teststat", "autolag": None}]},
                    np.arange(50.0), np.array([0, 50], dtype=np.int64))
    assert "one autolag value per plan" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES, ids=[str(m) for m in MODES])
@pytest.mark.parametrize("set_name", sorted(FILES))
def test_hip_matches_statsmodels(gpu, set_name, mode):
    from engines import hip_engine
    _check(hip_engine, set_name, mode)


@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_an_unknown_attr_is_nan_as_in_the_reference(engine):
    """fc.py:543: any other `attr` yields NaN (the column exists, under the name the caller gave it)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(200)
    names, got = engine({"augmented_dickey_fuller": [{"attr": "teststat"}, {"attr": "no_such_attr"}]}, x, np.array([0, 200], dtype=np.int64))
    assert names[1] == 'value__augmented_dickey_fuller__attr_"no_such_attr"__autolag_"AIC"'
    assert np.isfinite(got[0, 0]) and np.isnan(got[0, 1])


@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_the_string_none_is_not_none(engine):
    """settings.from_columns turns autolag_"None" into the STRING "None"; statsmodels raises ValueError for it and
    fc.py:523 answers (nan, nan, nan) -- verified against the real libraries: NaN for "None", -4.4378 for None."""
    x = np.random.default_rng(1).standard_normal(200)
    o = np.array([0, 200], dtype=np.int64)
    names, got = engine({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": "None"}]}, x, o)
    assert names == ['value__augmented_dickey_fuller__attr_"teststat"__autolag_"None"'] and np.isnan(got[0, 0])
    _, got = engine({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": None}]}, x, o)
    assert abs(got[0, 0] - (-4.437821986643767)) < 1e-9
    _, got = engine({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": "bic"}]}, x, o)
    assert abs(got[0, 0] - (-13.046525438501202)) < 1e-8


def test_several_autolag_values_in_one_settings_object(monkeypatch):
    """fc.py:499-545 evaluates every dict of the list on its own, so a settings object may mix lag selections; a native plan
    holds one (tsfa_validate_plan).  extract_features splits such a plan into one native plan per value and scatters the
    columns back (feature_extraction/extraction.py: _CompositePlan) -- here with the emulated kernels behind the parts."""
    import pandas as pd
    from emul_lib import emul_extract_specs
    from tsfresh_amd import extract_features
    from tsfresh_amd.feature_extraction import extraction

    class _Part:
        def __init__(self, specs):
            self.specs = list(specs)

        def extract_host(self, values, offsets, times=None):
            return emul_extract_specs(self.specs, values, offsets, times=times)

    made = []
    monkeypatch.setattr(extraction, "_acquire_plan_specs", lambda specs, device, pins=None: made.append(len(list(specs))) or _Part(specs))
    rng = np.random.default_rng(8)
    lens = [120, 300, 75]
    df = pd.DataFrame({"id": np.repeat(np.arange(3), lens), "time": np.concatenate([np.arange(m) for m in lens]),
                       "value": np.concatenate([np.cumsum(rng.standard_normal(m)) for m in lens])})
    params = {"mean": None,
              "augmented_dickey_fuller": [{"attr": "teststat", "autolag": "BIC"}, {"attr": "pvalue", "autolag": "AIC"},
                                          {"attr": "usedlag", "autolag": None}, {"attr": "teststat", "autolag": "AIC"},
                                          {"attr": "teststat", "autolag": "bogus"}, {"attr": "usedlag", "autolag": "t-stat"}],
              "median": None}
    got = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params, device=0)
    assert made == [4, 2, 1, 1]          # mean + BIC + the NaN column + median | AIC x 2 | None | t-stat
    values = df["value"].to_numpy()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    onames, want = oracle_engine(params, values, offsets)
    assert list(got.columns) == onames
    bad = compare(onames, got.to_numpy(), want, [values[offsets[i]:offsets[i + 1]] for i in range(3)])
    assert not bad, bad
    assert got['value__augmented_dickey_fuller__attr_"teststat"__autolag_"bogus"'].isna().all()


@pytest.mark.gpu
def test_hip_several_autolag_values_in_one_settings_object(gpu):
    """The same through the real native plans (one per lag selection), also for rolled windows."""
    import pandas as pd
    from tsfresh_amd import extract_features
    rng = np.random.default_rng(8)
    lens = [120, 300, 75, 1024]
    df = pd.DataFrame({"id": np.repeat(np.arange(4), lens), "time": np.concatenate([np.arange(m) for m in lens]),
                       "value": np.concatenate([np.cumsum(rng.standard_normal(m)) for m in lens])})
    params = {"mean": None,
              "augmented_dickey_fuller": [{"attr": "teststat", "autolag": "BIC"}, {"attr": "pvalue", "autolag": "AIC"},
                                          {"attr": "usedlag", "autolag": None}, {"attr": "teststat", "autolag": "AIC"},
                                          {"attr": "teststat", "autolag": "bogus"}, {"attr": "usedlag", "autolag": "t-stat"}],
              "median": None}
    got = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params, device=0)
    values = df["value"].to_numpy()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    onames, want = oracle_engine(params, values, offsets)
    assert list(got.columns) == onames
    bad = compare(onames, got.to_numpy(), want, [values[offsets[i]:offsets[i + 1]] for i in range(4)])
    assert not bad, bad
