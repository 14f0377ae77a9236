"""linear_trend_timewise (feature_calculators.py:2274): the calculator that regresses on the DatetimeIndex.

Golden: tests/golden/ref_timewise.npz, produced by the REAL reference's dispatcher (gen_golden_timewise.py).  The
timestamps travel as int64 ns; the tests rebuild the DataFrame, let the product's packer derive "hours since the
first stamp" exactly as the reference does, and compare oracle / emulation (CPU) and the HIP path (GPU)."""
import os
import warnings

import numpy as np
import pandas as pd
import pytest

from engines import emul_engine, hip_engine, oracle_engine
from parity import compare
from tsfresh_amd.feature_extraction.data import pack_timeseries

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAMS = {"linear_trend_timewise": [{"attr": a} for a in ["pvalue", "rvalue", "intercept", "slope", "stderr"]]}


def _golden_frame():
    g = np.load(os.path.join(G, "ref_timewise.npz"))
    values, stamps, offsets = g["values"], g["stamps_ns"], g["offsets"]
    ids = np.repeat(np.arange(len(offsets) - 1), np.diff(offsets))
    df = pd.DataFrame({"id": ids, "value": values}, index=pd.DatetimeIndex(stamps))
    return g, df


def _packed(df):
    packed, _, has_dt = pack_timeseries(df, column_id="id")
    assert has_dt and len(packed) == 1 and packed[0].times is not None
    return packed[0]


def test_packer_hours_match_the_reference_expression():
    g, df = _golden_frame()
    pk = _packed(df)
    for s in range(len(g["offsets"]) - 1):
        ix = df.index[g["offsets"][s]:g["offsets"][s + 1]]
        want = np.asarray((ix - ix[0]).total_seconds() / float(3600))  # fc.py:2294-2296
        assert np.array_equal(pk.times[pk.offsets[s]:pk.offsets[s + 1]], want)


@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_cpu_engines_match_reference_golden(engine):
    g, df = _golden_frame()
    pk = _packed(df)
    names, got = engine(PARAMS, pk.values, pk.offsets, times=pk.times)
    assert names == list(g["names"])
    series = [pk.values[pk.offsets[i]:pk.offsets[i + 1]] for i in range(pk.n_series)]
    bad = compare(names, got, g["matrix"], series)
    assert not bad, bad[:10]


def test_skipped_with_the_reference_warning_without_datetime_index():
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        plan = compile_fc_parameters(PARAMS, has_datetime_index=False)
    assert len(plan) == 0 and any("requires the data to have a index of type" in str(x.message) for x in w)
    assert len(compile_fc_parameters(PARAMS, has_datetime_index=True)) == 5


@pytest.mark.gpu
def test_hip_matches_reference_golden(gpu):
    g, df = _golden_frame()
    pk = _packed(df)
    names, got = hip_engine(PARAMS, pk.values, pk.offsets, times=pk.times)
    series = [pk.values[pk.offsets[i]:pk.offsets[i + 1]] for i in range(pk.n_series)]
    bad = compare(names, got, g["matrix"], series)
    assert not bad, bad[:10]


@pytest.mark.gpu
def test_extract_features_on_a_datetime_index(gpu):
    """DataFrame with a DatetimeIndex -> all 788 Comprehensive columns (783 + the five timewise attributes)."""
    from tsfresh_amd import ComprehensiveFCParameters, _native, extract_features
    g, df = _golden_frame()
    feats = extract_features(df.sample(frac=1.0, random_state=0).sort_index(kind="stable"), column_id="id",
                             default_fc_parameters=ComprehensiveFCParameters())
    assert feats.shape == (len(g["offsets"]) - 1, 788)
    got = feats[list(g["names"])].to_numpy()
    pk = _packed(df)
    series = [pk.values[pk.offsets[i]:pk.offsets[i + 1]] for i in range(pk.n_series)]
    assert not compare(list(g["names"]), got, g["matrix"], series)
    # the C-ABI refuses a timewise plan without times instead of silently skipping
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    plan = _native.Plan(compile_fc_parameters(PARAMS, has_datetime_index=True).native_specs(_native.calc_id))
    with pytest.raises(_native.NativeError):
        plan.extract_host(pk.values, pk.offsets)
    plan.close()
