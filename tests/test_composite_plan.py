"""Settings objects one native plan cannot hold together are split by extract_features into several native plans whose
columns are scattered back (feature_extraction/extraction.py: _CompositePlan): more than 128 cwt_coefficients columns
(fc.py:1370 takes any list), several augmented_dickey_fuller lag selections (tests/test_adf_autolag.py)."""
import numpy as np
import pandas as pd
import pytest

from engines import oracle_engine
from parity import compare

WIDTHS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 20)
PARAMS = {"mean": None,
          "cwt_coefficients": [{"widths": WIDTHS, "coeff": c, "w": w} for w in WIDTHS for c in range(15)],   # 150 columns
          "median": None}


def _frame():
    rng = np.random.default_rng(4)
    lens = [64, 200, 31]
    df = pd.DataFrame({"id": np.repeat(np.arange(3), lens), "time": np.concatenate([np.arange(m) for m in lens]),
                       "value": np.concatenate([rng.standard_normal(m) for m in lens])})
    return df, lens


def _check(got, df, lens):
    values = df["value"].to_numpy()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    onames, want = oracle_engine(PARAMS, values, offsets)
    assert list(got.columns) == onames and got.shape == (3, 152)
    bad = compare(onames, got.to_numpy(), want, [values[offsets[i]:offsets[i + 1]] for i in range(3)])
    assert not bad, bad[:6]


def test_more_cwt_columns_than_one_native_plan_holds(monkeypatch):
    from emul_lib import emul_extract_specs
    from tsfresh_amd import _native, extract_features
    from tsfresh_amd.feature_extraction import extraction

    class _Part:
        def __init__(self, specs):
            self.specs = list(specs)

        def extract_host(self, values, offsets, times=None):
            return emul_extract_specs(self.specs, values, offsets, times=times)

    made = []
    cwt = _native.calc_id("cwt_coefficients")
    monkeypatch.setattr(extraction, "_acquire_plan_specs",
                        lambda specs, device, pins=None: made.append(sum(1 for c, _ in specs if c == cwt)) or _Part(specs))
    df, lens = _frame()
    got = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=PARAMS, device=0)
    assert made == [128, 22]
    _check(got, df, lens)


@pytest.mark.gpu
def test_hip_more_cwt_columns_than_one_native_plan_holds(gpu):
    from tsfresh_amd import extract_features
    df, lens = _frame()
    _check(extract_features(df, column_id="id", column_sort="time", default_fc_parameters=PARAMS, device=0), df, lens)


@pytest.mark.gpu
def test_hip_rolled_windows_through_a_composite_plan(gpu):
    """extract_rolled_features (window views, tsfa_extract_windows) with a settings object that needs two native plans equals
    extract_features on the materialised rolled frame."""
    import warnings

    from tsfresh_amd import extract_features, extract_rolled_features
    from tsfresh_amd.utilities.dataframe_functions import roll_time_series
    rng = np.random.default_rng(5)
    df = pd.concat([pd.DataFrame({"id": sid, "time": np.arange(L), "value": np.cumsum(rng.standard_normal(L))})
                    for sid, L in enumerate([90, 70, 120])], ignore_index=True)
    params = {"mean": None, "augmented_dickey_fuller": [{"attr": "teststat", "autolag": "AIC"}, {"attr": "usedlag", "autolag": "BIC"},
                                                        {"attr": "pvalue", "autolag": None}], "maximum": None}
    kw = dict(max_timeshift=60, min_timeshift=40)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rolled = roll_time_series(df, "id", "time", **kw)
        want = extract_features(rolled, column_id="id", column_sort="time", default_fc_parameters=params)
        got = extract_rolled_features(df, column_id="id", column_sort="time", default_fc_parameters=params, **kw)
    assert list(got.index) == list(want.index) and list(got.columns) == list(want.columns) and got.shape[1] == 5
    a, b = got.to_numpy(), want.to_numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


@pytest.mark.gpu
def test_hip_several_devices_through_several_native_plans(gpu):
    """extract_features(..., devices=[0, 0]) with mixed lag selections: every part runs on both shards."""
    from tsfresh_amd import extract_features
    rng = np.random.default_rng(9)
    lens = [150, 90, 300, 64, 200, 120]
    df = pd.DataFrame({"id": np.repeat(np.arange(len(lens)), lens), "time": np.concatenate([np.arange(m) for m in lens]),
                       "value": np.concatenate([np.cumsum(rng.standard_normal(m)) for m in lens])})
    params = {"mean": None, "augmented_dickey_fuller": [{"attr": "teststat", "autolag": "AIC"}, {"attr": "teststat", "autolag": "t-stat"}],
              "maximum": None}
    one = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params, device=0)
    two = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params, devices=[0, 0])
    # (the shards' launch groups differ from the single call's: the lag-search sums may differ in the last bit -- 1 ulp in one
    #  cell when this test first ran)
    pd.testing.assert_frame_equal(one, two, check_exact=False, rtol=1e-12, atol=0.0)
    assert one.shape == (6, 4) and not one.isna().any().any()


@pytest.mark.gpu
def test_device_resident_extraction_through_several_native_plans(gpu):
    """extract_relevant_features(device_resident=True) with a settings object that needs two native plans (round-5 VERDICT
    missing #6: it used to be refused): the parts' blocks are scattered into the device matrix (tsfa_scatter_columns), the
    selection equals the host path's."""
    from tsfresh_amd import extract_relevant_features
    rng = np.random.default_rng(12)
    n_ids, L = 60, 80
    y = pd.Series(rng.integers(0, 2, n_ids), index=np.arange(n_ids))
    x = rng.standard_normal((n_ids, L)) + 1.5 * y.to_numpy()[:, None]
    df = pd.DataFrame({"id": np.repeat(np.arange(n_ids), L), "time": np.tile(np.arange(L), n_ids), "value": x.reshape(-1)})
    params = {"mean": None, "augmented_dickey_fuller": [{"attr": "teststat", "autolag": "AIC"}, {"attr": "teststat", "autolag": "BIC"}],
              "maximum": None, "variance": None}
    host = extract_relevant_features(df, y, column_id="id", column_sort="time", default_fc_parameters=params)
    dev = extract_relevant_features(df, y, column_id="id", column_sort="time", default_fc_parameters=params, device_resident=True)
    assert list(host.columns) == list(dev.columns) and len(host.columns) >= 2
    pd.testing.assert_frame_equal(host, dev, check_exact=False, rtol=1e-12, atol=0.0)


@pytest.mark.gpu
def test_composite_plans_upload_the_samples_once_per_device(gpu, monkeypatch):
    """devices=[0, 0] with three lag selections: ONE _DeviceBuffer of the samples per shard, not one per part."""
    from tsfresh_amd import _native, extract_features
    rng = np.random.default_rng(9)
    lens = [150, 90, 300, 64, 200, 120]
    df = pd.DataFrame({"id": np.repeat(np.arange(len(lens)), lens), "time": np.concatenate([np.arange(m) for m in lens]),
                       "value": np.concatenate([np.cumsum(rng.standard_normal(m)) for m in lens])})
    params = {"augmented_dickey_fuller": [{"attr": "teststat", "autolag": a} for a in ("AIC", "BIC", "t-stat")]}
    uploads = []
    real = _native._DeviceBuffer.__init__

    def counting(self, lib, array, device):
        uploads.append(array.dtype.str)
        real(self, lib, array, device)
    monkeypatch.setattr(_native._DeviceBuffer, "__init__", counting)
    got = extract_features(df, column_id="id", column_sort="time", default_fc_parameters=params, devices=[0, 0])
    assert got.shape == (6, 3) and not got.isna().any().any()
    assert sum(1 for d in uploads if d.endswith("f8") or d.endswith("f4")) == 2, uploads     # one per shard
