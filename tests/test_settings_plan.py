"""Settings dictionaries and column names must equal the reference's (golden names recorded from the real
reference by tests/golden/gen_golden_main.py; reference tests: tests/units/feature_extraction/test_settings.py)."""
import os
import warnings

import numpy as np
import pytest

from tsfresh_amd.feature_extraction import settings
from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
from tsfresh_amd.feature_extraction.registry import UnsupportedFeature
from tsfresh_amd.utilities.string_manipulation import convert_to_output_format, get_config_from_string

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THIRD = ("cwt_coefficients", "agg_autocorrelation", "partial_autocorrelation", "augmented_dickey_fuller", "ar_coefficient")


@pytest.mark.parametrize("cls,n_keys,n_cols", [("ComprehensiveFCParameters", 75, 783), ("EfficientFCParameters", 73, 777),
                                               ("MinimalFCParameters", 10, 10)])
def test_settings_match_reference(cls, n_keys, n_cols):
    g = np.load(os.path.join(G, "ref_main.npz"))
    ours = getattr(settings, cls)()
    assert list(ours.keys()) == list(g["names_%s_keys" % cls])  # same calculators, same order
    assert len(ours) == n_keys
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plan = compile_fc_parameters(ours)
    assert len(plan) == n_cols
    ref_names = list(g["names_" + cls])  # without the five third-party calculators
    mine = ["value__" + n for n in plan.names if n.split("__")[0] not in THIRD]
    assert mine == ref_names


def test_mean_n_absolute_max_quirk():
    # settings.py:272-278 repeats the dict key, so only number_of_maxima=7 survives
    assert settings.ComprehensiveFCParameters()["mean_n_absolute_max"] == [{"number_of_maxima": 7}]


def test_column_name_round_trip():
    p = settings.ComprehensiveFCParameters()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plan = compile_fc_parameters(p)
    cols = ["value__" + n for n in plan.names]
    back = settings.from_columns(cols)["value"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plan2 = compile_fc_parameters(back)
    assert sorted(plan2.names) == sorted(plan.names)


def test_param_formatting():
    assert convert_to_output_format({"p1": '"a"', "p2": 1}) == 'p1_""a""__p2_1'
    assert convert_to_output_format({"b": 1, "a": "x"}) == 'a_"x"__b_1'
    assert get_config_from_string(["k", "f", "q_0.5", 'attr_"x"']) == {"q": 0.5, "attr": "x"}
    assert get_config_from_string(["k", "f"]) is None


def test_unsupported_features_raise_instead_of_falling_back():
    with pytest.raises(UnsupportedFeature):
        compile_fc_parameters({"matrix_profile": [{"threshold": 0.98, "feature": "min"}]})
    # served since round 6 (fam_general.h: gen_query_count): the query travels in the spec's tail to the plan's float64 pool
    fp = compile_fc_parameters({"query_similarity_count": [{"query": [1.0, 2.0, 3.0], "threshold": 0.5}]})
    assert fp.specs[0][1] == (0.5, 1.0, 0.0, 3.0, 1.0, 2.0, 3.0)
    compile_fc_parameters({"augmented_dickey_fuller": [{"attr": "teststat", "autolag": "BIC"}]})   # served since round 5
    with pytest.raises(AttributeError):
        compile_fc_parameters({"not_a_calculator": None})


def test_custom_callable_calculators_are_spliced_in_at_their_dict_position():
    """Callable keys of the FCParameters dict (extraction.py:340-343, docs/text/how_to_add_custom_feature.rst): evaluated
    per series on the host with the reference's call shapes (simple with / without parameters, combiner) and column
    names, between the native columns in dict order."""
    import numpy as np

    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters

    def spread(x):
        return float(np.max(x) - np.min(x))

    def above(x, t):
        return int(np.sum(x > t))

    def ends(x, param):
        return [("which_%s" % p["which"], x[0] if p["which"] == "first" else x[-1]) for p in param]
    ends.fctype = "combiner"

    fc = {"maximum": None, spread: None, "minimum": None, above: [{"t": 0.5}, {"t": -1}],
          ends: [{"which": "first"}, {"which": "last"}], "mean": None}
    fplan = compile_fc_parameters(fc)
    assert fplan.names == ["maximum", "minimum", "mean"] and len(fplan) == 6
    rows = [np.array([1.0, -2.0, 3.0]), np.array([0.25, 0.75])]
    native = np.array([[3.0, -2.0, 2 / 3], [0.75, 0.25, 0.5]])
    names, matrix = fplan.finish(native, lambda i: rows[i], 2)
    assert names == ["maximum", "spread", "minimum", "above__t_0.5", "above__t_-1", "ends__which_first", "ends__which_last", "mean"]
    assert matrix.tolist() == [[3.0, 5.0, -2.0, 2.0, 2.0, 1.0, 3.0, 2 / 3], [0.75, 0.5, 0.25, 1.0, 2.0, 0.25, 0.75, 0.5]]
    only_host = compile_fc_parameters({spread: None})
    names, matrix = only_host.finish(np.empty((2, 0)), lambda i: rows[i], 2)
    assert names == ["spread"] and matrix.tolist() == [[5.0], [0.5]]


def test_settings_pickle_with_callable_keys():
    """The reference's own test (tests/units/feature_extraction/test_settings.py:329-352): lambdas and nested functions as
    keys survive pickle because the keys travel cloudpickled."""
    import pickle

    from tsfresh_amd.feature_extraction.settings import ComprehensiveFCParameters, PickableSettings
    settings = PickableSettings()
    settings["test"] = 3
    settings[lambda x: x + 1] = None

    def f(x):
        return x - 2

    settings[f] = {"this": "is a test"}
    settings = pickle.loads(pickle.dumps(settings))
    assert "test" in settings and len(settings) == 3
    for key in settings:
        assert (not callable(key) or (key(3) == 4 and settings[key] is None)
                or (key(3) == 1 and settings[key] == {"this": "is a test"}))
    full = ComprehensiveFCParameters()
    back = pickle.loads(pickle.dumps(full))
    assert type(back) is ComprehensiveFCParameters and dict(back) == dict(full) and list(back) == list(full)
