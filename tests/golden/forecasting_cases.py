"""Inputs of the make_forecasting_frame golden cases."""
import numpy as np
import pandas as pd


def forecasting_cases():
    rng = np.random.default_rng(2)
    return [
        ("list_shift1", [1, 2, 3, 4], {"kind": "test", "max_timeshift": 1, "rolling_direction": 1}),
        ("range_shift2", range(6), {"kind": "test", "max_timeshift": 2, "rolling_direction": 1}),
        ("series_dates", pd.Series(data=[1.5, 2.5, 3.5, 4.5, 0.5], index=pd.date_range("2011-01-01", periods=5, freq="h")),
         {"kind": "test", "max_timeshift": 3, "rolling_direction": 1}),
        ("random_min_shift", list(rng.standard_normal(9)), {"kind": "k", "max_timeshift": 4, "rolling_direction": 1,
                                                           "min_timeshift": 1}),
    ]
