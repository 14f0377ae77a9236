"""Input frames of the roll_time_series golden (shared by gen_golden_roll.py and tests/test_roll.py)."""
import numpy as np
import pandas as pd


def roll_cases():
    """-> list[(name, DataFrame, kwargs)]"""
    rng = np.random.default_rng(7)
    a = pd.DataFrame({"id": [1] * 5 + [2] * 3 + [3] * 1, "time": [0, 1, 2, 3, 4, 0, 1, 2, 0],
                      "x": np.arange(9, dtype=float), "y": rng.standard_normal(9).round(3)})
    shuffled = a.sample(frac=1.0, random_state=3).reset_index(drop=True)
    long = pd.DataFrame({"id": ["a"] * 4 + ["b"] * 4 + ["a"] * 3, "kind": ["u"] * 8 + ["v"] * 3,
                         "time": [0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2], "value": np.arange(11, dtype=float)})
    other_id = a.rename(columns={"id": "sensor"})
    return [
        ("plain", a, dict(column_id="id", column_sort="time")),
        ("max2", a, dict(column_id="id", column_sort="time", max_timeshift=2)),
        ("max2_min1", a, dict(column_id="id", column_sort="time", max_timeshift=2, min_timeshift=1)),
        ("negative", a, dict(column_id="id", column_sort="time", rolling_direction=-1)),
        ("negative_max1", a, dict(column_id="id", column_sort="time", rolling_direction=-1, max_timeshift=1)),
        ("step2", a, dict(column_id="id", column_sort="time", rolling_direction=2)),
        ("step_minus2_max3", a, dict(column_id="id", column_sort="time", rolling_direction=-2, max_timeshift=3)),
        ("shuffled", shuffled, dict(column_id="id", column_sort="time", max_timeshift=3)),
        ("no_sort", a[["id", "x"]], dict(column_id="id", max_timeshift=2)),
        ("kind", long, dict(column_id="id", column_sort="time", column_kind="kind", max_timeshift=2)),
        ("other_id_name", other_id, dict(column_id="sensor", column_sort="time", max_timeshift=1)),
    ]
