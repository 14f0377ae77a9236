"""GPUDistributor: the `map_reduce` a reference `extract_features(distributor=...)` would
call, fed with the reference's chunk shape `(id, kind, pd.Series)`."""
from collections import namedtuple

import numpy as np
import pandas as pd
import pytest

from tsfresh_amd.utilities.distribution import GPUDistributor

Timeseries = namedtuple("Timeseries", ["id", "kind", "data"])  # tsfresh/feature_extraction/data.py:53


def _chunks():
    rng = np.random.default_rng(4)
    out = []
    for sid in (7, 3, 5):
        for kind in ("a", "b"):
            out.append(Timeseries(sid, kind, pd.Series(rng.standard_normal(50 + sid))))
    return out


def test_rejects_foreign_distributors_and_accepts_its_own():
    from tsfresh_amd import extract_features
    df = pd.DataFrame({"id": [1, 1], "time": [0, 1], "x": [1.0, 2.0]})
    with pytest.raises(ValueError, match="not an DistributorBaseClass"):
        extract_features(df, column_id="id", column_sort="time", distributor=object())
    d = GPUDistributor(device=0)
    assert d.device == 0 and d.close() is None


@pytest.mark.gpu
def test_map_reduce_returns_the_reference_tuples(gpu):
    from tsfresh_amd import MinimalFCParameters, extract_features
    chunks = _chunks()
    params = MinimalFCParameters()
    tuples = GPUDistributor().map_reduce(None, data=iter(chunks),
                                         function_kwargs={"default_fc_parameters": params,
                                                          "kind_to_fc_parameters": {"b": {"maximum": None}}})
    assert all(len(t) == 3 for t in tuples)
    got = {(t[0], t[1]): t[2] for t in tuples}
    assert len(got) == 3 * (10 + 1)
    for c in chunks:
        assert got[(c.id, c.kind + "__maximum")] == np.max(c.data.to_numpy())
        if c.kind == "a":
            assert got[(c.id, "a__sum_values")] == np.sum(c.data.to_numpy())
    # same numbers as the DataFrame route
    df = pd.concat([pd.DataFrame({"id": c.id, "kind": c.kind, "t": np.arange(len(c.data)), "v": c.data.to_numpy()})
                    for c in chunks])
    feats = extract_features(df, column_id="id", column_sort="t", column_kind="kind", column_value="v",
                             default_fc_parameters=params, kind_to_fc_parameters={"b": {"maximum": None}},
                             distributor=GPUDistributor(device=0))
    for (sid, name), v in got.items():
        assert feats.loc[sid, name] == v
