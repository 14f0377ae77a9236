"""The C-ABI library loads and exports every symbol include/tsfresh_amd.h declares; without a GPU it refuses to
compute instead of falling back to the CPU."""
import ctypes
import os
import re

import pytest

from tsfresh_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tsfresh_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsfa_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    declared = _declared_symbols()
    assert set(declared) == set(_native.EXPORTS), set(declared) ^ set(_native.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_version_and_registry():
    lib = _native.load()
    assert lib.tsfa_version() == 100
    from tsfresh_amd.feature_extraction.registry import CALCULATORS
    n = lib.tsfa_calc_count()
    names = {lib.tsfa_calc_name(i).decode() for i in range(n)}
    native = {k for k, c in CALCULATORS.items() if c.native}
    assert names == native, names ^ native
    for name in names:
        assert lib.tsfa_calc_name(lib.tsfa_calc_id(name.encode())).decode() == name
    assert lib.tsfa_calc_id(b"no_such_calculator") == -1


def test_no_gpu_means_error_not_cpu_fallback():
    if _native.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_native.NativeError) as ei:
        _native.Plan([(_native.calc_id("mean"), (0, 0, 0, 0))])
    assert ei.value.code == _native.TSFA_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_extract_features_fails_loudly_without_gpu():
    if _native.device_count() > 0:
        pytest.skip("a HIP device is present")
    import numpy as np
    import pandas as pd
    from tsfresh_amd import MinimalFCParameters, extract_features
    df = pd.DataFrame({"id": [1, 1, 2, 2], "t": [0, 1, 0, 1], "v": np.arange(4.0)})
    with pytest.raises(_native.NativeError):
        extract_features(df, column_id="id", column_sort="t", default_fc_parameters=MinimalFCParameters())
