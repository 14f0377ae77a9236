"""Parameters away from the grids of ComprehensiveFCParameters (tests/golden/param_cases.py: 229 columns of 37
calculators -- what a from_columns / hand-written settings dict sends down the same kernels) against outputs of the REAL
reference (ref_main_sweep.npz / ref_conda_sweep.npz, `gen_golden_main.py --params sweep`, `gen_golden_conda.py --params
sweep`)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import goldens
from engines import emul_engine, oracle_engine
from param_cases import sweep_parameters


def test_fixture_has_every_column_of_the_sweep():
    import warnings
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fplan = compile_fc_parameters(sweep_parameters())
    g = goldens.load("sweep")
    assert sorted(g["names"]) == sorted("value__" + n for n in fplan.names)
    assert len(g["names"]) == 229


@pytest.mark.parametrize("engine", [oracle_engine, emul_engine], ids=["oracle", "emul"])
def test_engine_matches_the_reference_on_other_parameters(engine):
    bad, skipped, cells = goldens.check_engine(engine, "sweep", sweep_parameters())
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    assert len(skipped) <= 0.015 * cells, (len(skipped), cells)   # (71 of 5496: the set is made of the calculators the exclusions of tests/parity.py are about)


@pytest.mark.gpu
def test_hip_matches_the_reference_on_other_parameters(gpu):
    from engines import hip_engine
    bad, skipped, cells = goldens.check_engine(hip_engine, "sweep", sweep_parameters())
    assert not bad, "%d mismatches, first: %s" % (len(bad), bad[:10])
    assert len(skipped) <= 0.015 * cells, (len(skipped), cells)   # (71 of 5496: the set is made of the calculators the exclusions of tests/parity.py are about)
