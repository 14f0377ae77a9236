#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_g
mkdir -p $O
timeout 900 python profiles/fuzz_parity.py 40 11 > $O/fuzz_gpu.log 2>&1; tail -3 $O/fuzz_gpu.log
timeout 600 python -m pytest tests/test_distributor.py tests/test_gpu_parity.py -m gpu -q -k "config5 or map_reduce or stream" 2>&1 | tail -3
# which family blows up on very long series?  200 x 16384, one family at a time, 100 s each
for fam in entropy cwt seq spectral ar sort basic; do
  timeout 100 python - <<PY 2>&1 | tail -1
import time, numpy as np, sys
sys.path.insert(0, ".")
from tsfresh_amd import _native
from tsfresh_amd.feature_extraction import settings
from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
full = settings.ComprehensiveFCParameters()
groups = {"entropy": ["sample_entropy", "approximate_entropy"], "cwt": ["number_cwt_peaks", "cwt_coefficients"], "seq": ["lempel_ziv_complexity"],
          "spectral": ["fft_coefficient", "fft_aggregated", "spkt_welch_density", "fourier_entropy"],
          "ar": ["agg_autocorrelation", "partial_autocorrelation", "ar_coefficient", "augmented_dickey_fuller"],
          "sort": ["median", "quantile", "change_quantiles", "permutation_entropy", "friedrich_coefficients", "max_langevin_fixed_point", "symmetry_looking"],
          "basic": ["c3", "cid_ce", "number_peaks", "benford_correlation", "agg_linear_trend", "index_mass_quantile", "binned_entropy"]}
params = {k: full[k] for k in groups["$fam"]}
import warnings; warnings.simplefilter("ignore")
fplan = compile_fc_parameters(params)
plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0)
rng = np.random.default_rng(1)
for n, L in ((200, 16384), (200, 65535)):
    x = rng.standard_normal(n * L).astype(np.float32)
    off = np.arange(n + 1, dtype=np.int64) * L
    plan.extract_host(x, off)
    t = time.perf_counter(); plan.extract_host(x, off); dt = time.perf_counter() - t
    print("$fam", n, "x", L, "%.1f ms" % (1e3 * dt), "(%.2f ms / series)" % (1e3 * dt / n), end=" | ")
print()
PY
done > $O/long_families.txt 2>&1
cat $O/long_families.txt
