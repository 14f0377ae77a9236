#!/usr/bin/env python
"""number_cwt_peaks (n = 1, 5) alone on the headline shape: the kernel's time from HIP events, for counter passes around
k_cwtpeaks in both forms of phase A (TSFA_NO_CWT_MFMA=0 / 1).   python profiles/lab/cwtpeaks_only.py [n_series] [length]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tsfresh_amd import _native  # noqa: E402
from tsfresh_amd.feature_extraction.plan import compile_fc_parameters  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(42)
values = torch.randn(n * L, device=dev, dtype=torch.float32, generator=gen)
offsets = torch.arange(0, (n + 1) * L, L, device=dev, dtype=torch.int64)
fplan = compile_fc_parameters({"number_cwt_peaks": [{"n": 1}, {"n": 5}]})
plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0)
out = torch.empty((n, len(fplan)), device=dev, dtype=torch.float64)
run = lambda: plan.extract_device(values.data_ptr(), _native.TSFA_F32, offsets.data_ptr(), n, out.data_ptr(), len(fplan), None)  # noqa: E731
run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    run()
torch.cuda.synchronize()
print('{"n_series": %d, "length": %d, "no_mfma": "%s", "ms_per_call": %.3f, "checksum": %.1f}' % (
    n, L, os.environ.get("TSFA_NO_CWT_MFMA", "0"), (time.perf_counter() - t0) / 5 * 1e3, float(out.sum().item())))
