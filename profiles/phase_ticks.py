#!/usr/bin/env python
"""Per-phase latency breakdown of the kernels (diagnostics build: `make -C tsfresh_amd/csrc ticks`).

Thread 0 of every workgroup accumulates shader-clock cycles per phase id (TSFA_TICK in the fam_*.h sources); this
script runs one extraction of the bench workload with libtsfresh_amd_ticks.so and prints cycles per series per phase.
    TSFA_LIB=tsfresh_amd/libtsfresh_amd_ticks.so python profiles/phase_ticks.py [--n-series 20000] [--length 1024]
"""
import argparse
import ctypes
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TSFA_LIB", os.path.join(ROOT, "tsfresh_amd", "libtsfresh_amd_ticks.so"))

NAMED = {213: "basic: count pass (all count-type columns)", 214: "basic: sum pass (all sum-type columns)", 220: "sort/change_quantiles: corridor edges (quantiles)", 221: "sort/change_quantiles: pass 1",
         222: "sort/change_quantiles: reduce 12", 223: "sort/change_quantiles: pass 2", 224: "sort/change_quantiles: reduce 8",
         230: "perm: window codes", 231: "perm: logarithm table + clear", 232: "perm: histogram pass (all dimensions)",
         233: "perm: windows by pattern count (G)", 234: "perm: terms (first wavefront)", 235: "perm: block sum",
         236: "sort/permutation_entropy one by one: window codes", 237: "sort/permutation_entropy one by one: logarithm table", 238: "sort/permutation_entropy one by one: histogram passes + sums", 239: "sort/permutation_entropy one by one: block sum",
         225: "basic/number_peaks: near pass (L/R up to 10)", 226: "basic/number_peaks: far candidates", 210: "basic: spec fetch (all columns)", 211: "basic: column bodies (all columns)",
         212: "basic: output stores (all columns)", 200: "basic/agg_linear_trend: chunk aggregates", 201: "basic/agg_linear_trend: regression sums",
         202: "basic/agg_linear_trend: linregress tails (lane = regression)", 100: "basic: stage + stats", 104: "sort: stage + bitonic sort", 120: "ar: mean / demean / var", 121: "ar: spec scan + autocovariances", 122: "ar: Levinson-Durbin (pacf)",
         123: "ar/adf: lag products + normal matrix", 124: "ar/adf: Cholesky + nested AIC", 125: "ar/adf: final regression",
         130: "entropy: std + sentinels", 131: "entropy: template sort + refs", 132: "entropy: group setup",
         133: "entropy: sweep group 0 (incl. totals)", 134: "entropy: sweep groups 1+ (incl. totals)",
         136: "entropy: pair / bit sweep (thread 0's wave)", 137: "entropy: wait for the other waves",
         138: "entropy bits: ranges to registers", 139: "entropy bits: table build",
         151: "cwtpeaks: phase A (CWT rows, maxima)", 152: "cwtpeaks: phase B (ridge lines)",
         153: "cwtpeaks: phase C (SNR filter)", 154: "cwtpeaks: phase C argsort of row 0 (long series)",
         155: "cwtpeaks: phase C order walk (long series)", 156: "cwtpeaks: phase C signal / noise (long series)", 170: "spectral: Welch periodogram", 171: "spectral: Welch columns", 172: "spectral: full-length rfft", 173: "spectral: |X| moments + fft columns", 160: "seq: min/max", 161: "seq: edges + table clear",
         162: "seq: binning", 163: "seq: parse (thread 0's chain)", 164: "seq: wait for the other chains"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-series", type=int, default=20000)
    ap.add_argument("--length", type=int, default=1024)
    ap.add_argument("--params", default="comprehensive")
    ap.add_argument("--ragged", default="", help="LO:HI -> lengths uniform on [LO, HI]")
    args = ap.parse_args()
    import torch
    from tsfresh_amd import _native
    from tsfresh_amd.feature_extraction import settings
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
    lib = _native.load()
    lib.tsfa_debug_ticks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    dev = torch.device("cuda", 0)
    n, L = args.n_series, args.length
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    if args.ragged:
        lo, hi = (int(t) for t in args.ragged.split(":"))
        lens = torch.randint(lo, hi + 1, (n,), device=dev, generator=gen, dtype=torch.int64)
        offsets = torch.zeros(n + 1, device=dev, dtype=torch.int64)
        offsets[1:] = torch.cumsum(lens, 0)
        values = torch.randn(int(offsets[-1].item()), device=dev, dtype=torch.float32, generator=gen)
    else:
        values = torch.randn(n * L, device=dev, dtype=torch.float32, generator=gen)
        offsets = torch.arange(0, (n + 1) * L, L, device=dev, dtype=torch.int64)
    cls = {"comprehensive": settings.ComprehensiveFCParameters, "efficient": settings.EfficientFCParameters,
           "minimal": settings.MinimalFCParameters}[args.params]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fplan = compile_fc_parameters(cls())
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0)
    out = torch.empty((n, len(fplan)), device=dev, dtype=torch.float64)
    run = lambda: plan.extract_device(values.data_ptr(), _native.TSFA_F32, offsets.data_ptr(), n, out.data_ptr(), len(fplan), None)
    run()
    buf = (ctypes.c_ulonglong * 256)()
    lib.tsfa_debug_ticks(buf, 256, 1)
    run()
    lib.tsfa_debug_ticks(buf, 256, 0)
    rows = []
    for i in range(256):
        if buf[i]:
            name = NAMED.get(i) or ("calc: " + (lib.tsfa_calc_name(i) or b"?").decode())
            rows.append((buf[i] / n, name))
    rows.sort(reverse=True)
    print("| phase | cycles / series (thread 0 of the workgroup) |")
    print("|---|---|")
    for c, name in rows:
        print("| %s | %.0f |" % (name, c))


if __name__ == "__main__":
    main()
