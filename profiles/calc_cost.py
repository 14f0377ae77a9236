#!/usr/bin/env python
"""Per-calculator device cost: one plan per calculator of ComprehensiveFCParameters, timed with HIP events
(tsfa_plan_set_profiling) on N series x L samples resident in HBM.  Prints a markdown table sorted by cost.

    python profiles/calc_cost.py [--n-series 20000] [--length 1024]
"""
import argparse
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-series", type=int, default=20000)
    ap.add_argument("--length", type=int, default=1024)
    ap.add_argument("--only", default="")
    ap.add_argument("--together", action="store_true", help="one plan holding all --only calculators")
    args = ap.parse_args()
    import torch
    from tsfresh_amd import _native
    from tsfresh_amd.feature_extraction import settings
    from tsfresh_amd.feature_extraction.plan import compile_fc_parameters

    dev = torch.device("cuda", 0)
    n, L = args.n_series, args.length
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    values = torch.randn(n * L, device=dev, dtype=torch.float32, generator=gen)
    offsets = torch.arange(0, (n + 1) * L, L, device=dev, dtype=torch.int64)
    full = settings.ComprehensiveFCParameters()
    rows = []
    stream = torch.cuda.current_stream(dev).cuda_stream
    items = list(full.items())
    if args.together:
        keys = args.only.split(",")
        items = [("+".join(keys), None)]
    for key, plist in items:
        if args.only and not args.together and key not in args.only.split(","):
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fplan = compile_fc_parameters({k: full[k] for k in key.split("+")} if args.together else {key: plist})
        if len(fplan) == 0:
            continue
        plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0)
        out = torch.empty((n, len(fplan)), device=dev, dtype=torch.float64)
        plan.extract_device(values.data_ptr(), _native.TSFA_F32, offsets.data_ptr(), n, out.data_ptr(), len(fplan), stream)
        plan.set_profiling(True)
        tot = {}
        reps = 2
        for _ in range(reps):
            plan.extract_device(values.data_ptr(), _native.TSFA_F32, offsets.data_ptr(), n, out.data_ptr(), len(fplan), stream)
            for nm, ms in plan.last_timings():
                tot[nm] = tot.get(nm, 0.0) + ms / reps
        plan.close()
        rows.append((sum(tot.values()), key, len(fplan), "+".join("%s" % k for k in tot)))
    rows.sort(reverse=True)
    print("| calculator | cols | kernel | ms / %d series | us / series |" % n)
    print("|---|---|---|---|---|")
    for ms, key, ncol, kn in rows:
        print("| %s | %d | %s | %.3f | %.3f |" % (key, ncol, kn, ms, 1e3 * ms / n))
    print("| TOTAL | | | %.3f | |" % sum(r[0] for r in rows))


if __name__ == "__main__":
    main()
