"""Diagnostic (run on the GPU box): per-feature mismatch summary of the HIP path vs golden + oracle."""
import os
import sys
import warnings
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
warnings.simplefilter("ignore")
from engines import hip_engine, oracle_engine  # noqa: E402
from parity import compare, feature_of  # noqa: E402
from tsfresh_amd.feature_extraction import settings  # noqa: E402


def summarize(tag, bad):
    by = defaultdict(list)
    for b in bad:
        by[b.split(" ")[2].split("__")[1].rstrip(":")].append(b)
    print("== %s: %d mismatches in %d features" % (tag, len(bad), len(by)))
    for f, lst in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print("   %-45s %4d   e.g. %s" % (f, len(lst), lst[0][:230]))


def main():
    G = os.path.join(HERE, "golden")
    g1 = np.load(os.path.join(G, "ref_main.npz"))
    g2 = np.load(os.path.join(G, "ref_conda.npz"))
    names = list(g1["names"]) + list(g2["names"])
    want = np.concatenate([g1["matrix"], g2["matrix"]], axis=1)
    values, offsets = g1["values"], g1["offsets"]
    series = [values[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]
    gn, got = hip_engine(settings.ComprehensiveFCParameters(), values, offsets)
    got = got[:, [gn.index(n) for n in names]]
    summarize("golden(nt=64)", compare(names, got, want, series))

    rng = np.random.default_rng(11)
    lens = [3000, 2500, 4096, 100, 2049, 777]
    vals = np.concatenate([np.cumsum(rng.standard_normal(n)) if i % 2 else rng.standard_normal(n) for i, n in enumerate(lens)])
    offs = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    params = settings.EfficientFCParameters()
    params["approximate_entropy"] = [{"m": 2, "r": 0.3}, {"m": 2, "r": 0.7}]
    params["sample_entropy"] = None
    gn, got = hip_engine(params, vals, offs)
    on, ow = oracle_engine(params, vals, offs)
    got = got[:, [gn.index(n) for n in on]]
    summarize("long(nt=256)", compare(on, got, ow, [vals[offs[i]:offs[i + 1]] for i in range(len(lens))]))


if __name__ == "__main__":
    main()
