import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from engines import hip_engine
rng = np.random.default_rng(31)
lens = [257, 258, 300, 301, 510, 511, 513, 514, 1022, 1023, 1025, 1026, 1281, 1500, 2046, 2047, 2049, 2050, 3001, 4094, 4095, 4097, 4098, 6000, 6001, 8190, 8191, 8193, 8194, 12001, 16382, 16385, 20000, 32766, 32767]
series = [(np.cumsum(rng.standard_normal(n)) if i % 3 == 0 else rng.standard_normal(n) + (5.0 if i % 3 == 1 else 0.0)).astype(np.float32) for i, n in enumerate(lens)]
values = np.concatenate(series); offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
params = {"fft_coefficient": [{"attr": a, "coeff": k} for a in ("real", "imag", "abs", "angle") for k in (0, 1, 2, 5, 33, 99)],
          "fft_aggregated": [{"aggtype": t} for t in ("centroid", "variance", "skew", "kurtosis")]}
os.environ["TSFA_BLUESTEIN_MIN"] = "257"
n1, a = hip_engine(params, values, offsets)
n2, b = hip_engine(params, values, offsets)
print("repeat equal:", np.array_equal(a, b, equal_nan=True))
os.environ["TSFA_GSCRATCH_SLOTS"] = "1"
n3, c = hip_engine(params, values, offsets)
os.environ["TSFA_GSCRATCH_SLOTS"] = "3"
n4, d = hip_engine(params, values, offsets)
for nm, x in (("1 slot", c), ("3 slots", d), ("repeat", b)):
    rows = np.nonzero(~((a == x) | (np.isnan(a) & np.isnan(x))).all(axis=1))[0]
    print(nm, "differing rows:", [(int(r), lens[r], float(np.nanmax(np.abs(a[r] - x[r]) / (np.abs(a[r]) + 1e-300))), [n1[j] for j in np.nonzero(a[r] != x[r])[0]][:3]) for r in rows])
