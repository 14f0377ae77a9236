import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.stats import rankdata
from tsfresh_amd import _native
rng = np.random.default_rng(5)
for n, m, C in [(7, 3, 2), (100, 9, 2), (3000, 17, 3), (5000, 5, 2), (20000, 40, 4), (100000, 64, 2)]:
    X = rng.standard_normal((n, m))
    X[:, 0] = np.round(X[:, 0], 1)            # heavy ties
    if m > 1: X[:, 1] = (X[:, 1] > 0.3) * 2.5  # binary
    if m > 2: X[:, 2] = 4.0                    # constant
    if m > 3: X[:, 3] = rng.integers(0, 3, n)  # three values
    y = rng.integers(0, C, n).astype(np.int32)
    t0 = time.perf_counter()
    nu, lo, hi, tie, rs, hc = _native.relevance_classes(X, y, C)
    dt = time.perf_counter() - t0
    for c in range(m):
        r = rankdata(X[:, c])
        u, cnt = np.unique(X[:, c], return_counts=True)
        assert nu[c] == len(u), (n, c, nu[c], len(u))
        assert lo[c] == u[0] and hi[c] == u[-1]
        assert tie[c] == float(np.sum(cnt.astype(float) ** 3 - cnt)), (tie[c], np.sum(cnt.astype(float) ** 3 - cnt))
        for k in range(C):
            assert rs[c, k] == r[y == k].sum(), (n, c, k, rs[c, k], r[y == k].sum())
            assert hc[c, k] == np.sum((X[:, c] == u[-1]) & (y == k))
    print("ok", n, m, C, "%.4f s" % dt)
