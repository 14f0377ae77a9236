import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import cases
from engines import hip_engine
from tsfresh_amd.feature_extraction import settings
for dtype in (np.float64, np.float32):
    rng = np.random.default_rng(45)
    n, L = 4096, 1024
    half = n // 2
    base = rng.standard_normal((half, L)).astype(dtype)
    base[1::2] = np.cumsum(base[1::2], axis=1)
    special = cases.config3_rows(dtype, L)
    where = [7 + 113 * k for k in range(len(special))]
    for w, row in zip(where, special):
        base[w] = row
    x = np.concatenate([base, base[::-1]])
    offsets = np.arange(n + 1, dtype=np.int64) * L
    params = settings.ComprehensiveFCParameters()
    for rep in range(2):
        names, got = hip_engine(params, x.reshape(-1), offsets)
        a, b = got[:half], got[half:][::-1]
        diff = ~((a == b) | (np.isnan(a) & np.isnan(b)))
        rows, cols = np.nonzero(diff)
        print(dtype.__name__, 'rep', rep, 'differing cells', diff.sum())
        for r, c in list(zip(rows, cols))[:12]:
            print('   row', r, 'special' if r in where else ('walk' if r % 2 else 'iid'), names[c], repr(a[r, c]), repr(b[r, c]))
