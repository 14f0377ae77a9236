import sys, warnings
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np
from engines import hip_engine
warnings.simplefilter("ignore")
rng = np.random.default_rng(3)
series = [rng.standard_normal(300), np.cumsum(rng.standard_normal(500))]
values = np.concatenate(series); offsets = np.concatenate([[0], np.cumsum([len(s) for s in series])])
for ns in ((1,2,3,4,5,6,8,12,16), (1,2,3,4,5,6,8,12,16,17), (17,), (30, 17, 3)):
    n, m = hip_engine({"number_cwt_peaks": [{"n": k} for k in ns]}, values, offsets)
    print(ns, m.tolist())
