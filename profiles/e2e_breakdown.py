"""Where a DataFrame -> DataFrame call spends its time (pack / plan / tsfa_extract / assemble), 20 000 x 1024."""
import time, warnings, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from tsfresh_amd import ComprehensiveFCParameters, _native, extract_features
from tsfresh_amd.feature_extraction.data import pack_timeseries
from tsfresh_amd.feature_extraction.plan import compile_fc_parameters
from tsfresh_amd.feature_extraction import extraction as E
n, L = 20000, 1024
rng = np.random.default_rng(42)
x = rng.standard_normal((n, L), dtype=np.float32)
df = pd.DataFrame({"id": np.repeat(np.arange(n), L), "time": np.tile(np.arange(L), n), "value": x.reshape(-1)})
warnings.simplefilter("ignore")
for rep in range(3):
    t0 = time.perf_counter(); packed, idd, hd = pack_timeseries(df, column_id="id", column_sort="time"); t1 = time.perf_counter()
    fplan = compile_fc_parameters(ComprehensiveFCParameters()); t2 = time.perf_counter()
    plan = _native.Plan(fplan.native_specs(_native.calc_id), device=0); t3 = time.perf_counter()
    m = plan.extract_host(packed[0].values, packed[0].offsets); t4 = time.perf_counter()
    plan.close(); t5 = time.perf_counter()
    out = E._assemble([(packed[0], ["value__" + k for k in fplan.names], m)], idd, True, None); t6 = time.perf_counter()
    print("pack %.4f compile %.4f plan %.4f extract_host %.4f close %.4f assemble %.4f total %.4f" % (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t6-t5, t6-t0))
t0 = time.perf_counter(); f = extract_features(df, column_id="id", column_sort="time"); print("extract_features %.4f" % (time.perf_counter() - t0))
