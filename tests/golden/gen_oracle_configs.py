"""tests/golden/oracle_configs.npz: oracle values of ~800 rows of the BASELINE config batches (tests/config_inputs.py).
    python tests/golden/gen_oracle_configs.py        (~5 minutes on 8 cores)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    import config_inputs
    from engines import oracle_engine_parallel
    out = {}
    for key, build in config_inputs.CONFIGS.items():
        pname, series, rows = build()
        chosen = [np.asarray(series[i], dtype=np.float64) for i in rows]
        values = np.concatenate(chosen)
        offsets = np.concatenate([[0], np.cumsum([len(c) for c in chosen])]).astype(np.int64)
        names, want = oracle_engine_parallel(config_inputs.parameters(pname), values, offsets, workers=8)
        out[key + "_names"] = np.array(names)
        out[key + "_rows"] = np.array(rows)
        out[key + "_matrix"] = want
        print(key, want.shape, flush=True)
    np.savez_compressed(os.path.join(HERE, "oracle_configs.npz"), **out)


if __name__ == "__main__":
    main()
