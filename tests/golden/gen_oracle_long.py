"""tests/golden/oracle_beyond_65535.npz: EfficientFCParameters of four series beyond the old 65 535-sample cap (70 001,
100 001 and 200 000 float32 samples + a monotone series of 70 000), evaluated by oracle/ -- the restatement pinned against the real reference on the
ref_*.npz fixtures -- so that the gpu test does not spend two minutes of the GPU box's time in the oracle.
    python tests/golden/gen_oracle_long.py      (~3 minutes, three processes)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
LENS = [70001, 100001, 200000, 70000]


def series():
    rng = np.random.default_rng(65536)
    return [rng.standard_normal(LENS[0]).astype(np.float32),
            np.cumsum(rng.standard_normal(LENS[1])).astype(np.float32),
            (rng.standard_normal(LENS[2]) * np.linspace(0.5, 2.0, LENS[2]) + 3.0).astype(np.float32),
            # monotone: every ordinal pattern window is the SAME pattern (a count beyond 16 bits), every value unique
            (np.arange(LENS[3], dtype=np.float64) * 0.5 + np.sin(np.arange(LENS[3]) * 0.001) * 0.1).astype(np.float32)]


def main():
    from engines import oracle_engine_parallel
    from tsfresh_amd.feature_extraction.settings import EfficientFCParameters
    xs = series()
    values = np.concatenate(xs).astype(np.float64)
    offsets = np.concatenate([[0], np.cumsum(LENS)]).astype(np.int64)
    names, want = oracle_engine_parallel(EfficientFCParameters(), values, offsets, workers=4)
    np.savez_compressed(os.path.join(HERE, "oracle_beyond_65535.npz"), names=np.array(names), matrix=want, lens=np.array(LENS))
    print("wrote", len(names), "columns x", len(LENS), "series")


if __name__ == "__main__":
    main()
