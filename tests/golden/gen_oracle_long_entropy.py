"""tests/golden/oracle_beyond_65535_entropy.json: sample_entropy of the 70 001-sample iid series and of the 100 001-sample walk
of gen_oracle_long.py, evaluated by oracle/ (the reference's own loop: one O(n) numpy pass per template, fc.py:1729-1754) --
about two minutes per series here, so the gpu test compares against the stored values.  The reference's approximate_entropy
cannot be evaluated at these lengths at all (an n x n x m float64 array: 160 GB at 100 000 samples).
    python tests/golden/gen_oracle_long_entropy.py"""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def _one(i):
    import numpy as np
    from gen_oracle_long import series
    from engines import oracle_engine
    x = series()[i].astype(np.float64)
    names, want = oracle_engine({"sample_entropy": None}, x, np.array([0, len(x)], dtype=np.int64))
    return i, len(x), float(want[0, 0])


def main():
    with mp.get_context("spawn").Pool(2) as pool:
        res = pool.map(_one, [0, 1])
    doc = {"series_%d" % i: {"n": n, "sample_entropy": repr(v)} for i, n, v in res}
    json.dump(doc, open(os.path.join(HERE, "oracle_beyond_65535_entropy.json"), "w"), indent=1)
    print(doc)


if __name__ == "__main__":
    main()
