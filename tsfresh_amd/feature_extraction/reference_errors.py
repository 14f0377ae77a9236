"""Exceptions of the reference that are a function of the DATA ("mirror the exception in the host
pre-check").

Two calculators of the reference do not return a number for a series that holds an infinite sample -- they raise, and
`extract_features` propagates the exception (found by running the real reference calculator by calculator on series
with +inf / -inf planted at several positions, `tests/golden/gen_golden_nonfinite.py` -> `ref_nonfinite.json`):

* `binned_entropy` (fc.py:1666-1698): `np.histogram` -> ``ValueError("autodetected range of [{min}, {max}] is not
  finite")`` whenever the minimum or the maximum of the series is infinite;
  and, under numpy >= 2.0, ``ValueError("Too many bins for data range. Cannot create {max_bins} finite-sized bins.")``
  when the range of the series is so few ulps wide that two edges of the linspace coincide (2^53 + {0, 2, 4});
* `ar_coefficient` (fc.py:1459-1510): `AutoReg(x, lags=k, trend="c")` -> statsmodels' ``MissingDataError("exog contains
  inf or nans")`` when a sample that enters the lag matrix (x[0] .. x[n-2]) is +inf and the series is long enough for
  the fit to be set up (n >= 2 k + 1; shorter series raise the ValueError that fc.py:1496 catches -> NaN).  -inf does
  not raise (statsmodels tests the column maxima, base/data.py `_handle_constant`): the columns are NaN, as here.

* `query_similarity_count` with a query of m >= 3 samples (fc.py:2513-2516): `stumpy.core.mass` /  `mass_absolute` check the
  window size first (core.check_window_size) -> ``ValueError("The window size must be less than or equal to {len(x)}")`` for
  a series shorter than the query.  (stumpy is not installed in the build image: the text is its published one, unpinned.)

The kernels never raise: `binned_entropy` writes NaN for a non-finite range (tsfa_common.h: blk_binned_entropy) and the
AR sums of such a series are NaN.  This module looks at those cells of the finished matrix -- a few columns, no pass over
the samples -- and only for rows that hold a NaN there goes back to the series to decide whether the reference would have
raised, with the reference's own message.  The first failing series in row order raises, and within a series the
calculator that comes first in the settings object, as in the reference's per-chunk loop (extraction.py:339-378).
"""
import numpy as np

try:  # pragma: no cover - statsmodels is not installed in the build image
    from statsmodels.tools.sm_exceptions import MissingDataError
except Exception:
    class MissingDataError(Exception):
        """Stand-in for statsmodels.tools.sm_exceptions.MissingDataError (same name, same base) where statsmodels is absent."""


def _nan_rows(matrix, cols):
    if not cols:
        return np.zeros(0, dtype=np.int64)
    sub = matrix[:, cols]
    return np.flatnonzero(np.isnan(sub).any(axis=1))


def _query_candidates(specs, starts, ends):
    qs = [(j, int(p[3])) for j, (name, p) in enumerate(specs) if name == "query_similarity_count" and len(p) > 3 and p[3] >= 3]
    if not qs:
        return []
    lengths = np.asarray(ends, dtype=np.int64) - np.asarray(starts, dtype=np.int64)
    out = []
    for j, m in qs:   # per series the reference evaluates the parameter sets in list order: the first too-long query raises
        rows = np.flatnonzero(lengths < m)
        if len(rows):
            r = int(rows[0])
            out.append((r, j, ValueError("The window size must be less than or equal to {}".format(int(lengths[r])))))
    return out


def check_query_lengths(specs, starts, ends):
    """query_similarity_count alone: raises for the first series that is shorter than a query of the plan."""
    c = _query_candidates(specs, starts, ends)
    if c:
        c.sort(key=lambda t: (t[0], t[1]))
        raise c[0][2]


def check_reference_data_errors(specs, matrix, values, starts, ends):
    """specs: [(calculator name, p)] aligned with the columns of `matrix` (FeaturePlan.specs); series i =
    values[starts[i]:ends[i]].  Raises what the reference raises for the first series it cannot evaluate; else returns."""
    if matrix.shape[0] == 0:
        return
    be_cols = [j for j, (name, _) in enumerate(specs) if name == "binned_entropy"]
    ar = [(j, int(p[1])) for j, (name, p) in enumerate(specs) if name == "ar_coefficient"]
    # a coefficient beyond the order is NaN for every series (fc.py:1503-1506): such a column says nothing about the fit
    ar_fit = [j for j, (name, p) in enumerate(specs) if name == "ar_coefficient" and 0 <= int(p[0]) <= int(p[1])]
    candidates = _query_candidates(specs, starts, ends)   # (row, first column of the calculator, exception)
    if not be_cols and not ar and not candidates:
        return
    for r in _nan_rows(matrix, be_cols):
        x = np.asarray(values[starts[r]:ends[r]], dtype=np.float64)
        lo, hi = x.min(), x.max()
        if not (np.isfinite(lo) and np.isfinite(hi)):
            candidates.append((int(r), be_cols[0], ValueError(
                "autodetected range of [{}, {}] is not finite".format(lo, hi))))
            break   # rows ascend: the first one decides for this calculator
        # numpy >= 2.0: a range of a few ulps cannot be cut into max_bins distinct edges (_histograms_impl.py:449-455)
        if lo == hi:
            lo, hi = lo - 0.5, hi + 0.5
        stuck = None
        for j in be_cols:
            if np.isnan(matrix[r, j]):
                bins = int(specs[j][1][0])
                with np.errstate(all="ignore"):
                    edges = np.linspace(lo, hi, bins + 1, endpoint=True, dtype=np.float64)
                if np.any(edges[:-1] >= edges[1:]):
                    stuck = (j, bins)
                    break
        if stuck is not None:
            candidates.append((int(r), stuck[0], ValueError(
                "Too many bins for data range. Cannot create {} finite-sized bins.".format(stuck[1]))))
            break
    if ar:
        lengths = np.asarray(ends, dtype=np.int64) - np.asarray(starts, dtype=np.int64)
        ks = sorted({k for _, k in ar})
        rows = _nan_rows(matrix, ar_fit) if ar_fit else np.arange(matrix.shape[0])
        # shorter series: AutoReg refuses them before it looks at the samples (ValueError, caught at fc.py:1496: NaN columns)
        rows = rows[lengths[rows] >= 2 * ks[0] + 1]
        if len(rows) > 64:   # many NaN rows (a batch of constant series ...): one pass over the samples instead of a loop
            cs = np.concatenate([[0], np.cumsum(np.isposinf(values), dtype=np.int64)])
            st, en = np.asarray(starts, dtype=np.int64)[rows], np.asarray(ends, dtype=np.int64)[rows]
            rows = rows[(cs[en - 1] - cs[st]) > 0]
        for r in rows:
            x = np.asarray(values[starts[r]:ends[r]])
            if np.isposinf(x[:-1]).any():
                candidates.append((int(r), min(j for j, _ in ar), MissingDataError("exog contains inf or nans")))
                break
    if candidates:
        candidates.sort(key=lambda c: (c[0], c[1]))
        raise candidates[0][2]
