"""Feature selection on the extracted matrix.

Mirrors tsfresh/feature_selection/__init__.py: `select_features`, `calculate_relevance_table` and helpers.  The
per-feature work of the reference (a scipy call per column: sort, rank, count) runs as ONE batched sweep of HIP
kernels behind `tsfa_relevance_classes` (include/tsfresh_amd.h); what is left on the host is the O(1) p-value tail of
each test and the Benjamini-Hochberg / -Yekutieli procedure.  Classification targets: Mann-Whitney U / Fisher
(`tsfa_relevance_classes`; `'smir'`: Kolmogorov-Smirnov, `tsfa_relevance_classes_ks`); regression targets: Kendall's tau /
Kolmogorov-Smirnov (`tsfa_relevance_real`).
"""
from tsfresh_amd.feature_selection.relevance import (calculate_relevance_table, combine_relevance_tables,  # noqa: F401
                                                     get_feature_type, infer_ml_task)
from tsfresh_amd.feature_selection.selection import select_features  # noqa: F401
