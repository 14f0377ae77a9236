"""tsfresh_amd: the feature-extraction hot path of tsfresh, MI355X-native.

`extract_features` keeps the reference's DataFrame-in / DataFrame-out contract and FCParameters dictionaries
(tsfresh/feature_extraction/extraction.py:30); the arithmetic runs in hand-written HIP kernels for gfx950 behind the
C-ABI declared in include/tsfresh_amd.h.  There is no CPU compute path.
"""
from tsfresh_amd.feature_extraction.extraction import extract_features, extract_rolled_features  # noqa: F401
from tsfresh_amd.feature_extraction.settings import (  # noqa: F401
    ComprehensiveFCParameters,
    EfficientFCParameters,
    MinimalFCParameters,
)

from tsfresh_amd.feature_selection import calculate_relevance_table, select_features  # noqa: F401,E402
from tsfresh_amd.convenience import extract_relevant_features  # noqa: F401,E402

__version__ = "0.1.0"
