from tsfresh_amd.feature_extraction.extraction import extract_features  # noqa: F401
from tsfresh_amd.feature_extraction.settings import (  # noqa: F401
    ComprehensiveFCParameters,
    EfficientFCParameters,
    IndexBasedFCParameters,
    MinimalFCParameters,
    TimeBasedFCParameters,
    from_columns,
)
