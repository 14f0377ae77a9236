"""High-level compositions of the reference (tsfresh/convenience/)."""
from tsfresh_amd.convenience.relevant_extraction import extract_relevant_features  # noqa: F401
