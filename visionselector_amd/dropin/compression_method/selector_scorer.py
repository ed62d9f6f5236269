"""Import-path shim for `compression_method.selector_scorer` (reference: */compression_method/selector_scorer.py)."""
from visionselector_amd.selector import TransformerScorer  # noqa: F401
