"""Import-path shim for `token_compression.selector_scorer` (reference: qwen-evaluation/token_compression/selector_scorer.py)."""
from visionselector_amd.selector import TransformerScorer  # noqa: F401
