"""`token_compression.monkeypatch` (reference: qwen-evaluation/token_compression/monkeypatch.py)."""
from visionselector_amd.monkeypatch import replace_qwen25vl  # noqa: F401
