"""`compression_method.monkeypatch` (reference: llava-ov-15/compression_method/monkeypatch.py)."""
from visionselector_amd.monkeypatch import replace_llavaov15  # noqa: F401
