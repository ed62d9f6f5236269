"""`token_compression.monkeypatch` (reference: qwen-evaluation/token_compression/monkeypatch.py:50-107).
`selector` needs no patching (the reference has no branch for it either); a baseline method is forwarded to the
reference's own module when its package root sits behind dropin/ on sys.path."""
from visionselector_amd import monkeypatch as _mp


def replace_qwen25vl(args, model, method):
    ref = None if method in _mp._PASSTHROUGH else _mp.forward_to_reference(__package__, __file__, "replace_qwen25vl")
    return _mp.replace_qwen25vl(args, model, method, ref)
