"""`compression_method.monkeypatch` (reference: llava-ov-15/compression_method/monkeypatch.py:34-70).
`selector` needs no patching (the reference has no branch for it either); a baseline method is forwarded to the
reference's own module when its package root sits behind dropin/ on sys.path."""
from visionselector_amd import monkeypatch as _mp


def replace_llavaov15(args, model, method):
    ref = None if method in _mp._PASSTHROUGH else _mp.forward_to_reference(__package__, __file__, "replace_llavaov15")
    return _mp.replace_llavaov15(args, model, method, ref)
