"""visionselector_amd -- MI355X (gfx950) implementation of the VisionSelector hot path.

The arithmetic lives in libvsel.so (hand-written HIP, C-ABI declared in include/vsel.h); this package
is the PyTorch-ROCm host side that mirrors the reference's ``compression_method`` / ``token_compression``
module API.  There is no CPU fallback: every op raises if libvsel.so is missing or the tensors are
not on a ROCm device.
"""
__version__ = "0.1.0"
