"""Import-path shim for the `flash_attn` package (third party, no ROCm build in this image): the functions the reference's
vendored modeling files import, backed by libvsel.  See visionselector_amd/flash_attn_compat.py."""
from visionselector_amd.flash_attn_compat import flash_attn_func, flash_attn_varlen_func  # noqa: F401

__version__ = "2.7.4.post1+vsel"
