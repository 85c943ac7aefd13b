"""drn-wsod-pytorch_amd — MI355X-native (gfx950) DRN-WSOD / OICR hot path.

The directory name carries a hyphen (it is the name the build contract asks for), so the package is
imported through `__graft_entry__.load_package()` under the module name `drn_wsod_pytorch_amd`.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/drn_wsod.h), lib/ (built .so),
_cabi.py (ctypes binding, fails loudly), ops.py (tensor-level wrappers), and the host-side mirror of
the reference's operator surface (layers, structures, config, modeling/*, engine).
"""
from . import _cabi  # noqa: F401

_PRECISION = "bf16"


def set_precision(p):
    """'bf16' (fast mode: bf16 operands, fp32 accumulate) or 'fp32' (parity mode: exact fp32 MFMA)."""
    global _PRECISION
    assert p in ("bf16", "fp32")
    _PRECISION = p


def get_precision():
    return _PRECISION


def compute_dtype():
    import torch

    return torch.bfloat16 if _PRECISION == "bf16" else torch.float32
