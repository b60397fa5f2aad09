"""cerebro_amd -- MI355X-native loop-detection core of mpkuse/cerebro (descriptor dot-product scan +
DLS-PnP-in-RANSAC) behind the C-ABI of include/cerebro_hip.h.  Python here is plumbing (ctypes,
torch.distributed); the product is cerebro_amd/lib/libcerebro_hip.so (hand-written gfx950 HIP)."""
from .capi import Chip, ChipError, load_library, default_dot_params, default_ransac_params  # noqa: F401

__all__ = ["Chip", "ChipError", "load_library", "default_dot_params", "default_ransac_params"]
