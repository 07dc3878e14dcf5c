"""ks265codec_amd — MI355X-native HEVC encode pixel-kernel path (hand-written HIP behind a C ABI).

The product is `libks265hip.so` (csrc/*.hip, include/ks265_hip.h).  This package is the thin host-side
mirror used by tests and bench: ctypes bindings + torch for device memory/streams.  There is NO CPU
fallback: importing `ks265codec_amd.lib` without the built library, or creating a context without a
gfx950 device, raises.
"""
from .lib import Ks265Error, KsContext, load_library  # noqa: F401

__all__ = ["Ks265Error", "KsContext", "load_library"]
