"""MI355X-native hot path of ylsung/vl-merging (directory `vl-merging_amd/`, import name `vl_merging_amd`).

Host side: Python mirroring the reference's module / merge API.  Device side: hand-written HIP kernels for
gfx950 behind the C ABI declared in include/vlm_hip.h (libvlm_hip.so, built by build_ext.py).
There is NO CPU fallback: every op raises if the HIP library is missing or a tensor is not on the GPU.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
