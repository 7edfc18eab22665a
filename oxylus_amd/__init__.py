"""oxylus_amd -- MI355X-native meshlet visibility pipeline behind Oxylus' cull entry points.

Package contents are only what the hot path needs: csrc/ (HIP kernels + C ABI), host/ (C++
drop-in shim), lib.py (ctypes binding), renderer.py (Python twin of the shim), synth.py
(synthetic scenes in the reference layouts).
"""
from . import lib  # noqa: F401
