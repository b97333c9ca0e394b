"""mitsuba-renderer_amd: MI355X-native path-tracing hot path behind Mitsuba 0.2.1's
integrator interface.  Host-side mirror (ctypes over the C ABI in include/mtsgpu.h)."""
from . import abi, scenes  # noqa: F401
