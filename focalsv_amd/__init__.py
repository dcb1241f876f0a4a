"""focalsv_amd -- MI355X-native replacement for FocalSV's per-region local assembly +
contig-based DEL/INS calling hot path (focalsv/3_assembly*, focalsv/4_sv_calling*).

All compute goes through libfocalsv_hip.so (hand-written HIP for gfx950) via the
C ABI in include/focalsv_hip.h.  There is no CPU fallback: importing the kernels
without the built library, or running them without an MI355X, raises.
"""
__version__ = "0.1.0"
