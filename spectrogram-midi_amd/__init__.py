"""MI355X-native analyze hot path of Aegis Engine (audio -> mel/dB/rake, pYIN, RMS -> note
events), behind the reference's own `AegisEngine` surface (/root/reference/aegis_engine.py).
The arithmetic runs in hand-written HIP kernels (csrc/) reached through the C ABI declared in
include/aegis_hip.h; there is no CPU fallback -- a missing extension or GPU is an error."""
__version__ = "0.1.0"
