"""svim_asm_amd — MI355X-native hot path of SVIM-asm (SV-signature extraction + diploid pairing).

Host side mirrors the reference's Python seams (SVIM_intra / SVIM_inter / SVIM_COLLECT /
SVIM_COMBINE / SVCandidate); the arithmetic runs in hand-written HIP kernels (libsvx.so,
C-ABI in include/svx.h) bound through ctypes in `_lib`.  No CPU fallback.
"""
__version__ = "1.0.3+svx0.1"
