"""music2dance_amd — MI355X-native WGAN-GP training engine for audio-conditioned
dance-motion generation (drop-in for the phase1/2/3 hot path of clementabary/music2dance).

Host code is Python on PyTorch-ROCm; all hot ops are hand-written gfx950 HIP kernels in
libm2d_hip.so (include/m2d.h), loaded with ctypes. There is no CPU fallback.
"""
__version__ = "0.1.0"
