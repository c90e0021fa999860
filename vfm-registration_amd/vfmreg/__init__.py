"""vfmreg -- MI355X-native correspondence-and-solve hot path of VFM-Registration.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic
of the path runs in hand-written HIP kernels behind the C ABI of ``include/vfmreg.h``.
"""
__version__ = "0.1.0"
