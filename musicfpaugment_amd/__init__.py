"""MI355X-native implementation of the musicFPaugment hot path (STFT -> UNet denoiser -> peak picking ->
peak-mask metrics).  Host side: Python on PyTorch-ROCm mirroring the reference's call surface; device
side: hand-written gfx950 HIP kernels in libmfpa.so behind the C ABI of include/mfpa.h."""
__version__ = "0.1.0"
