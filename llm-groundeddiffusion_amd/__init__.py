"""lgd_amd — MI355X-native stage-2 denoising hot path of LLM-grounded Diffusion (LMD / LMD+).

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed only).
Device side: liblgd_hip.so, hand-written gfx950 kernels behind the C ABI of include/lgd_hip.h.
"""
__version__ = "0.1.0"
