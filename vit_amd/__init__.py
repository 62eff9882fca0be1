"""vit_amd: MI355X (gfx950) implementation of the video-ViT encoder hot path of HananKibria/VIT.

Python host code (PyTorch-ROCm for memory, streams and torch.distributed) over hand-written HIP
kernels in libvit_amd.so (C ABI: include/vit_amd.h).  There is no CPU or eager fallback.
"""
__version__ = "0.1.0"
