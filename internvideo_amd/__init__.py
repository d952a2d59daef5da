"""internvideo_amd: MI355X (gfx950) native InternVideo2 masked video-ViT training path.

Host code is PyTorch-ROCm (memory, streams, torch.distributed/RCCL); all hot-path arithmetic is hand-written HIP
in csrc/ behind the C ABI of include/internvideo_hip.h.  There is no CPU or PyTorch-op fallback."""
from .lib import InternVideoHipError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
