"""i-vit_amd: MI355X-native integer-only ViT inference path.

Public surface mirrors the reference's `models` package for the I-ViT path
(/root/reference/models/__init__.py): the quantised operator modules, the DeiT/ViT and Swin factories and
freeze_model/unfreeze_model.  Everything computes through hand-written HIP kernels behind the C ABI of
include/ivit_hip.h; there is no PyTorch or CPU fallback."""
from .quantization_utils import *  # noqa: F401,F403
from .vit_quant import *  # noqa: F401,F403
from .swin_quant import *  # noqa: F401,F403
from .model_utils import freeze_model, unfreeze_model  # noqa: F401
