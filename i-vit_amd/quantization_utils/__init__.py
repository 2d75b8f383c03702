"""Same public names as /root/reference/models/quantization_utils/__init__.py:1-4 (the 'ivit' family)."""
from .quant_modules import (QuantLinear, QuantAct, QuantConv2d, QuantMatMul, attach_io_stat_hooks, save_io_stats_df, get_io_stats_df,
                            enable_io_stats, disable_io_stats, clear_io_stats, softmax)
from .ivit_modules import IVITIntGELU, IVITIntSoftmax, IVITIntLayerNorm
from .ibert_modules import IBERTIntGELU, IBERTIntSoftmax, IBERTIntLayerNorm
from .layer_selection import get_gelu, get_softmax, get_layernorm

# upstream I-ViT names, which the reference's swin_quant.py imports (swin_quant.py:11)
IntLayerNorm, IntSoftmax, IntGELU = IVITIntLayerNorm, IVITIntSoftmax, IVITIntGELU

__all__ = ["QuantLinear", "QuantAct", "QuantConv2d", "QuantMatMul", "IVITIntGELU", "IVITIntSoftmax",
           "IVITIntLayerNorm", "IBERTIntGELU", "IBERTIntSoftmax", "IBERTIntLayerNorm", "IntLayerNorm", "IntSoftmax", "IntGELU", "get_gelu", "get_softmax", "get_layernorm", "attach_io_stat_hooks", "save_io_stats_df",
           "get_io_stats_df", "enable_io_stats", "disable_io_stats", "clear_io_stats", "softmax"]
