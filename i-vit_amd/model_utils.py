"""freeze_model / unfreeze_model (/root/reference/models/model_utils.py:5-35)."""
import torch.nn as nn


def freeze_model(model: nn.Module):
    """eval() and fix() every sub-module that has it -- call after calibration / checkpoint load."""
    model.eval()
    for m in model.modules():
        fix = getattr(m, "fix", None)
        if callable(fix):
            fix()


def unfreeze_model(model: nn.Module):
    for m in model.modules():
        unfix = getattr(m, "unfix", None)
        if callable(unfix):
            unfix()
