"""String -> operator class registry (/root/reference/models/quantization_utils/layer_selection.py:116-236).
The integer families 'ivit' and 'ibert' are implemented on MI355X; the other names of the reference ('ppoly',
'float') are outside this build's scope and raise."""
from .ibert_modules import IBERTIntGELU, IBERTIntLayerNorm, IBERTIntSoftmax
from .ivit_modules import IVITIntGELU, IVITIntLayerNorm, IVITIntSoftmax

GELU_REGISTRY = {"ivit": IVITIntGELU, "ibert": IBERTIntGELU}
SOFTMAX_REGISTRY = {"ivit": IVITIntSoftmax, "ibert": IBERTIntSoftmax}
LN_REGISTRY = {"ivit": IVITIntLayerNorm, "ibert": IBERTIntLayerNorm}


def _parse_layer_name(name: str):
    """'base_arg_value_arg_value...' -> (base, {arg: value}); hyphens in arg names become underscores."""
    parts = name.lower().split("_")
    if len(parts) < 3:
        return name.lower(), {}
    params = {}
    for arg, val in zip(parts[1::2], parts[2::2]):
        if val in ("true", "false"):
            v = val == "true"
        elif val.isdigit():
            v = int(val)
        else:
            try:
                v = float(val)
            except ValueError:
                v = val
        params[arg.replace("-", "_")] = v
    return parts[0], params


def _get(registry, name, kind):
    base, params = _parse_layer_name(name)
    if base not in registry:
        raise KeyError(f"{kind} type {name!r}: only {sorted(registry)} are implemented by the MI355X integer path")
    cls = registry[base]
    if not params:
        return cls

    class Parameterized(cls):  # same mechanism as the reference: defaults injected through kwargs
        def __init__(self, *args, **kwargs):
            for k, v in params.items():
                kwargs.setdefault(k, v)
            super().__init__(*args, **kwargs)

    Parameterized.__name__ = f"Parameterized{cls.__name__}"
    return Parameterized


def get_gelu(name: str):
    return _get(GELU_REGISTRY, name, "gelu")


def get_softmax(name: str):
    return _get(SOFTMAX_REGISTRY, name, "softmax")


def get_layernorm(name: str):
    return _get(LN_REGISTRY, name, "layernorm")
