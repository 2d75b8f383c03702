"""Which execution path a model mirror takes, and when its fused engine must be rebuilt.

The reference has one path (module by module, /root/reference/models/vit_quant.py:285-312).  The mirrors have two that
must agree bit for bit: the module path and the fused integer engine.  The engine is a snapshot of the model (integer
weights, requantisers, tables derived from the float parameters and the QuantAct ranges), so it is only taken

  * when it implements exactly what the module tree would compute -- `engine_unsupported_reason()` returns None:
    I-ViT operators, every QuantAct / Shiftmax / ShiftGELU width that the kernels hard-wire (the eight width knobs of
    vit_quant.py:180-187 / quant_train.py:295-306 change some of them), supported geometry; anything else runs module by
    module, which honours every width it can represent and raises for those it cannot;
  * while the snapshot is current -- the cache is keyed on the device and on a fingerprint of every parameter's and
    buffer's autograd version counter, and dropped by `load_state_dict`, `.to()/.cuda()` (`_apply`), `train()`,
    `fix()/unfix()` (freeze_model / unfreeze_model).  Tensors edited through `.data` bypass version counters: call
    `invalidate_engine()` after such surgery.
"""
from __future__ import annotations

import torch

from .quantization_utils import QuantAct
from .quantization_utils import quant_modules as _qm


class EngineDispatch:
    """Mixin for VisionTransformer / SwinTransformer (nn.Module subclasses)."""

    # QuantActs that never gate the engine (Swin's unused act_out, swin_quant.py:518)
    _frozen_exempt = ()

    def _init_dispatch(self):
        self._engine = None          # (device, fingerprint, engine)
        self._tracked = None         # flat tensor list behind the fingerprint
        self._qacts = None
        self.use_engine = True       # frozen models take the fused engine when it is exact for them
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_engine())

    # -- invalidation ---------------------------------------------------------------------------------------------
    def invalidate_engine(self):
        self._engine = None
        self._tracked = None

    def _apply(self, fn, *a, **k):          # .to() / .cuda() / .float(): parameters move or change
        self.invalidate_engine()
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        self.invalidate_engine()
        return super().train(mode)

    def fix(self):
        self.invalidate_engine()

    def unfix(self):
        self.invalidate_engine()

    def _fingerprint(self):
        if self._tracked is None:
            self._tracked = list(self.parameters()) + list(self.buffers())
        # version counters only grow, so their sum over a fixed tensor list changes whenever any tensor is written in place
        return sum(t._version for t in self._tracked)

    # -- state ----------------------------------------------------------------------------------------------------
    def _quant_acts(self):
        if self._qacts is None:
            self._qacts = [(n, m) for n, m in self.named_modules() if isinstance(m, QuantAct) and n not in self._frozen_exempt]
        return self._qacts

    def is_frozen(self):
        return all(not m.running_stat for _, m in self._quant_acts())

    def engine_unsupported_reason(self):
        raise NotImplementedError

    def _build_engine(self, device, max_batch):
        raise NotImplementedError

    def _width_mismatch(self, expected: dict):
        """first QuantAct whose activation_bit differs from what the fused kernels hard-wire (`expected`: full module name ->
        width; every QuantAct not listed is 8 bit)"""
        for n, m in self._quant_acts():
            want = expected.get(n, 8)
            if int(m.activation_bit) != want:
                return f"QuantAct {n} is {int(m.activation_bit)}-bit (fused engine: {want})"
        return None

    def engine(self, max_batch):
        """The fused engine for the model's CURRENT parameters and ranges (rebuilt when either changed)."""
        device = next(self.parameters()).device
        fp = self._fingerprint()
        cur = self._engine
        if cur is not None and cur[0] == device and cur[1] == fp and cur[2].max_batch >= max_batch:
            return cur[2]
        if cur is not None and cur[0] == device and cur[1] == fp:
            max_batch = max(max_batch, cur[2].max_batch)     # grow, never thrash between batch sizes
        eng = self._build_engine(device, max_batch)
        self._engine = (device, fp, eng)
        return eng

    def _reason_cached(self):
        """engine_unsupported_reason(), re-evaluated only when a QuantAct's width changed (it builds 4 * depth dictionaries; this
        predicate runs on every forward, and a DeiT-T batch-1 forward is 0.7 ms)"""
        key = tuple(int(m.activation_bit) for _, m in self._quant_acts())
        c = self.__dict__.get("_reason_cache")
        if c is None or c[0] != key:
            c = self.__dict__["_reason_cache"] = (key, self.engine_unsupported_reason())
        return c[1]

    def takes_engine(self, x: torch.Tensor) -> bool:
        if getattr(self, "_io_stat_hooks", False) and _qm.io_stats_enabled():
            return False      # attach_io_stat_hooks: the collector's hooks sit on the sub-modules, which the fused engine never calls
        return (self.use_engine and not self.training and x.is_cuda and self.is_frozen()
                and self._reason_cached() is None)

    def ranges(self):
        return {n: (float(m.x_min.reshape(-1)[0]), float(m.x_max.reshape(-1)[0]))
                for n, m in self.named_modules() if isinstance(m, QuantAct)}
