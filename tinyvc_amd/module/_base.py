"""Shared machinery of the host-side modules: every `Encoder` / `Decoder` / `Generator` lazily owns
an `Engine` (a tvc_ctx with its packed weights) per device and re-uploads only when a parameter
changed (load_state_dict, .to(), in-place edits)."""
import operator

import torch
import torch.nn as nn

from .. import _lib
from ..engine import Engine


_VERSION = operator.attrgetter("_version")


_EPOCH = [0]        # bumped whenever ANY module's structure may have changed: every cached parameter list (a child's or its owner's) is stale then


def _bump(*_args):
    _EPOCH[0] += 1


# torch calls these for every Parameter / sub-module registration of every nn.Module in the process - also the ones behind
# `module.weight = nn.Parameter(...)`, `gen.decoder = other` and load_state_dict(assign=True) on any descendant, stock nn.Conv1d
# leaves included.  A registration anywhere only costs the next call one walk of its own parameter tree.  The hooks are process-global
# (torch has no per-module form); `_HOOK_HANDLES` keeps their handles so an embedding application can remove them
# (`for h in _HOOK_HANDLES: h.remove()` - the periodic re-walk in `_param_list` then remains as the only invalidation).
_HOOK_HANDLES = (torch.nn.modules.module.register_module_parameter_registration_hook(_bump),
                 torch.nn.modules.module.register_module_module_registration_hook(_bump))


class HipModule(nn.Module):
    """nn.Module whose parameters are a checkpoint container; compute happens in the HIP library."""

    def _weight_tensors(self):
        return self.state_dict()

    def _param_list(self):
        """The parameter objects, cached: walking the module tree costs ~240 us for the 282 tensors of a Generator, which
        would be paid on every convert call and every streaming block.  The cache is keyed on a process-wide epoch that every
        structural change bumps - torch's global registration hooks fire for a Parameter or sub-module assigned anywhere in the tree
        (`gen.decoder = other`, `conv.weight = nn.Parameter(...)`, load_state_dict(assign=True) on a child), `_apply` (.to(), .float())
        and `load_state_dict` bump it themselves - so an owner never keeps
        running (or keeps replaying a captured stream graph) on a child's replaced weights."""
        d = self.__dict__
        c = d.get("_plist")
        n = d["_plist_uses"] = d.get("_plist_uses", 0) + 1
        # Mutations that bypass the registration hooks (`module._parameters[k] = ...`, `del module.weight`) are caught by a re-walk every
        # 256th use: at most 256 calls run on the replaced weights, and the walk costs < 1 us per call amortised.
        if c is None or c[0] != _EPOCH[0] or (n & 255) == 0:
            c = d["_plist"] = (_EPOCH[0], list(self.parameters()))
        return c[1]

    def _weights_key(self):
        # in-place edits bump `_version`; .to() / .float() swap the storage (data_ptr)
        ps = self._param_list()
        return tuple(map(_VERSION, ps)), tuple(p.data_ptr() for p in ps)

    def _apply(self, fn, *args, **kwargs):
        _bump()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        _bump()
        r = super().load_state_dict(*args, **kwargs)
        _bump()                      # (assign=True swaps the Parameter objects during the call)
        return r

    def _module_device(self):
        p = next(self.parameters(), None)
        return p.device if p is not None else torch.device("cpu")

    def engine(self, device=None):
        dev = torch.device(device) if device is not None else self._module_device()
        if dev.type != "cuda":
            raise _lib.TinyVCError(
                f"{type(self).__name__} is on {dev}: tinyvc_amd executes on an AMD GPU only and has "
                "no CPU fallback; call .to('cuda') (the CPU reference lives in oracle/ for tests).")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        engines = self.__dict__.setdefault("_engines", {})
        eng = engines.get(dev.index)
        if eng is None:
            eng = engines[dev.index] = Engine(dev)
        key = self._weights_key()
        if eng.weights_key != key:
            eng.load_weights(self._weight_tensors())
            eng.weights_key = key
        return eng

    def _input_device(self, t):
        """Tensors handed in on the CPU are moved to the module's device (the reference's infer.py
        leaves the waveform on the CPU, infer.py:62-66)."""
        dev = self._module_device()
        return t.to(dev) if t.device != dev else t
