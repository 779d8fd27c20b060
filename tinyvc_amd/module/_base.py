"""Shared machinery of the host-side modules: every `Encoder` / `Decoder` / `Generator` lazily owns
an `Engine` (a tvc_ctx with its packed weights) per device and re-uploads only when a parameter
changed (load_state_dict, .to(), in-place edits)."""
import operator

import torch
import torch.nn as nn

from .. import _lib
from ..engine import Engine


_VERSION = operator.attrgetter("_version")


class HipModule(nn.Module):
    """nn.Module whose parameters are a checkpoint container; compute happens in the HIP library."""

    def _weight_tensors(self):
        return self.state_dict()

    def _param_list(self):
        """The parameter objects, cached: walking the module tree costs ~240 us for the 282 tensors of a Generator, which
        would be paid on every convert call and every streaming block.  Dropped whenever the module is moved / cast
        (`_apply`) or a checkpoint is loaded, and refreshed every 256 uses in case a Parameter object was re-assigned."""
        d = self.__dict__
        age = d.get("_plist_age", 0) + 1
        if d.get("_plist") is None or age > 256:
            d["_plist"] = list(self.parameters())
            age = 0
        d["_plist_age"] = age
        return d["_plist"]

    def _weights_key(self):
        # in-place edits bump `_version`; .to() / .float() swap the storage (data_ptr)
        ps = self._param_list()
        return tuple(map(_VERSION, ps)), tuple(p.data_ptr() for p in ps)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_plist", None)
        return super().load_state_dict(*args, **kwargs)

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
