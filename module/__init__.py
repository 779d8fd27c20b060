"""Import shim: `from module.tinyvc import Encoder, Decoder`, `from module.infer import Generator,
StreamInfer`, `from module.utils import spectrogram` resolve to the MI355X implementation in
`tinyvc_amd.module`, so scripts written against the reference's package run unchanged."""
import sys

from tinyvc_amd.module import infer, tinyvc, utils  # noqa: F401

for _name, _mod in (("tinyvc", tinyvc), ("infer", infer), ("utils", utils)):
    sys.modules[f"{__name__}.{_name}"] = _mod
