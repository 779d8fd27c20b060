"""WAV file I/O for the entry scripts (the reference uses torchaudio.load / torchaudio.save,
infer.py:45,62,69; torchaudio is not a dependency here).  Conventions follow torchaudio:
`load` returns (float32 tensor [channels, frames] scaled to [-1, 1), sample_rate); `save` takes
[channels, frames] float32 and writes a 32-bit float WAV (what torchaudio.save does for a float32
tensor by default)."""
import numpy as np
import torch
from scipy.io import wavfile

SUPPORTED = ("wav",)


def load(path):
    if not str(path).lower().endswith(".wav"):
        raise ValueError(f"{path}: only WAV is readable without an audio codec library (ogg/mp3 need torchaudio/ffmpeg)")
    sr, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def info(path):
    """(frames, sample_rate, channels) from the RIFF header alone - what the entry scripts need to balance files over ranks before any
    rank reads audio data.  Chunks are walked until `data`; a streaming writer's 0 / 0xFFFFFFFF data size means "to the end of the file"."""
    import os
    import struct
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        channels = sr = align = None
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                raise ValueError(f"{path}: no data chunk")
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            if cid == b"fmt ":
                fmt = f.read(size + (size & 1))
                _tag, channels, sr, _rate, align, _bits = struct.unpack("<HHIIHH", fmt[:16])
            elif cid == b"data":
                if channels is None or not align:
                    raise ValueError(f"{path}: data chunk before fmt")
                if size in (0, 0xFFFFFFFF):
                    size = os.path.getsize(path) - f.tell()
                return size // align, int(sr), int(channels)
            else:
                f.seek(size + (size & 1), 1)


def save(path, src, sample_rate):
    x = src.detach().to("cpu", torch.float32)
    if x.dim() == 1:
        x = x[None]
    wavfile.write(path, int(sample_rate), np.ascontiguousarray(x.numpy().T))
