from .generator import Generator
from .stream import StreamInfer, BatchedStreamInfer

__all__ = ["Generator", "StreamInfer", "BatchedStreamInfer"]
