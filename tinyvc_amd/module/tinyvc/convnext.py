"""Parameter containers of the ConvNeXt-v2 1-D layer (reference module/tinyvc/convnext.py:7-58).
State-dict keys and shapes match the reference; the arithmetic
(dw conv k7 -> LayerNorm -> 1x1 -> GELU -> GRN -> 1x1 -> +res) runs in csrc/encoder.hip."""
import torch
import torch.nn as nn


class LayerNorm(nn.Module):
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))


class GRN(nn.Module):
    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.beta = nn.Parameter(torch.zeros(1, channels, 1))
        self.gamma = nn.Parameter(torch.zeros(1, channels, 1))


class ConvNeXtLayer(nn.Module):
    def __init__(self, channels=512, kernel_size=7, mlp_mul=2, dilation=1):
        super().__init__()
        if kernel_size != 7 or mlp_mul != 2:
            raise NotImplementedError("the HIP path implements kernel_size=7, mlp_mul=2 (the reference's only use)")
        self.dilation = dilation
        self.c1 = nn.Conv1d(channels, channels, kernel_size, groups=channels, dilation=dilation,
                            padding=(kernel_size - 1) * dilation // 2, padding_mode="replicate")
        self.norm = LayerNorm(channels)
        self.c2 = nn.Conv1d(channels, channels * mlp_mul, 1)
        self.grn = GRN(channels * mlp_mul)
        self.c3 = nn.Conv1d(channels * mlp_mul, channels, 1)
