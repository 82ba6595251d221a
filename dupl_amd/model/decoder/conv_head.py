"""LargeFOV segmentation head (reference: model/decoder/conv_head.py:11-41): parameter container only --
its forward/backward are scheduled by dupl_amd.engine (im2col + MFMA GEMM with fused ReLU)."""
import torch.nn as nn


class _Weight(nn.Module):
    def __init__(self, weight):
        super().__init__()
        self.weight = weight


class LargeFOV(nn.Module):
    def __init__(self, conv6, conv7, conv8, dilation=5):
        super().__init__()
        self.embed_dim = 512
        self.dilation = dilation
        self.conv6, self.conv7, self.conv8 = _Weight(conv6), _Weight(conv7), _Weight(conv8)

    def forward(self, x):
        raise RuntimeError("LargeFOV runs inside network.forward (dupl_amd.engine.network_forward)")
