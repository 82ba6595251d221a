"""LargeFOV segmentation head (reference: model/decoder/conv_head.py:11-41).  Inside network.forward its forward / backward are
scheduled by dupl_amd.engine (im2col + split GEMM with fused ReLU, engine.network_forward / network_backward); called on its own
(`model.decoder(x4)`, as the reference's tools may) it is an autograd node of its own on the same kernels (engine.LargeFOVFn)."""
import torch
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
        """conv_head.py:32-41 on a feature map x (B, C, h, w) -> (B, classes, h, w): conv3x3 d5 + ReLU, conv3x3 d5 + ReLU, conv1x1.
        Differentiable on its own (engine.LargeFOVFn: x and the three conv weights receive gradients, like the reference's module);
        the training path back-propagates the head inside network.forward and never comes through here."""
        from ... import engine
        if not x.is_cuda:
            raise RuntimeError("dupl_amd runs on an MI355X only (no CPU path)")
        w6, w7, w8 = self.conv6.weight, self.conv7.weight, self.conv8.weight
        if torch.is_grad_enabled() and (x.requires_grad or w6.requires_grad or w7.requires_grad or w8.requires_grad):
            return engine.LargeFOVFn.apply(x, w6, w7, w8, self.dilation)
        with torch.no_grad():
            return engine.large_fov_forward(x.contiguous().float(), w6, w7, w8, self.dilation)
