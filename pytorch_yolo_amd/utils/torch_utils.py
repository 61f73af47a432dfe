"""Host-side weight preparation.

``fold_conv_bn`` / ``fuse_conv_and_bn`` restate the BN folding of the reference
(/root/reference/pytorch_yolo/utils/torch_utils.py:33-60):
    W' = diag(gamma / sqrt(eps + var)) W ,  b' = b + beta - gamma*mean/sqrt(var + eps)
It runs once, on the host, when a model is packed for the HIP path.
"""
from __future__ import annotations

import torch
from torch import nn


def fold_conv_bn(weight, conv_bias, gamma, beta, mean, var, eps):
    """Returns (folded weight, folded bias) as float32 CPU tensors."""
    weight = weight.detach().float().cpu()
    gamma, beta = gamma.detach().float().cpu(), beta.detach().float().cpu()
    mean, var = mean.detach().float().cpu(), var.detach().float().cpu()
    scale = gamma / torch.sqrt(eps + var)
    w = weight * scale.view(-1, 1, 1, 1)
    b = beta - gamma * mean / torch.sqrt(var + eps)
    if conv_bias is not None:
        b = conv_bias.detach().float().cpu() + b
    return w, b


def fuse_conv_and_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """Same contract as the reference helper: a biased Conv2d equivalent to bn(conv(x)) in eval mode."""
    w, b = fold_conv_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
    fused.train(conv.training)
    with torch.no_grad():
        fused.weight.copy_(w)
        fused.bias.copy_(b)
    return fused.to(conv.weight.device)
