"""Darknet53 feature extractor, same interface as the reference's backbones.py.

CSPDarknet53 of the reference (backbones.py:62-264) cannot be constructed there
(it passes three positionals to ResidialBlock, layers.py:93) and is not provided.
"""
import torch
from torch import nn

from .layers import YOLOConv, ResidialBlock


def flat_convs(stack):
    """The YOLOConv layers of a stack in execution order."""
    out = []
    for layer in stack:
        if isinstance(layer, ResidialBlock):
            out.extend(layer.layer_stack)
        else:
            out.append(layer)
    return out


class Darknet53(nn.Module):
    """52-conv Darknet-53 (backbones.py:6-60).

    ``Darknet53(**kwargs)(x)`` with x [B,S,S,3] returns (scale_1 [B,S/8,S/8,256],
    scale_2 [B,S/16,S/16,512], scale_3 [B,S/32,S/32,1024]).  Layers use the
    YOLOConv defaults: bias + Mish, no batch norm (backbones.py:10-42 override
    nothing).  kwargs: name, dtype (compute type; default follows the input, with
    an fp32 image accepted by a bf16 network's first layer), seed (parameter init).
    """

    def __init__(self, **kwargs):
        super().__init__()
        self.name = kwargs.pop("name", None)
        dtype = kwargs.pop("dtype", None)
        seed = kwargs.pop("seed", None)
        if kwargs:
            raise TypeError("Darknet53: unexpected keyword arguments %s" % sorted(kwargs))
        d = dict(dtype=dtype)

        s1 = [YOLOConv((3, 3, 32), **d), YOLOConv((3, 3, 64), downsample=True, **d)]
        for _ in range(1):
            s1.append(ResidialBlock(32, 64, **d))
        s1.append(YOLOConv((3, 3, 128), downsample=True, **d))
        for _ in range(2):
            s1.append(ResidialBlock(64, 128, **d))
        s1.append(YOLOConv((3, 3, 256), downsample=True, **d))
        for _ in range(8):
            s1.append(ResidialBlock(128, 256, **d))
        s2 = [YOLOConv((3, 3, 512), downsample=True, **d)]
        for _ in range(8):
            s2.append(ResidialBlock(256, 512, **d))
        s3 = [YOLOConv((3, 3, 1024), downsample=True, **d)]
        for _ in range(4):
            s3.append(ResidialBlock(512, 1024, **d))
        self.layer_stack_1 = nn.ModuleList(s1)
        self.layer_stack_2 = nn.ModuleList(s2)
        self.layer_stack_3 = nn.ModuleList(s3)

        # the channel flow is static, so build every kernel now (Keras would on first call)
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        cin = 3
        for conv in self.conv_layers():
            conv.build(cin, generator=gen)
            cin = conv.filters

    def conv_layers(self):
        """All 52 YOLOConv layers in execution order."""
        return flat_convs(self.layer_stack_1) + flat_convs(self.layer_stack_2) + flat_convs(self.layer_stack_3)

    def load_reference_params(self, params):
        convs = self.conv_layers()
        if len(params) != len(convs):
            raise ValueError("expected %d parameter dicts, got %d" % (len(convs), len(params)))
        for conv, p in zip(convs, params):
            conv.load_reference_params(p)

    def forward(self, x):
        for layer in self.layer_stack_1:
            x = layer(x)
        scale_1 = x
        for layer in self.layer_stack_2:
            x = layer(x)
        scale_2 = x
        for layer in self.layer_stack_3:
            x = layer(x)
        scale_3 = x
        return scale_1, scale_2, scale_3
