"""Darknet53 and CSPDarknet53 feature extractors, same interface as the reference's
backbones.py.

CSPDarknet53 of the reference (backbones.py:62-264) cannot be constructed there: it passes
three positionals to ResidialBlock(filter_num1, filter_num2, activation) (layers.py:93) and
gives its second conv 54 filters where every consumer expects 64 (backbones.py:68).  The
class here is the repaired definition (ResidialBlock(in, f1, f2) -> ResidialBlock(f1, f2),
54 -> 64), with the same attribute names and data flow.
"""
import torch
from torch import nn

from . import training as _training
from .layers import YOLOConv, ResidialBlock, spp_concat


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


class CSPDarknet53(nn.Module):
    """CSP backbone with the SPP block (backbones.py:62-264, repaired: see the module text).

    ``CSPDarknet53()(x)`` with x [B,S,S,3] returns (route_1 [B,S/8,S/8,256], route_2
    [B,S/16,S/16,512], x [B,S/32,S/32,512]).  Every tf.concat of two branches
    (backbones.py:187, 200, 213, 231, 249) is written in place: the last convolution of each
    branch stores straight into its channel slice of the concatenated tensor.  The three
    max-pools and their concat (backbones.py:254-259) are one kernel.  kwargs: name, dtype,
    seed, as Darknet53.
    """

    # (attribute, input channels, items); an item is (k, filters, downsample) or ("res", f1, f2)
    _STACKS = (
        ("layer_stack_1", 3, ((3, 32, False), (3, 64, True))),
        ("layer_stack_2", 64, ((1, 64, False),)),
        ("layer_stack_2_skip", 64, ((1, 64, False), ("res", 32, 64), (1, 64, False))),
        ("layer_stack_3", 128, ((1, 64, False), (3, 128, True))),
        ("layer_stack_4", 128, ((1, 64, False),) + (("res", 64, 64),) * 2 + ((1, 64, False),)),
        ("layer_stack_4_skip", 128, ((1, 64, False),)),
        ("layer_stack_5", 128, ((1, 128, False), (3, 256, True))),
        ("layer_stack_6", 256, ((1, 128, False),) + (("res", 128, 128),) * 8 + ((1, 128, False),)),
        ("layer_stack_6_skip", 256, ((1, 128, False),)),
        ("layer_stack_7", 256, ((1, 256, False),)),
        ("layer_stack_8", 256, ((3, 512, True),)),
        ("layer_stack_9", 512, ((1, 256, False),) + (("res", 256, 256),) * 8 + ((1, 256, False),)),
        ("layer_stack_9_skip", 512, ((1, 256, False),)),
        ("layer_stack_10", 512, ((1, 512, False),)),
        ("layer_stack_11", 512, ((3, 1024, True),)),
        ("layer_stack_12", 1024, ((1, 512, False),) + (("res", 512, 512),) * 4 + ((1, 512, False),)),
        ("layer_stack_12_skip", 1024, ((1, 512, False),)),
        ("layer_stack_13", 1024, ((1, 1024, False), (1, 512, False), (3, 1024, False), (1, 512, False))),
        ("layer_stack_14", 2048, ((1, 512, False), (3, 1024, False), (1, 512, False))),
    )

    def __init__(self, **kwargs):
        super().__init__()
        self.name = kwargs.pop("name", None)
        dtype = kwargs.pop("dtype", None)
        seed = kwargs.pop("seed", None)
        if kwargs:
            raise TypeError("CSPDarknet53: unexpected keyword arguments %s" % sorted(kwargs))
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for attr, cin, items in self._STACKS:
            layers = []
            for it in items:
                if it[0] == "res":
                    layers.append(ResidialBlock(it[1], it[2], activation="mish", dtype=dtype))
                else:
                    layers.append(YOLOConv((it[0], it[0], it[1]), downsample=it[2], activation="mish", dtype=dtype))
            stack = nn.ModuleList(layers)
            setattr(self, attr, stack)
            for conv in flat_convs(stack):
                conv.build(cin, generator=gen)
                cin = conv.filters

    def conv_layers(self):
        """All 78 YOLOConv layers in attribute order (residual blocks expanded)."""
        out = []
        for attr, _cin, _items in self._STACKS:
            out.extend(flat_convs(getattr(self, attr)))
        return out

    def load_reference_params(self, params):
        convs = self.conv_layers()
        if len(params) != len(convs):
            raise ValueError("expected %d parameter dicts, got %d" % (len(convs), len(params)))
        for conv, p in zip(convs, params):
            conv.load_reference_params(p)

    @staticmethod
    def _run(stack, x, out=None):
        last = len(stack) - 1
        for i, layer in enumerate(stack):
            x = layer(x, out=out) if (i == last and out is not None) else layer(x)
        return x

    def _csp(self, main, skip, x_main, x_skip):
        """concat([main(x_main), skip(x_skip)], -1) with both branches' last convolutions
        writing into their slices of one buffer."""
        c1 = flat_convs(main)[-1].filters
        c2 = flat_convs(skip)[-1].filters
        probe = flat_convs(main)[-1]
        dt = probe.compute_dtype or x_main.dtype
        b, h, w = x_main.shape[0], x_main.shape[1], x_main.shape[2]
        if _training.enabled():
            # recorded for back-propagation: the branches' gradients are the two channel slices of the concat's
            return torch.cat([self._run(main, x_main), self._run(skip, x_skip)], dim=-1)
        cat = torch.empty((b, h, w, c1 + c2), device=x_main.device, dtype=dt)
        self._run(main, x_main, out=cat[..., :c1])
        self._run(skip, x_skip, out=cat[..., c1:])
        return cat

    def forward(self, x):
        x = self._run(self.layer_stack_1, x)
        x = self._csp(self.layer_stack_2_skip, self.layer_stack_2, x, x)              # [x, route]
        x = self._run(self.layer_stack_3, x)
        x = self._csp(self.layer_stack_4, self.layer_stack_4_skip, x, x)
        x = self._run(self.layer_stack_5, x)
        x = self._csp(self.layer_stack_6, self.layer_stack_6_skip, x, x)
        route_1 = x = self._run(self.layer_stack_7, x)
        x = self._run(self.layer_stack_8, x)
        x = self._csp(self.layer_stack_9, self.layer_stack_9_skip, x, x)
        route_2 = x = self._run(self.layer_stack_10, x)
        x = self._run(self.layer_stack_11, x)
        x = self._csp(self.layer_stack_12, self.layer_stack_12_skip, x, x)
        x = self._run(self.layer_stack_13, x)
        x = spp_concat(x)
        x = self._run(self.layer_stack_14, x)
        return route_1, route_2, x
