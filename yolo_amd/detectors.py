"""YOLOv3 detection head, same interface as the reference's detectors.py."""
import torch
from torch import nn

from .layers import YOLOConv, upsample2x_concat


class YoloV3(nn.Module):
    """21-conv YOLOv3 head (detectors.py:4-77).

    ``YoloV3(num_classes, **kwargs)(scale_1, scale_2, scale_3, training=False)``
    returns the raw conv maps (sbbox [B,S/8,S/8,3(C+5)], mbbox [B,S/16,..],
    lbbox [B,S/32,..]), small stride first (detectors.py:77).  Unlike canonical
    YOLOv3 there is no channel-reducing conv before an upsample and the /8 and /16
    branches keep 512/1024 channels (detectors.py:23-44).  The three output convs are
    1x1, linear, with bias (detectors.py:20,33,43).  ``training`` is accepted and
    unused, as in the reference.

    kwargs: name, dtype (compute type), seed, out_dtype (element type of the three
    output maps; default float32 so the decode that follows sees full-precision
    logits even from a bf16 network).
    """

    def __init__(self, num_classes, **kwargs):
        super().__init__()
        self.name = kwargs.pop("name", None)
        dtype = kwargs.pop("dtype", None)
        seed = kwargs.pop("seed", None)
        self.out_dtype = kwargs.pop("out_dtype", torch.float32)
        if kwargs:
            raise TypeError("YoloV3: unexpected keyword arguments %s" % sorted(kwargs))
        self.num_classes = num_classes
        nout = 3 * (num_classes + 5)
        d = dict(dtype=dtype)

        def five():
            return [YOLOConv((1, 1, 512), **d), YOLOConv((3, 3, 1024), **d), YOLOConv((1, 1, 512), **d),
                    YOLOConv((3, 3, 1024), **d), YOLOConv((1, 1, 512), **d)]

        self.layer_stack_1 = nn.ModuleList(five())
        self.lbbox_layer_stack = nn.ModuleList([
            YOLOConv((3, 3, 1024), **d),
            YOLOConv((1, 1, nout), activation=None, batchNormalization=False, name="lbbox", **d)])
        self.layer_stack_2 = nn.ModuleList(five())
        self.mbbox_layer_stack = nn.ModuleList([
            YOLOConv((3, 3, 1024), **d),
            YOLOConv((1, 1, nout), activation=None, batchNormalization=False, name="mbbox", **d)])
        self.sbbox_layer_stack = nn.ModuleList(five() + [
            YOLOConv((3, 3, 1024), **d),
            YOLOConv((1, 1, nout), activation=None, batchNormalization=False, name="sbbox", **d)])

        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for stack, cin in ((self.layer_stack_1, 1024), (self.lbbox_layer_stack, 512),
                           (self.layer_stack_2, 512 + 512), (self.mbbox_layer_stack, 512),
                           (self.sbbox_layer_stack, 256 + 512)):
            for conv in stack:
                conv.build(cin, generator=gen)
                cin = conv.filters

    def conv_layers(self):
        """All 21 YOLOConv layers in execution (= attribute) order."""
        return (list(self.layer_stack_1) + list(self.lbbox_layer_stack) + list(self.layer_stack_2) +
                list(self.mbbox_layer_stack) + list(self.sbbox_layer_stack))

    def load_reference_params(self, params):
        convs = self.conv_layers()
        if len(params) != len(convs):
            raise ValueError("expected %d parameter dicts, got %d" % (len(convs), len(params)))
        for conv, p in zip(convs, params):
            conv.load_reference_params(p)

    def forward(self, scale_1, scale_2, scale_3, training=False):
        x = scale_3
        for layer in self.layer_stack_1:
            x = layer(x)
        lbbox_output = self.lbbox_layer_stack[0](x)
        lbbox_output = self.lbbox_layer_stack[1](lbbox_output, out_dtype=self.out_dtype)

        x = upsample2x_concat(scale_2, x)                 # Upsample()(x); concat([scale_2, x], -1)
        for layer in self.layer_stack_2:
            x = layer(x)
        mbbox_output = self.mbbox_layer_stack[0](x)
        mbbox_output = self.mbbox_layer_stack[1](mbbox_output, out_dtype=self.out_dtype)

        sbbox_output = upsample2x_concat(scale_1, x)      # Upsample()(x); concat([scale_1, x], -1)
        for layer in self.sbbox_layer_stack[:-1]:
            sbbox_output = layer(sbbox_output)
        sbbox_output = self.sbbox_layer_stack[-1](sbbox_output, out_dtype=self.out_dtype)
        return sbbox_output, mbbox_output, lbbox_output
