"""Conv blocks of the detector, same names and signatures as the reference's layers.py.

Every class mirrors /root/reference/layers.py (line numbers in each docstring).
Tensors are NHWC on a GPU; kernels are stored HWIO in fp32 like Keras stores them and
packed for the MFMA kernels on first use.  All device work goes through libymx.
"""
import ctypes

import torch
from torch import nn

from . import _lib
from . import training as _training

BN_EPSILON = 1e-3      # Keras BatchNormalization default, used unchanged at layers.py:4


def _empty_like_nhwc(x, channels, h=None, w=None, dtype=None):
    return torch.empty((x.shape[0], h if h is not None else x.shape[1],
                        w if w is not None else x.shape[2], channels),
                       device=x.device, dtype=dtype if dtype is not None else x.dtype)


def _as_nhwc(x):
    if x.dim() != 4:
        raise ValueError("expected an NHWC tensor [B,H,W,C], got shape %s" % (tuple(x.shape),))
    _lib.require_gpu(x)
    return x if x.is_contiguous() else x.contiguous()


def affine_act(x, scale, shift, act):
    """y = act(x*scale[c] + shift[c]) over the last axis (ymx_affine_act)."""
    _lib.require_gpu(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    c = x.shape[-1]
    _lib.launch("ymx_affine_act", x, _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y),
                                         x.numel() // c, c, act, _lib.dtype_code(x.dtype))
    return y


class BatchNormalization(nn.Module):
    """Batch norm over the last axis (layers.py:4-15).

    The reference forces ``training = training and trainable`` and then defers to Keras:
    with the flag off (or a frozen layer) the moving statistics normalise, folded into one
    scale and shift; with it on, the statistics of the batch do, the moving ones are
    updated (momentum 0.99, Keras' default) and the call is differentiable
    (yolo_amd.training).  Inside a YOLOConv the flag is ``yolo_amd.training.backprop()``
    being active, the counterpart of Keras handing ``training=True`` down under model.fit.
    """

    def __init__(self, num_features=None, epsilon=BN_EPSILON, momentum=0.99):
        super().__init__()
        self.epsilon = epsilon
        self.momentum = momentum
        self.trainable = True
        self.num_features = None
        if num_features is not None:
            self.build(num_features)

    def build(self, num_features):
        self.num_features = num_features
        self.gamma = nn.Parameter(torch.ones(num_features))
        self.beta = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("moving_mean", torch.zeros(num_features))
        self.register_buffer("moving_variance", torch.ones(num_features))

    def folded(self):
        """(scale, shift) with y = x*scale + shift, fp32 on the parameters' device."""
        with torch.no_grad():
            scale = self.gamma.float() / torch.sqrt(self.moving_variance.float() + self.epsilon)
            shift = self.beta.float() - self.moving_mean.float() * scale
        return scale.contiguous(), shift.contiguous()

    def forward(self, x, training=False):
        if self.num_features is None:
            self.build(x.shape[-1])
            self.to(x.device)
        if training and self.trainable:
            return _training.batch_norm_act(self, _as_nhwc(x), _lib.ACT_NONE)
        scale, shift = self.folded()
        return affine_act(x, scale, shift, _lib.ACT_NONE)


class Mish(nn.Module):
    """x * tanh(softplus(x)) (layers.py:17-22)."""

    def forward(self, x):
        return affine_act(x, None, None, _lib.ACT_MISH)


class RouteGroup(nn.Module):
    """tf.split(x, groups, axis=-1)[group_id] as a zero-copy view (layers.py:24-31)."""

    def __init__(self, groups, group_id):
        super().__init__()
        self.groups = groups
        self.group_id = group_id

    def forward(self, x):
        c = x.shape[-1]
        if c % self.groups:
            raise ValueError("RouteGroup: %d channels do not split into %d groups" % (c, self.groups))
        return torch.split(x, c // self.groups, dim=-1)[self.group_id]


def upsample2x_concat(skip, x, _differentiable=True):
    """[skip, bilinear_x2(x)] along channels in one kernel (ymx_upsample2x_concat)."""
    if _differentiable and _training.enabled():
        return _training.upsample_concat(skip, x)
    x = _as_nhwc(x)
    b, h, w, c2 = x.shape
    c1 = 0
    if skip is not None:
        skip = _as_nhwc(skip)
        if skip.shape[:3] != (b, 2 * h, 2 * w) or skip.dtype != x.dtype:
            raise ValueError("upsample2x_concat: skip %s does not match 2x of %s" % (tuple(skip.shape), tuple(x.shape)))
        c1 = skip.shape[3]
    y = _empty_like_nhwc(x, c1 + c2, 2 * h, 2 * w)
    _lib.launch("ymx_upsample2x_concat", x, _lib.ptr(skip), _lib.ptr(x), _lib.ptr(y), b, h, w, c1, c2,
                                                _lib.dtype_code(x.dtype))
    return y


def spp_concat(x):
    """concat([maxpool13(x), maxpool9(x), maxpool5(x), x], -1), stride-1 SAME pools, in one
    kernel (ymx_spp_concat; /root/reference/backbones.py:254-259)."""
    if _training.enabled() and x.requires_grad:
        return _training.spp_concat(x)
    x = _as_nhwc(x)
    b, h, w, c = x.shape
    y = _empty_like_nhwc(x, 4 * c)
    _lib.launch("ymx_spp_concat", x, _lib.ptr(x), _lib.ptr(y), b, h, w, c, _lib.dtype_code(x.dtype))
    return y


class Upsample(nn.Module):
    """Bilinear x2, half-pixel centres, edge clamp (layers.py:33-39)."""

    def forward(self, x):
        return upsample2x_concat(None, x)


class YOLOConv(nn.Module):
    """[pad top/left] -> Conv2D -> [BatchNormalization] -> [activation] (layers.py:41-90).

    filters_shape = (kh, kw, Cout).  downsample: stride 2 with one zero row/column on
    the top/left only, VALID (layers.py:46-54); else stride 1 SAME (layers.py:55-57).
    use_bias = not batchNormalization (layers.py:61).  activation: "mish" (default),
    "leaky" (slope 0.1), a callable, or anything falsy for none; any other value
    raises ValueError (layers.py:71-83).  Kernel init N(0, 0.01^2), bias 0
    (layers.py:62-63).  Keras builds the kernel when it first sees an input; so does
    this class (or pass in_channels= / call build()).

    Extra keyword arguments: name (kept), dtype (torch.float32 or torch.bfloat16: the
    compute type; default = the input's type), in_channels.
    """

    def __init__(self, filters_shape, downsample=False, batchNormalization=False, activation="mish", **kwargs):
        super().__init__()
        self.name = kwargs.pop("name", None)
        self.compute_dtype = kwargs.pop("dtype", None)
        in_channels = kwargs.pop("in_channels", None)
        kwargs.pop("trainable", None)
        if kwargs:
            raise TypeError("YOLOConv: unexpected keyword arguments %s" % sorted(kwargs))
        if len(filters_shape) != 3 or filters_shape[0] != filters_shape[1]:
            raise ValueError("filters_shape must be (k, k, filters), got %s" % (tuple(filters_shape),))
        self.kernel_size = int(filters_shape[0])
        self.filters = int(filters_shape[-1])
        self.downsample = bool(downsample)
        self.batch_normalization = bool(batchNormalization)
        self.act_fn = None
        if activation:
            if callable(activation):
                self.act_code = _lib.ACT_NONE
                self.act_fn = activation
            elif activation == "leaky":
                self.act_code = _lib.ACT_LEAKY
            elif activation == "mish":
                self.act_code = _lib.ACT_MISH
            else:
                raise ValueError("Activation should either be a function or one of \"leaky\" or \"mish\".")
        else:
            self.act_code = _lib.ACT_NONE
        self.bn = BatchNormalization(self.filters) if self.batch_normalization else None
        self.in_channels = None
        self._packed = {}
        if in_channels is not None:
            self.build(in_channels)

    # -- parameters -------------------------------------------------------------
    def build(self, in_channels, generator=None):
        k = self.kernel_size
        self.in_channels = int(in_channels)
        self.kernel = nn.Parameter(torch.randn(k, k, self.in_channels, self.filters, generator=generator) * 0.01)
        if not self.batch_normalization:
            self.bias = nn.Parameter(torch.zeros(self.filters))
        self._packed = {}
        return self

    def load_reference_params(self, p):
        """Copy a parameter dict {'kernel', 'bias' | 'gamma','beta','mean','var'}."""
        if self.in_channels is None:
            self.build(p["kernel"].shape[2])
        with torch.no_grad():
            self.kernel.copy_(p["kernel"])
            if self.batch_normalization:
                self.bn.gamma.copy_(p["gamma"]); self.bn.beta.copy_(p["beta"])
                self.bn.moving_mean.copy_(p["mean"]); self.bn.moving_variance.copy_(p["var"])
            else:
                self.bias.copy_(p["bias"])
        self._packed = {}

    def packed(self, dtype):
        """(weights packed for libymx, scale or None, shift) for a compute dtype."""
        dev = self.kernel.device
        key = (dtype, dev, self.kernel._version,
               self.bias._version if not self.batch_normalization else
               (self.bn.gamma._version, self.bn.beta._version, self.bn.moving_mean._version,
                self.bn.moving_variance._version))
        hit = self._packed.get("entry")
        if hit is not None and hit[0] == key:
            return hit[1]
        _lib.require_gpu(self.kernel)
        code = _lib.dtype_code(dtype)
        k, cin, cout = self.kernel_size, self.in_channels, self.filters
        nbytes = _lib.lib().ymx_packed_weight_bytes(k, cin, cout, code)
        wp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        src = self.kernel.detach().float().contiguous()
        _lib.launch("ymx_pack_weights", wp, _lib.ptr(src), _lib.ptr(wp), k, cin, cout, code)
        if self.batch_normalization:
            scale, shift = self.bn.folded()
        else:
            scale, shift = None, self.bias.detach().float().contiguous()
        entry = (wp, scale, shift, src)      # src kept alive until the pack kernel has run
        self._packed["entry"] = (key, entry)
        return entry

    def packed_training(self, dtype):
        """(forward pack, shift, pack of the input-gradient convolution or None) for a training
        step: both layouts come out of one launch (ymx_pack_weights_pair) and are kept until
        the kernel or the bias changes, i.e. for the forward and the backward pass of a step.
        Layers whose input channels do not take the MFMA kernels (the first: 3) have no
        input-gradient pack; stride-2 layers get the 2x2 depth-to-space form
        (ymx_conv2d_dgrad_s2)."""
        bn = self.batch_normalization              # the convolution is then linear: no bias, normalisation follows
        dev = self.kernel.device
        key = (dtype, dev, self.kernel._version, None if bn else self.bias._version)
        hit = self._packed.get("train")
        if hit is not None and hit[0] == key:
            return hit[1]
        k, cin, cout = self.kernel_size, self.in_channels, self.filters
        if cin % 8:
            wp, _, shift, _ = self.packed(dtype)
            entry = (wp, None if bn else shift, None)
        else:
            _lib.require_gpu(self.kernel)
            L = _lib.lib()
            code = _lib.dtype_code(dtype)
            coutp = -(-cout // 8) * 8
            wp = torch.empty(L.ymx_packed_weight_bytes(k, cin, cout, code), dtype=torch.uint8, device=dev)
            stride = 2 if self.downsample else 1
            if stride == 2 and k != 3:
                raise ValueError("YOLOConv.packed_training: stride-2 convolutions are 3x3")
            nb = L.ymx_packed_weight_bytes(2, coutp, 4 * cin, code) if stride == 2 else L.ymx_packed_weight_bytes(k, coutp, cin, code)
            wdp = torch.empty(nb, dtype=torch.uint8, device=dev)
            src = self.kernel.detach().float().contiguous()
            _lib.launch("ymx_pack_weights_pair", wp, _lib.ptr(src), _lib.ptr(wp), _lib.ptr(wdp), k, cin, cout, coutp, stride, code)
            entry = (wp, None if bn else self.bias.detach().float().contiguous(), wdp, src)
        self._packed["train"] = (key, entry)
        return entry

    # -- geometry ---------------------------------------------------------------
    def output_hw(self, h, w):
        if not self.downsample:
            return h, w
        k = self.kernel_size
        return (h + 1 - k) // 2 + 1, (w + 1 - k) // 2 + 1

    # -- forward ----------------------------------------------------------------
    def forward(self, x, residual=None, out_dtype=None, out=None, _grad_link=None):
        """x NHWC -> NHWC.  residual (same shape as the output) is added after the
        activation inside the kernel; out_dtype overrides the output element type; out, if
        given, receives the result: a [B,H',W',Cout] view whose pixels may be spaced wider than
        Cout (a channel slice of a larger NHWC buffer: how a concat is written in place)."""
        x = _as_nhwc(x)
        if self.in_channels is None:
            self.build(x.shape[-1])
            self.to(x.device)
        if x.shape[-1] != self.in_channels:
            raise ValueError("YOLOConv %s: expected %d input channels, got %d" % (self.name or "", self.in_channels, x.shape[-1]))
        if _training.enabled():
            # differentiable path (yolo_amd.training): z = conv(x) kept, y = act(z) + shortcut
            if out is not None:
                raise ValueError("YOLOConv: out= is not supported while back-propagation is recorded")
            y = _training.conv_block(self, x, residual, _grad_link)
            return y if out_dtype in (None, y.dtype) else y.to(out_dtype)
        cdt = self.compute_dtype or x.dtype
        code = _lib.dtype_code(cdt)
        k, cin, cout = self.kernel_size, self.in_channels, self.filters
        mfma = (cin % 8 == 0) if code == _lib.BF16 else (cin % 4 == 0)
        if mfma and x.dtype != cdt:
            x = x.to(cdt)
        odt = out_dtype or cdt
        wp, scale, shift, _ = self.packed(cdt)
        b, h, w, _c = x.shape
        oh, ow = self.output_hw(h, w)
        y_ld = 0
        if out is not None:
            if self.act_fn is not None:
                raise ValueError("YOLOConv: out= is not supported with a callable activation")
            if (tuple(out.shape) != (b, oh, ow, cout) or out.dtype != odt or out.device != x.device or out.stride(3) != 1 or
                    out.stride(1) != ow * out.stride(2) or out.stride(0) != oh * ow * out.stride(2)):
                raise ValueError("YOLOConv: out must be a [%d,%d,%d,%d] %s view with dense pixels" % (b, oh, ow, cout, odt))
            y, y_ld = out, out.stride(2)
        else:
            y = _empty_like_nhwc(x, cout, oh, ow, dtype=odt)
        fuse_res = residual is not None and self.act_fn is None
        if fuse_res:
            residual = _as_nhwc(residual)
            want = cdt if mfma else odt
            if residual.shape != y.shape:
                raise ValueError("YOLOConv: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
            if residual.dtype != want:
                residual = residual.to(want)
        _lib.launch("ymx_conv2d_fused_ex", x, _lib.ptr(x), _lib.ptr(wp), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual) if fuse_res else None,
            _lib.ptr(y), b, h, w, cin, cout, k, 2 if self.downsample else 1, self.act_code, code,
            _lib.dtype_code(x.dtype), _lib.dtype_code(odt), 0, 0, y_ld)
        if self.act_fn is not None:
            y = self.act_fn(y)
            if residual is not None:
                y = residual + y
        return y


class ResidialBlock(nn.Module):
    """x + conv3x3(conv1x1(x)); the add follows the second activation (layers.py:92-106).

    The spelling is the reference's.  The add is fused into the 3x3 conv's epilogue.
    """

    def __init__(self, filter_num1, filter_num2, activation="mish", **kwargs):
        super().__init__()
        dtype = kwargs.pop("dtype", None)
        if kwargs:
            raise TypeError("ResidialBlock: unexpected keyword arguments %s" % sorted(kwargs))
        self.layer_stack = nn.ModuleList([
            YOLOConv(filters_shape=(1, 1, filter_num1), activation=activation, dtype=dtype),
            YOLOConv(filters_shape=(3, 3, filter_num2), activation=activation, dtype=dtype),
        ])

    def forward(self, x, out=None):
        short_cut = x
        if _training.enabled() and torch.is_tensor(x) and x.requires_grad and x.dtype == torch.bfloat16:
            # both blocks see x: the second hands the shortcut's gradient to the first (training.GradLink)
            link = _training.GradLink()
            x = self.layer_stack[0](x, _grad_link=link)
            return self.layer_stack[1](x, residual=short_cut, out=out, _grad_link=link)
        x = self.layer_stack[0](x)
        return self.layer_stack[1](x, residual=short_cut, out=out)
