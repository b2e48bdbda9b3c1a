"""Back-propagation through the convolution blocks (bf16 networks).

The reference trains by letting TensorFlow differentiate YOLOConv / ResidialBlock / Upsample
(/root/reference/layers.py:33-106) under ``model.fit``.  Here ``torch.autograd`` only keeps the
graph (which tensor feeds which, summing gradients where a tensor is used twice, e.g. the
shortcut of a ResidialBlock); every forward and backward step of a block is a launch of
libymx:

  forward   z = conv(x) + bias            } ymx_conv2d_fused_z: both outputs of one launch
            y = act(z) + shortcut         } (z is kept for the derivative)
  backward  dz = dy * act'(z)             ymx_act_backward
            dbias = sum_p dz              ymx_bias_grad
            dkernel = x (*) dz            ymx_conv_wgrad (HWIO fp32, the kernel's own layout)
            dx = conv(dz, kernel flipped and transposed)      the forward kernels again (stride 2:
                                                              a 2x2 window, depth-to-space outputs)
            dshortcut = dy

Use::

    with yolo_amd.training.backprop():
        heads = head(*backbone(images))
        loss = sum(f(t, o) for f, t, o in zip(loss_fns, targets, heads[::-1]))
    loss.backward()              # .grad of every kernel and bias, fp32

Outside the context (the default) the layers run the fused inference kernels and build no
graph.  A YOLOConv with batch normalisation (layers.py:65-68) runs as a linear convolution
followed by ``batch_norm_act``: batch statistics, gamma / beta, activation and shortcut, with
the moving statistics updated as Keras does (frozen layers, ``trainable = False``, keep
normalising with the moving statistics, layers.py:11-15).
"""
import threading

import torch

from . import _lib

_state = threading.local()


def enabled():
    return getattr(_state, "on", False) and torch.is_grad_enabled()


class backprop(object):
    """Context manager: YOLOConv and upsample2x_concat record a differentiable graph."""

    def __enter__(self):
        self._prev = getattr(_state, "on", False)
        _state.on = True
        return self

    def __exit__(self, *exc):
        _state.on = self._prev
        return False


# Tuning options (names of ymx_set_option) for the launches of the backward pass.  libymx keeps
# its options per calling thread and torch.autograd runs backward() on a worker thread of its
# own, so an option set by the caller's thread never reaches those launches: they are listed
# here instead and applied by the backward step itself, on whatever thread it runs on.
options = {}
# If set to a list, every weight-gradient launch of the backward pass appends the label of the
# kernel it takes (ymx_conv_wgrad_kernel_label): what tests and timing tools read.
trace = None


def _apply_options():
    L = _lib.lib()
    for name, value in options.items():
        _lib.check(L.ymx_set_option(name.encode(), int(value)))


class L2Regulariser(object):
    """The kernel regulariser of the reference's convolutions: every YOLOConv adds
    ``coefficient * sum(kernel ** 2)`` to the loss Keras minimises (kernel_regularizer =
    l2(0.0005), /root/reference/layers.py:61; biases are not regularised), i.e.
    ``2 * coefficient * kernel`` to each kernel's gradient.

        reg = L2Regulariser(model)            # every YOLOConv kernel below `model`
        loss.backward()
        reg.add_gradients_()                  # .grad += 2 * coefficient * kernel, one fused launch per chunk of tensors
        total = loss + reg.penalty()          # the number Keras reports

    With data-parallel gradient averaging call add_gradients_() after the averaging (the term
    is the same on every rank)."""

    def __init__(self, model_or_kernels, coefficient=0.0005):
        from .layers import YOLOConv
        if isinstance(model_or_kernels, torch.nn.Module):
            self.kernels = [m.kernel for m in model_or_kernels.modules() if isinstance(m, YOLOConv) and m.in_channels is not None]
        else:
            self.kernels = list(model_or_kernels)
        self.coefficient = float(coefficient)

    def penalty(self):
        with torch.no_grad():
            norms = torch._foreach_norm([k.detach() for k in self.kernels])
            return self.coefficient * torch.stack(norms).double().square().sum()

    def add_gradients_(self):
        ks = [k for k in self.kernels if k.grad is not None]
        with torch.no_grad():
            torch._foreach_add_([k.grad for k in ks], [k.detach() for k in ks], alpha=2.0 * self.coefficient)


overlap_wgrad = False       # True: weight gradients on a second HIP stream, next to the input gradients (gained 0.3-0.4 ms
                            # per step while the input gradients left the GPU slack; measured 0.5 ms slower since they do not)
_side = {}


def _side_stream(device):
    key = (threading.get_ident(), device.index if device.index is not None else torch.cuda.current_device())
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=device)
    return st


def _slot(param):
    """(GradientBuckets, view of the parameter's place in its bucket) or None.  Only while the
    parameter's .grad still IS that view: a caller who went back to plain gradients (``p.grad =
    None``) gets them from autograd as usual."""
    slot = getattr(param, "_grad_bucket", None)
    if slot is None or param.grad is None or param.grad.data_ptr() != slot[1].data_ptr():
        return None
    return slot


def _empty(shape, like, dtype=None):
    return torch.empty(shape, device=like.device, dtype=dtype or like.dtype)


def _pad_channels(t, mult=8):
    """[..., C] -> [..., C rounded up to a multiple of mult] (zeros), contiguous."""
    c = t.shape[-1]
    cp = -(-c // mult) * mult
    if cp == c:
        return t.contiguous()
    out = torch.zeros(t.shape[:-1] + (cp,), device=t.device, dtype=t.dtype)
    out[..., :c] = t
    return out


def _conv(x, wp, shift, cin, cout, k, stride, cdt, odt, residual=None):
    b, h, w, _ = x.shape
    oh, ow = (h, w) if stride == 1 else ((h + 1 - k) // 2 + 1, (w + 1 - k) // 2 + 1)
    y = _empty((b, oh, ow, cout), x, odt)
    code = _lib.dtype_code(cdt)
    _lib.launch("ymx_conv2d_fused_ex", x, _lib.ptr(x), _lib.ptr(wp), None, _lib.ptr(shift), _lib.ptr(residual), _lib.ptr(y), b, h, w, cin, cout,
                k, stride, _lib.ACT_NONE, code, _lib.dtype_code(x.dtype), _lib.dtype_code(odt), 0, 0, 0)
    return y


class GradLink(object):
    """Joins the two blocks of a ResidialBlock in the backward pass: the block that adds the
    shortcut leaves the shortcut's gradient (its dy) here instead of handing it to autograd,
    and the block that produced the shortcut's source adds it inside its input-gradient
    convolution (the kernel's shortcut input), so x + f(x) costs no separate addition."""

    def __init__(self):
        self.pending = None


class _ConvBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, shortcut, layer, link=None):
        cdt = layer.compute_dtype or x.dtype
        if cdt != torch.bfloat16:
            raise ValueError("yolo_amd.training: back-propagation is provided for bfloat16 networks")
        k, cin, cout = layer.kernel_size, layer.in_channels, layer.filters
        stride = 2 if layer.downsample else 1
        act = _lib.ACT_NONE if layer.batch_normalization else layer.act_code      # normalisation and activation follow separately
        ctx.act = act
        x_dtype_in = x.dtype
        x = x.contiguous()
        if cin % 8 == 0 and x.dtype != cdt:
            x = x.to(cdt)
        entry = layer.packed_training(cdt)
        wp, shift = entry[0], entry[1]
        sc = None if shortcut is None else shortcut.to(cdt).contiguous()
        if act != _lib.ACT_NONE:
            b, h, w, _ = x.shape
            oh, ow = layer.output_hw(h, w)
            z = _empty((b, oh, ow, cout), x, cdt)
            y = torch.empty_like(z)
            _lib.launch("ymx_conv2d_fused_z", x, _lib.ptr(x), _lib.ptr(wp), None, _lib.ptr(shift), _lib.ptr(sc), _lib.ptr(y), _lib.ptr(z),
                        b, h, w, cin, cout, k, stride, act, _lib.dtype_code(x.dtype))
        else:
            z = _conv(x, wp, shift, cin, cout, k, stride, cdt, cdt)
            if sc is not None:
                y = torch.empty_like(z)
                _lib.launch("ymx_act_apply", z, _lib.ptr(z), _lib.ptr(sc), _lib.ptr(y), z.numel(), act, _lib.dtype_code(cdt))
            else:
                y = z
        ctx.save_for_backward(x, z, kernel)
        ctx.layer = layer
        ctx.has_shortcut = shortcut is not None
        ctx.shortcut_dtype = None if shortcut is None else shortcut.dtype
        ctx.x_dtype = x_dtype_in
        ctx.link = link
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, kernel = ctx.saved_tensors
        layer = ctx.layer
        act = ctx.act
        cdt = torch.bfloat16
        code = _lib.dtype_code(cdt)
        k, cin, cout = layer.kernel_size, layer.in_channels, layer.filters
        stride = 2 if layer.downsample else 1
        b, h, w, _ = x.shape
        oh, ow = z.shape[1], z.shape[2]
        dy = dy.to(cdt).contiguous()
        L = _lib.lib()
        _apply_options()
        npix = b * oh * ow
        dbias = None
        # gradient buckets (yolo_amd.dist.GradientBuckets): the kernels write a parameter's
        # gradient straight into its place in the bucket and report it ready, which starts the
        # bucket's all-reduce as soon as its last gradient is there; autograd gets None
        kslot = _slot(layer.kernel) if ctx.needs_input_grad[1] else None
        bslot = _slot(layer.bias) if ctx.needs_input_grad[2] else None
        if act != _lib.ACT_NONE:
            dz = torch.empty_like(z)
            nb = L.ymx_act_backward_bias_workspace_bytes(npix, cout, act, code) if ctx.needs_input_grad[2] else 0
            if nb:                       # derivative of the activation and the bias gradient in one pass
                dbias = bslot[1] if bslot is not None else torch.empty((cout,), device=x.device, dtype=torch.float32)
                ws = torch.empty((nb,), device=x.device, dtype=torch.uint8)
                _lib.launch("ymx_act_backward_bias", z, _lib.ptr(z), _lib.ptr(dy), _lib.ptr(dz), npix, cout, act, code,
                            _lib.ptr(dbias), _lib.ptr(ws), nb)
            else:
                _lib.launch("ymx_act_backward", z, _lib.ptr(z), _lib.ptr(dy), _lib.ptr(dz), z.numel(), act, code)
        else:
            dz = dy
        # channel counts the MFMA kernels take (multiples of 8): zero padding changes nothing
        dzp = _pad_channels(dz)
        coutp = dzp.shape[-1]
        if ctx.needs_input_grad[2] and dbias is None:
            dbp = torch.empty((coutp,), device=x.device, dtype=torch.float32)
            nb = L.ymx_bias_grad_workspace_bytes(npix, coutp)
            ws = torch.empty((nb,), device=x.device, dtype=torch.uint8)
            _lib.launch("ymx_bias_grad", dzp, _lib.ptr(dzp), npix, coutp, 0, code, _lib.ptr(dbp), _lib.ptr(ws), nb)
            dbias = dbp[:cout]
            if bslot is not None:
                bslot[1].copy_(dbias)
        if bslot is not None:
            bslot[0].ready(layer.bias)
            dbias = None
        dkernel = None
        side = None
        if ctx.needs_input_grad[1]:
            # the first layer (3 input channels) has a kernel of its own that reads the image as it is
            nbf = 0
            if k == 3 and stride == 1 and cin <= 3 and x.dtype in (torch.float32, torch.bfloat16):
                nbf = L.ymx_conv_wgrad_first_workspace_bytes(b, h, w, cin, coutp)
            if nbf:
                xp, cinp, nb = x, cin, nbf
            else:
                xp = _pad_channels(x.to(cdt))
                cinp = xp.shape[-1]
                nb = L.ymx_conv_wgrad_workspace_bytes(b, h, w, cinp, coutp, k, stride)
            direct = kslot is not None and cinp == cin and coutp == cout
            dwp = kslot[1] if direct else torch.empty((k, k, cinp, coutp), device=x.device, dtype=torch.float32)
            ws = torch.empty((max(nb, 16),), device=x.device, dtype=torch.uint8)
            if trace is not None:
                trace.append("wgrad_first" if nbf else L.ymx_conv_wgrad_kernel_label(b, h, w, cinp, coutp, k, stride).decode())

            def launch_wgrad():
                if nbf:
                    _lib.launch("ymx_conv_wgrad_first", xp, _lib.ptr(xp), _lib.dtype_code(xp.dtype), _lib.ptr(dzp), 0, _lib.ptr(dwp), b, h, w, cin, coutp,
                                _lib.ptr(ws), nb)
                else:
                    _lib.launch("ymx_conv_wgrad", xp, _lib.ptr(xp), 0, _lib.ptr(dzp), 0, _lib.ptr(dwp), b, h, w, cinp, coutp, k, stride,
                                _lib.ptr(ws), nb)
            # the weight gradient and the input gradient of a block are independent: the former
            # runs on a second stream while this one goes on with the latter
            side = _side_stream(x.device) if (overlap_wgrad and ctx.needs_input_grad[0]) else None
            if side is not None:
                main = torch.cuda.current_stream(x.device)
                side.wait_stream(main)
                for t_ in (xp, dzp, dwp, ws):
                    t_.record_stream(side)
                with torch.cuda.stream(side):
                    launch_wgrad()
            else:
                launch_wgrad()
        dx = None
        if ctx.needs_input_grad[0]:
            # d loss / d x = conv(dz, kernel flipped over the taps, input and output channels swapped)
            wdp = layer.packed_training(cdt)[2]                   # packed with the forward kernel, in the same launch
            if wdp is None:
                raise ValueError("yolo_amd.training: no input gradient for a layer of %d input channels" % cin)
            res = None
            if ctx.link is not None and not ctx.has_shortcut:
                res, ctx.link.pending = ctx.link.pending, None        # the shortcut's gradient, left by the block after this one
                if res is not None and (res.dtype != cdt or tuple(res.shape) != (b, h, w, cin)):
                    raise RuntimeError("yolo_amd.training: shortcut gradient does not match the block's input")
            if stride == 2:
                # a 2x2 convolution over dz whose 4 * cin outputs per pixel land depth-to-space in dx
                if res is not None:
                    raise RuntimeError("yolo_amd.training: a stride-2 block does not start a ResidialBlock")
                even = h % 2 == 0 and w % 2 == 0             # the last row / column of an odd-sized input has no gradient
                dx = (torch.empty if even else torch.zeros)((b, h, w, cin), device=x.device, dtype=cdt)
                _lib.launch("ymx_conv2d_dgrad_s2", dzp, _lib.ptr(dzp), _lib.ptr(wdp), _lib.ptr(dx), b, oh, ow, coutp, cin, h, w)
            else:
                dx = _conv(dzp, wdp, None, coutp, cin, k, 1, cdt, cdt, residual=res)
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            if side is not None:
                torch.cuda.current_stream(x.device).wait_stream(side)
            # (handed over as it is where nothing was padded: autograd keeps a tensor that is not a
            # view as the parameter's .grad without copying it)
            if kslot is not None:
                if not direct:
                    kslot[1].copy_(dwp[:, :, :cin, :cout])
                kslot[0].ready(layer.kernel)
            else:
                dkernel = dwp if (cinp == cin and coutp == cout) else dwp[:, :, :cin, :cout].contiguous()
        dshortcut = None
        if ctx.has_shortcut and ctx.needs_input_grad[3]:
            if ctx.link is not None and ctx.shortcut_dtype == cdt:
                ctx.link.pending = dy                                  # added by the first block's input gradient
            else:
                dshortcut = dy.to(ctx.shortcut_dtype)
        return dx, dkernel, dbias, dshortcut, None, None


def conv_block(layer, x, shortcut=None, link=None):
    """YOLOConv forward that autograd can differentiate (called by YOLOConv.forward inside
    ``backprop()``).  link: the GradLink of the ResidialBlock the layer belongs to."""
    if layer.act_fn is not None:
        raise ValueError("yolo_amd.training: callable activations are not differentiated")
    if layer.batch_normalization:
        u = _ConvBlock.apply(x, layer.kernel, None, None, layer, None)            # linear, no bias (layers.py:61)
        return batch_norm_act(layer.bn, u, layer.act_code, shortcut)
    return _ConvBlock.apply(x, layer.kernel, layer.bias, shortcut, layer, link)


class _BatchNormAct(torch.autograd.Function):
    """y = act(gamma * (u - mean) / sqrt(var + eps) + beta) + shortcut over the last axis of u, mean and
    var the batch's own (trainable layer) or the moving ones (frozen layer: constants)."""

    @staticmethod
    def forward(ctx, u, gamma, beta, shortcut, bn, act):
        if u.dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("batch_norm_act: bfloat16 or float32 tensors")
        u = u.contiguous()
        c = u.shape[-1]
        n = u.numel() // c
        code = _lib.dtype_code(u.dtype)
        L = _lib.lib()
        frozen = not bn.trainable
        if frozen:
            mean, var = bn.moving_mean.float(), bn.moving_variance.float()
        else:
            nb = L.ymx_bn_workspace_bytes(n, c)
            ws = torch.empty((nb,), device=u.device, dtype=torch.uint8)
            sums = torch.empty((2 * c,), device=u.device, dtype=torch.float32)
            _lib.launch("ymx_bn_stats", u, _lib.ptr(u), n, c, code, _lib.ptr(sums), _lib.ptr(ws), nb)
            mean = sums[:c] / n
            var = (sums[c:] / n - mean * mean).clamp_min_(0.0)               # biased, as Keras normalises
            with torch.no_grad():                                             # the moving statistics Keras keeps for inference
                m = bn.momentum
                bn.moving_mean.mul_(m).add_(mean.to(bn.moving_mean.dtype), alpha=1.0 - m)
                bn.moving_variance.mul_(m).add_((var * (n / max(n - 1, 1))).to(bn.moving_variance.dtype), alpha=1.0 - m)
        inv = torch.rsqrt(var + bn.epsilon)
        scale = (gamma.detach().float() * inv).contiguous()
        shift = (beta.detach().float() - mean * scale).contiguous()
        sc = None if shortcut is None else shortcut.to(u.dtype).contiguous()
        y = torch.empty_like(u)
        _lib.launch("ymx_bn_act_forward", u, _lib.ptr(u), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(sc), _lib.ptr(y), n, c, act, code)
        ctx.save_for_backward(u, scale, shift, mean, inv, gamma)
        ctx.act, ctx.frozen = act, frozen
        ctx.shortcut_dtype = None if shortcut is None else shortcut.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        u, scale, shift, mean, inv, gamma = ctx.saved_tensors
        c = u.shape[-1]
        n = u.numel() // c
        code = _lib.dtype_code(u.dtype)
        L = _lib.lib()
        dy = dy.to(u.dtype).contiguous()
        dz = torch.empty_like(u)
        nb = L.ymx_bn_workspace_bytes(n, c)
        ws = torch.empty((nb,), device=u.device, dtype=torch.uint8)
        sums = torch.empty((2 * c,), device=u.device, dtype=torch.float32)
        _lib.launch("ymx_bn_act_backward", u, _lib.ptr(u), _lib.ptr(dy), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(dz), n, c, ctx.act, code,
                    _lib.ptr(sums), _lib.ptr(ws), nb)
        dbeta = sums[:c]
        dgamma = inv * (sums[c:] - mean * dbeta)
        g = gamma.detach().float()
        a = (g * inv).contiguous()
        if ctx.frozen:                                   # constants: du = scale * dz, no parameter gradients through the statistics
            b = torch.zeros_like(a)
            c0 = torch.zeros_like(a)
        else:
            b = (-g * inv * inv * dgamma / n).contiguous()
            c0 = (g * inv * (inv * mean * dgamma - dbeta) / n).contiguous()
        du = None
        if ctx.needs_input_grad[0]:
            du = torch.empty_like(u)
            _lib.launch("ymx_bn_backward_apply", u, _lib.ptr(dz), _lib.ptr(u), _lib.ptr(a), _lib.ptr(b), _lib.ptr(c0), _lib.ptr(du), n, c, code)
        dsc = dy.to(ctx.shortcut_dtype) if (ctx.shortcut_dtype is not None and ctx.needs_input_grad[3]) else None
        return (du, dgamma.to(gamma.dtype) if ctx.needs_input_grad[1] else None, dbeta.clone() if ctx.needs_input_grad[2] else None,
                dsc, None, None)


def batch_norm_act(bn, u, act=_lib.ACT_NONE, shortcut=None):
    """Training-mode call of a yolo_amd.layers.BatchNormalization on u [..., C] (a GPU tensor,
    bfloat16 or float32), with an activation code and a shortcut applied in the same pass."""
    _lib.require_gpu(u)
    return _BatchNormAct.apply(u, bn.gamma, bn.beta, shortcut, bn, act)


class _UpsampleConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, x):
        from .layers import upsample2x_concat
        ctx.c1 = 0 if skip is None else skip.shape[-1]
        ctx.xshape = tuple(x.shape)
        return upsample2x_concat(skip, x, _differentiable=False)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        b, h, w, c2 = ctx.xshape
        c1 = ctx.c1
        dskip = dy[..., :c1].contiguous() if (c1 and ctx.needs_input_grad[0]) else None
        dx = None
        if ctx.needs_input_grad[1]:
            dx = torch.empty(ctx.xshape, device=dy.device, dtype=dy.dtype)
            _lib.launch("ymx_upsample2x_backward", dy, _lib.ptr(dy), _lib.ptr(dx), b, h, w, c2, c1 + c2, c1, _lib.dtype_code(dy.dtype))
        return dskip, dx


def upsample_concat(skip, x):
    return _UpsampleConcat.apply(skip, x)


class _SppConcat(torch.autograd.Function):
    """concat([maxpool13, maxpool9, maxpool5, x]) (backbones.py:254-259) with the argmax of every
    pooled value kept for the backward pass."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        b, h, w, c = x.shape
        y = torch.empty((b, h, w, 4 * c), device=x.device, dtype=x.dtype)
        arg = torch.empty((b, h, w, 3, c), device=x.device, dtype=torch.uint8)
        _lib.launch("ymx_spp_concat_train", x, _lib.ptr(x), _lib.ptr(y), _lib.ptr(arg), b, h, w, c, _lib.dtype_code(x.dtype))
        ctx.save_for_backward(arg)
        ctx.shape = (b, h, w, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        b, h, w, c = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((b, h, w, c), device=dy.device, dtype=dy.dtype)
        _lib.launch("ymx_spp_concat_backward", dy, _lib.ptr(dy), _lib.ptr(arg), _lib.ptr(dx), b, h, w, c, _lib.dtype_code(dy.dtype))
        return dx


def spp_concat(x):
    _lib.require_gpu(x)
    return _SppConcat.apply(x)
