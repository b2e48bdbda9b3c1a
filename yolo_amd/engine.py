"""Whole-detector executor: Darknet53 -> YoloV3 -> decode -> filter -> combined NMS.

The reference has no model-assembly class; users compose backbones.Darknet53,
detectors.YoloV3 and utils.bbox_nms by hand (the intended composition is visible at
/root/reference/losses.py:57-64 and /root/reference/utils.py:170-194).  ``Detector``
is that composition with the per-call Python work removed: for each input shape it
plans every activation buffer once, records the 75 launches (73 convs + 2
upsample/concat) as a ``ymx_op`` program, and runs the program with one C call;
decode+filter and NMS are two more calls on preallocated outputs.  Nothing is
allocated in steady state and nothing synchronises.
"""
import ctypes

import torch
from torch import nn

from . import _lib
from .backbones import Darknet53
from .detectors import YoloV3
from .layers import ResidialBlock
from .config import config as _config
from . import utils as _utils

_YOLO = _config["yolo"]


class _Plan(object):
    pass


class Detector(nn.Module):
    def __init__(self, num_classes=80, dtype=torch.bfloat16, seed=0, max_outputs_per_class=32, max_outputs=32,
                 iou_threshold=0.45, score_threshold=0.25):
        super().__init__()
        self.num_classes = num_classes
        self.dtype = dtype
        self.backbone = Darknet53(dtype=dtype, seed=seed)
        self.head = YoloV3(num_classes, dtype=dtype, seed=None if seed is None else seed + 1)
        self.max_outputs_per_class = max_outputs_per_class
        self.max_outputs = max_outputs
        self.iou_threshold = iou_threshold
        self.score_threshold = score_threshold
        self._plans = {}

    def load_reference_params(self, backbone_params, head_params):
        self.backbone.load_reference_params(backbone_params)
        self.head.load_reference_params(head_params)
        self._plans = {}

    @torch.no_grad()
    def calibrate_(self, x, out_std=2.0, seed=0):
        """Data-dependent re-scaling of the random parameters (in place).

        The reference's init (N(0, 0.01^2), /root/reference/layers.py:62) drives every
        activation towards 0 within a few layers, so a randomly initialised detector
        computes on (near-)zeros.  For benchmarks and tests on random weights this
        rescales each kernel, layer by layer on the images ``x``, so that its
        pre-activation has unit standard deviation (0.5 on the second conv of a residual
        block, ``out_std`` on the three linear output convs) and draws small random
        biases.  Runs on the GPU through the same kernels as inference.
        """
        x = self._check_input(x)
        gen = torch.Generator().manual_seed(seed)

        def fit(layer, inp, target, residual=None, out_dtype=None):
            act, layer.act_code = layer.act_code, _lib.ACT_NONE
            layer.bias.zero_()
            z = layer(inp, out_dtype=torch.float32 if layer.in_channels % 8 == 0 else None)
            layer.kernel.mul_(target / float(z.float().std()))
            layer.act_code = act
            if act != _lib.ACT_NONE:
                layer.bias.copy_((0.05 * torch.randn(layer.filters, generator=gen)).to(layer.bias.device))
            return layer(inp, residual=residual, out_dtype=out_dtype)

        def stack(layers, t):
            for layer in layers:
                if isinstance(layer, ResidialBlock):
                    y = fit(layer.layer_stack[0], t, 1.0)
                    t = fit(layer.layer_stack[1], y, 0.5, residual=t)
                else:
                    t = fit(layer, t, 1.0)
            return t

        from .layers import upsample2x_concat
        bb, hd = self.backbone, self.head
        s1 = stack(bb.layer_stack_1, x)
        s2 = stack(bb.layer_stack_2, s1)
        s3 = stack(bb.layer_stack_3, s2)
        t = stack(hd.layer_stack_1, s3)
        fit(hd.lbbox_layer_stack[1], fit(hd.lbbox_layer_stack[0], t, 1.0), out_std, out_dtype=hd.out_dtype)
        t = stack(hd.layer_stack_2, upsample2x_concat(s2, t))
        fit(hd.mbbox_layer_stack[1], fit(hd.mbbox_layer_stack[0], t, 1.0), out_std, out_dtype=hd.out_dtype)
        t = stack(list(hd.sbbox_layer_stack)[:-1], upsample2x_concat(s1, t))
        fit(hd.sbbox_layer_stack[-1], t, out_std, out_dtype=hd.out_dtype)
        self._plans = {}
        return self

    # ------------------------------------------------------------------ planning
    def _plan(self, B, S, x_dtype, device):
        key = (B, S, x_dtype, str(device))
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        cdt = self.dtype
        code = _lib.dtype_code(cdt)
        ops = []
        keep = []                       # tensors the program points at

        def buf(h, w, c, dtype=cdt):
            t = torch.empty((B, h, w, c), device=device, dtype=dtype)
            keep.append(t)
            return t

        def conv(layer, x, h, w, residual=None, x_code=code, out_dtype=cdt, pad_out=False):
            wp, scale, shift, _src = layer.packed(cdt)
            oh, ow = layer.output_hw(h, w)
            cout = layer.filters
            if pad_out and cout % 8:
                # Output maps whose channel count is not a multiple of 8 (255 for 80 classes)
                # are computed as the next multiple: the packed weights already hold zero rows
                # up to a multiple of 128, scale/shift get zeros, so the extra channels are
                # exact zeros.  Rows are then 16-byte aligned and the convolution takes its
                # vector store path; the caller sees the first `filters` channels as a view.
                cout = (cout + 7) // 8 * 8
                extra = cout - layer.filters
                if scale is not None:
                    scale = torch.cat([scale, scale.new_zeros(extra)])
                if shift is not None:
                    shift = torch.cat([shift, shift.new_zeros(extra)])
            keep.extend([wp, scale, shift])
            y = buf(oh, ow, cout, out_dtype)
            o = _lib.ymx_op()
            o.kind = 1
            o.B, o.H, o.W, o.Cin, o.Cout = B, h, w, layer.in_channels, cout
            o.k, o.stride, o.act = layer.kernel_size, 2 if layer.downsample else 1, layer.act_code
            o.dtype, o.x_dtype, o.out_dtype = code, x_code, _lib.dtype_code(out_dtype)
            o.x_ld = o.res_ld = o.y_ld = 0
            o.x = None if x is None else x.data_ptr()
            o.w = wp.data_ptr()
            o.scale = None if scale is None else scale.data_ptr()
            o.shift = None if shift is None else shift.data_ptr()
            o.residual = None if residual is None else residual.data_ptr()
            o.y = y.data_ptr()
            if layer.act_fn is not None:
                raise NotImplementedError("Detector: callable activations are not supported in the planned path")
            ops.append(o)
            return y, oh, ow

        def upcat(skip, x, h, w, c1, c2):
            y = buf(2 * h, 2 * w, c1 + c2)
            o = _lib.ymx_op()
            o.kind = 2
            o.B, o.H, o.W, o.C1, o.C2, o.dtype = B, h, w, c1, c2, code
            o.residual, o.x, o.y = skip.data_ptr(), x.data_ptr(), y.data_ptr()
            ops.append(o)
            return y

        def run_stack(stack, x, h, w, first_x_code=code):
            xc = first_x_code
            for layer in stack:
                if isinstance(layer, ResidialBlock):
                    y, _, _ = conv(layer.layer_stack[0], x, h, w)
                    x, h, w = conv(layer.layer_stack[1], y, h, w, residual=x)
                else:
                    x, h, w = conv(layer, x, h, w, x_code=xc)
                xc = code
            return x, h, w

        bb, hd = self.backbone, self.head
        # op 0 reads the caller's image tensor; its pointer is patched on every call
        s1, h1, w1 = run_stack(bb.layer_stack_1, None, S, S, first_x_code=_lib.dtype_code(x_dtype))
        s2, h2, w2 = run_stack(bb.layer_stack_2, s1, h1, w1)
        s3, h3, w3 = run_stack(bb.layer_stack_3, s2, h2, w2)
        x, _, _ = run_stack(hd.layer_stack_1, s3, h3, w3)
        l, _, _ = conv(hd.lbbox_layer_stack[0], x, h3, w3)
        l, _, _ = conv(hd.lbbox_layer_stack[1], l, h3, w3, out_dtype=hd.out_dtype, pad_out=True)
        x = upcat(s2, x, h3, w3, s2.shape[-1], x.shape[-1])
        x, _, _ = run_stack(hd.layer_stack_2, x, h2, w2)
        m, _, _ = conv(hd.mbbox_layer_stack[0], x, h2, w2)
        m, _, _ = conv(hd.mbbox_layer_stack[1], m, h2, w2, out_dtype=hd.out_dtype, pad_out=True)
        s = upcat(s1, x, h2, w2, s1.shape[-1], x.shape[-1])
        s, _, _ = run_stack(list(hd.sbbox_layer_stack)[:-1], s, h1, w1)
        s, _, _ = conv(hd.sbbox_layer_stack[-1], s, h1, w1, out_dtype=hd.out_dtype, pad_out=True)

        plan = _Plan()
        plan.ops = (_lib.ymx_op * len(ops))(*ops)
        plan.n_ops = len(ops)
        plan.keep = keep
        C = self.num_classes
        padded = (s, m, l)                                  # [B,S,S,cell_ld] buffers the convs write
        plan.heads = tuple(t[..., :3 * (5 + C)] for t in padded)    # what callers see: [B,S,S,3*(5+C)]
        plan.head_dtype = s.dtype
        plan.first_is_direct = True
        # post-processing buffers
        anc = _utils.default_anchors()
        plan.scales = _utils._Scales(padded, _YOLO["strides"], _YOLO["xyscale"], anc,
                                     cell_ld=[t.shape[-1] for t in padded])
        cap = _utils._round4(plan.scales.N)
        plan.cap = cap
        f32, i32 = torch.float32, torch.int32
        plan.boxes = torch.empty((B, cap, 4), device=device, dtype=f32)
        plan.scores_cm = torch.empty((B, C, cap), device=device, dtype=f32)
        plan.src_idx = torch.empty((B, cap), device=device, dtype=i32)
        plan.counts = torch.empty((B,), device=device, dtype=i32)
        L = _lib.lib()
        plan.dec_ws_bytes = L.ymx_decode_workspace_bytes(B, plan.scales.n, plan.scales.S)
        plan.dec_ws = torch.empty((plan.dec_ws_bytes,), device=device, dtype=torch.uint8)
        T = self.max_outputs
        plan.out_boxes = torch.empty((B, T, 4), device=device, dtype=f32)
        plan.out_scores = torch.empty((B, T), device=device, dtype=f32)
        plan.out_classes = torch.empty((B, T), device=device, dtype=f32)
        plan.out_valid = torch.empty((B,), device=device, dtype=i32)
        plan.out_keep = torch.empty((B, T), device=device, dtype=i32)
        plan.nms_ws_bytes = L.ymx_nms_workspace_bytes(B, C, self.max_outputs_per_class)
        plan.nms_ws = torch.empty((max(plan.nms_ws_bytes, 16),), device=device, dtype=torch.uint8)
        self._plans[key] = plan
        return plan

    # ------------------------------------------------------------------ execution
    def _check_input(self, x):
        _lib.require_gpu(x)
        if x.dim() != 4 or x.shape[-1] != 3 or x.shape[1] != x.shape[2]:
            raise ValueError("expected square NHWC images [B,S,S,3], got %s" % (tuple(x.shape),))
        if x.shape[1] % 32:
            raise ValueError("image side must be a multiple of 32, got %d" % x.shape[1])
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("images must be float32 or bfloat16")
        return x.contiguous()

    def forward_heads(self, x):
        """images [B,S,S,3] -> (sbbox, mbbox, lbbox) raw conv maps (plan-owned buffers)."""
        x = self._check_input(x)
        plan = self._plan(x.shape[0], x.shape[1], x.dtype, x.device)
        plan.ops[0].x = x.data_ptr()
        _lib.check(_lib.lib().ymx_run_program(plan.ops, plan.n_ops, _lib.stream()))
        return plan.heads

    def postprocess(self, plan, image_size):
        L = _lib.lib()
        sc = plan.scales
        B, C = plan.boxes.shape[0], self.num_classes
        _lib.check(L.ymx_decode_filter_strided(sc.ptrs, sc.cell_ld, sc.n, sc.S, B, C, _lib.dtype_code(plan.head_dtype),
                                               sc.anchors, sc.strides, sc.xyscale, 1, float(self.score_threshold),
                                               float(image_size), _lib.ptr(plan.boxes), _lib.ptr(plan.scores_cm),
                                               _lib.ptr(plan.src_idx), _lib.ptr(plan.counts), plan.cap,
                                               _lib.ptr(plan.dec_ws), plan.dec_ws_bytes, _lib.stream()))
        _lib.check(L.ymx_nms_combined(_lib.ptr(plan.boxes), _lib.ptr(plan.scores_cm), _lib.ptr(plan.counts),
                                      _lib.ptr(plan.src_idx), B, plan.cap, C, int(self.max_outputs_per_class),
                                      int(self.max_outputs), float(self.iou_threshold), float(self.score_threshold),
                                      _lib.ptr(plan.out_boxes), _lib.ptr(plan.out_scores), _lib.ptr(plan.out_classes),
                                      _lib.ptr(plan.out_valid), _lib.ptr(plan.out_keep), _lib.ptr(plan.nms_ws),
                                      plan.nms_ws_bytes, _lib.stream()))
        return _utils.Detections(plan.out_boxes, plan.out_scores, plan.out_classes, plan.out_valid, plan.out_keep)

    def forward(self, x, image_size=None):
        """images -> (boxes [B,T,4], scores [B,T], classes [B,T], valid [B]) with .keep_idx.

        The returned tensors are the plan's output buffers: they are overwritten by
        the next call with the same input shape."""
        x = self._check_input(x)
        plan = self._plan(x.shape[0], x.shape[1], x.dtype, x.device)
        plan.ops[0].x = x.data_ptr()
        _lib.check(_lib.lib().ymx_run_program(plan.ops, plan.n_ops, _lib.stream()))
        return self.postprocess(plan, x.shape[1] if image_size is None else image_size)
