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


# Tile threshold (see ymx_op.min_tiles) for programs that share the GPU with another
# stream's program: with the chip already busy, fewer and larger tiles win.
_SHARED_MIN_TILES = 80


class Detector(nn.Module):
    def __init__(self, num_classes=80, dtype=torch.bfloat16, seed=0, max_outputs_per_class=32, max_outputs=32,
                 iou_threshold=0.45, score_threshold=0.25, split="auto", fuse_stem=True):
        """``split``: how a batch is cut into sub-batches that run concurrently on separate
        HIP streams (the tail of one sub-batch's launch overlaps the other's work):
        "auto" = time a few cuts once per input shape and keep the fastest, "off" = never,
        or a tuple of sub-batch sizes used for batches of exactly their sum.
        fuse_stem: the backbone's first two layers (backbones.py:11-12) run as one launch that
        keeps the 32-channel full-resolution map in LDS, and so do its first three ResidialBlocks
        (backbones.py:13-16) with the narrow map between their two convolutions (same bits as the
        separate launches; those maps are then never written, so per-layer hooks on the planned
        path need fuse_stem=False)."""
        super().__init__()
        if not (split in ("auto", "off") or (isinstance(split, (tuple, list)) and split and
                                            all(isinstance(n, int) and n > 0 for n in split))):
            raise ValueError("split must be 'auto', 'off' or a tuple of positive sub-batch sizes")
        self.split = split
        self.fuse_stem = bool(fuse_stem)
        self._side_streams = {}             # device -> streams of the sub-batches after the first
        self.num_classes = num_classes
        self.dtype = dtype
        self.backbone = Darknet53(dtype=dtype, seed=seed)
        self.head = YoloV3(num_classes, dtype=dtype, seed=None if seed is None else seed + 1)
        self.max_outputs_per_class = max_outputs_per_class
        self.max_outputs = max_outputs
        self.iou_threshold = iou_threshold
        self.score_threshold = score_threshold
        self._plans = {}
        self._tensors = None                # parameters and buffers whose versions a plan depends on

    def load_reference_params(self, backbone_params, head_params):
        self.backbone.load_reference_params(backbone_params)
        self.head.load_reference_params(head_params)
        self._plans = {}
        self._tensors = None

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float(): parameters move, every plan points at stale copies
        out = super()._apply(fn, *args, **kwargs)
        self._plans = {}
        self._tensors = None
        return out

    def _versions(self):
        """Version counters of every parameter and buffer.  A plan holds pointers to weights
        packed from the parameters as they were when it was built; any in-place update since
        (load_state_dict, an optimiser step, copy_) bumps a counter and the plan is rebuilt."""
        ts = self._tensors
        if ts is None:
            ts = self._tensors = list(self.parameters()) + list(self.buffers())
        return tuple(t._version for t in ts)

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
    def _build_part(self, plan, n, start, S, x_dtype, device, min_tiles):
        """Program, activation buffers and post-processing buffers for images
        [start, start + n) of the batch; outputs land in slices of the plan's buffers."""
        cdt = self.dtype
        code = _lib.dtype_code(cdt)
        ops = []
        keep = []                       # activation buffers the program points at
        sites = []                      # (op index, layer, zero output channels appended) of every conv

        def buf(h, w, c, dtype=cdt):
            t = torch.empty((n, h, w, c), device=device, dtype=dtype)
            keep.append(t)
            return t

        def ld(t):
            """Pixel stride (elements) of a tensor that is a channel slice of a wider buffer; 0 = dense."""
            return 0 if t is None or t.stride(2) == t.shape[3] else t.stride(2)

        def conv(layer, x, h, w, residual=None, x_code=code, out_dtype=cdt, head=None, out=None):
            oh, ow = layer.output_hw(h, w)
            cout = layer.filters
            if head is not None and cout % 8:
                # Output maps whose channel count is not a multiple of 8 (255 for 80 classes)
                # are computed as the next multiple: the packed weights already hold zero rows
                # up to a multiple of 128, scale/shift get zeros, so the extra channels are
                # exact zeros.  Rows are then 16-byte aligned and the convolution takes its
                # vector store path; the caller sees the first `filters` channels as a view.
                cout = (cout + 7) // 8 * 8
            sites.append((len(ops), layer, cout - layer.filters))
            if out is not None:
                y = out                 # a channel slice of a concat buffer: the concat costs no copy
                if tuple(y.shape) != (n, oh, ow, cout) or y.dtype != out_dtype:
                    raise RuntimeError("Detector: concat slice %s does not match the layer's output" % (tuple(y.shape),))
            elif head is None:
                y = buf(oh, ow, cout, out_dtype)
            else:
                # the three output maps are whole-batch buffers owned by the plan
                if plan.padded[head] is None:
                    plan.padded[head] = torch.empty((plan.B, oh, ow, cout), device=device, dtype=out_dtype)
                y = plan.padded[head][start:start + n]
            o = _lib.ymx_op()
            o.kind = 1
            o.B, o.H, o.W, o.Cin, o.Cout = n, h, w, layer.in_channels, cout
            o.k, o.stride, o.act = layer.kernel_size, 2 if layer.downsample else 1, layer.act_code
            o.dtype, o.x_dtype, o.out_dtype = code, x_code, _lib.dtype_code(out_dtype)
            o.x_ld, o.res_ld, o.y_ld = ld(x), ld(residual), ld(y)
            o.min_tiles = min_tiles
            o.x = None if x is None else x.data_ptr()
            o.residual = None if residual is None else residual.data_ptr()
            o.y = y.data_ptr()
            if layer.act_fn is not None:
                raise NotImplementedError("Detector: callable activations are not supported in the planned path")
            ops.append(o)
            return y, oh, ow

        def upcat(cat, x, h, w, c1, c2):
            """cat [n, 2h, 2w, c1 + c2] already holds the skip tensor in its first c1 channels (its producer
            wrote it there); the resized x goes into the rest."""
            o = _lib.ymx_op()
            o.kind = 2
            o.B, o.H, o.W, o.C1, o.C2, o.dtype = n, h, w, c1, c2, code
            o.residual, o.x, o.y = None, x.data_ptr(), cat.data_ptr()
            ops.append(o)
            return cat

        def run_stack(stack, x, h, w, first_x_code=code, out_last=None):
            """out_last: where the stack's last convolution writes (a channel slice of a concat buffer)."""
            xc = first_x_code
            stack = list(stack)
            for li, layer in enumerate(stack):
                out = out_last if li == len(stack) - 1 else None
                if isinstance(layer, ResidialBlock):
                    first = len(ops)
                    y, _, _ = conv(layer.layer_stack[0], x, h, w)
                    x, h, w = conv(layer.layer_stack[1], y, h, w, residual=x, out=out)
                    a, b = ops[first], ops[first + 1]
                    if self.fuse_stem and (a.Cin, a.Cout, a.k, b.Cout, b.k) in ((64, 32, 1, 64, 3), (128, 64, 1, 128, 3)) \
                            and a.act == b.act:
                        # ResidialBlock(32, 64) / (64, 128): the narrow map is read by the 3x3 convolution alone,
                        # so ymx_run_program may run the block as one launch (ymx_conv_resblock_fused)
                        a.C1 = 1
                else:
                    x, h, w = conv(layer, x, h, w, x_code=xc, out=out)
                xc = code
            return x, h, w

        bb, hd = self.backbone, self.head
        # op 0 reads the caller's image tensor; its pointer is patched on every call
        # The two concats of the head (detectors.py:59, 72) cost no copy: the backbone's last block of a
        # scale writes its output straight into the first channels of the concat buffer (pixel stride =
        # the buffer's), the next backbone stage reads it from there, and the upsample kernel fills the rest.
        def last_filters(stack):
            last = list(stack)[-1]
            return (last.layer_stack[1] if isinstance(last, ResidialBlock) else last).filters
        c_s1, c_s2 = last_filters(bb.layer_stack_1), last_filters(bb.layer_stack_2)
        c_u2, c_u1 = last_filters(hd.layer_stack_1), last_filters(hd.layer_stack_2)
        cat1 = torch.empty((n, S // 8, S // 8, c_s1 + c_u1), device=device, dtype=cdt)
        cat2 = torch.empty((n, S // 16, S // 16, c_s2 + c_u2), device=device, dtype=cdt)
        s1, h1, w1 = run_stack(bb.layer_stack_1, None, S, S, first_x_code=_lib.dtype_code(x_dtype), out_last=cat1[..., :c_s1])
        s2, h2, w2 = run_stack(bb.layer_stack_2, s1, h1, w1, out_last=cat2[..., :c_s2])
        s3, h3, w3 = run_stack(bb.layer_stack_3, s2, h2, w2)
        x, _, _ = run_stack(hd.layer_stack_1, s3, h3, w3)
        l, _, _ = conv(hd.lbbox_layer_stack[0], x, h3, w3)
        l, _, _ = conv(hd.lbbox_layer_stack[1], l, h3, w3, out_dtype=hd.out_dtype, head=2)
        x = upcat(cat2, x, h3, w3, c_s2, x.shape[-1])
        x, _, _ = run_stack(hd.layer_stack_2, x, h2, w2)
        m, _, _ = conv(hd.mbbox_layer_stack[0], x, h2, w2)
        m, _, _ = conv(hd.mbbox_layer_stack[1], m, h2, w2, out_dtype=hd.out_dtype, head=1)
        s = upcat(cat1, x, h2, w2, c_s1, x.shape[-1])
        keep.extend([cat1, cat2])       # (after the per-layer buffers, whose order tools rely on)
        s, _, _ = run_stack(list(hd.sbbox_layer_stack)[:-1], s, h1, w1)
        s, _, _ = conv(hd.sbbox_layer_stack[-1], s, h1, w1, out_dtype=hd.out_dtype, head=0)

        if self.fuse_stem and len(ops) > 1 and ops[1].x == ops[0].y:
            # the first layer's output is read by the second layer alone: ymx_run_program may run
            # the pair as one launch (it checks the shapes itself)
            ops[0].C1 = 1
        part = _Plan()
        part.B, part.start = n, start
        part.ops = (_lib.ymx_op * len(ops))(*ops)
        part.n_ops = len(ops)
        part.keep = keep
        part.sites = sites
        self._bind_weights(part)
        part.x_offset = start * S * S * 3 * torch.empty((), dtype=x_dtype).element_size()
        C = self.num_classes
        padded = (s, m, l)                                  # this part's slices of the [B,S,S,cell_ld] maps
        part.head_dtype = s.dtype
        anc = _utils.default_anchors()
        part.scales = _utils._Scales(padded, _YOLO["strides"], _YOLO["xyscale"], anc,
                                     cell_ld=[t.shape[-1] for t in padded])
        cap = _utils._round4(part.scales.N)
        part.cap = cap
        f32, i32 = torch.float32, torch.int32
        part.boxes = torch.empty((n, cap, 4), device=device, dtype=f32)
        part.scores_cm = torch.empty((n, C, cap), device=device, dtype=f32)
        part.src_idx = torch.empty((n, cap), device=device, dtype=i32)
        L = _lib.lib()
        part.dec_ws_bytes = L.ymx_decode_workspace_bytes(n, part.scales.n, part.scales.S)
        part.dec_ws = torch.empty((part.dec_ws_bytes,), device=device, dtype=torch.uint8)
        part.nms_ws_bytes = L.ymx_nms_workspace_bytes(n, C, self.max_outputs_per_class)
        part.nms_ws = torch.empty((max(part.nms_ws_bytes, 16),), device=device, dtype=torch.uint8)
        part.stream = part.done = None
        return part

    def _bind_weights(self, part):
        """Point every conv op of the part at the current packed weights / scale / shift of its
        layer (YOLOConv.packed re-packs when a parameter has changed since it last did)."""
        held = []
        for idx, layer, extra in part.sites:
            wp, scale, shift, _src = layer.packed(self.dtype)
            if extra:
                if scale is not None:
                    scale = torch.cat([scale, scale.new_zeros(extra)])
                if shift is not None:
                    shift = torch.cat([shift, shift.new_zeros(extra)])
            held.extend([wp, scale, shift])
            o = part.ops[idx]
            o.w = wp.data_ptr()
            o.scale = None if scale is None else scale.data_ptr()
            o.shift = None if shift is None else shift.data_ptr()
        part.weights = held             # keeps the buffers alive for as long as the program names them

    def _build_plan(self, sizes, S, x_dtype, device):
        """One plan = the batch cut into len(sizes) consecutive sub-batches, each with its own
        program; sub-batches after the first run on side streams, concurrently."""
        B = sum(sizes)
        plan = _Plan()
        plan.B, plan.sizes = B, tuple(sizes)
        plan.padded = [None, None, None]
        T, C = self.max_outputs, self.num_classes
        f32, i32 = torch.float32, torch.int32
        plan.counts = torch.empty((B,), device=device, dtype=i32)
        plan.out_boxes = torch.empty((B, T, 4), device=device, dtype=f32)
        plan.out_scores = torch.empty((B, T), device=device, dtype=f32)
        plan.out_classes = torch.empty((B, T), device=device, dtype=f32)
        plan.out_valid = torch.empty((B,), device=device, dtype=i32)
        plan.out_keep = torch.empty((B, T), device=device, dtype=i32)
        shared = len(sizes) > 1
        plan.parts = []
        start = 0
        for i, n in enumerate(sizes):
            part = self._build_part(plan, n, start, S, x_dtype, device, _SHARED_MIN_TILES if shared else 0)
            sl = slice(start, start + n)
            part.counts = plan.counts[sl]
            part.out = tuple(t[sl] for t in (plan.out_boxes, plan.out_scores, plan.out_classes, plan.out_valid,
                                             plan.out_keep))
            if i > 0:
                side = self._side_streams.setdefault(str(device), [])
                while len(side) < i:
                    side.append(torch.cuda.Stream(device=device))
                part.stream = side[i - 1]
                part.done = torch.cuda.Event()
            plan.parts.append(part)
            start += n
        plan.start_event = torch.cuda.Event() if shared else None
        plan.heads = tuple(t[..., :3 * (5 + C)] for t in plan.padded)    # what callers see: [B,S,S,3*(5+C)]
        plan.versions = self._versions()
        # single-part views kept for callers that inspect a plan (benchmarks, tests)
        first = plan.parts[0]
        plan.scales, plan.cap, plan.head_dtype = first.scales, first.cap, first.head_dtype
        return plan

    def _candidate_splits(self, B, S):
        """Sub-batch sizes worth timing for a batch of B images of side S."""
        if isinstance(self.split, (tuple, list)):
            return [tuple(self.split)] if sum(self.split) == B else [(B,)]
        if self.split == "off" or B < 16 or B * S * S < 16 * 416 * 416:
            return [(B,)]
        cands = [(B,)]
        for num, den in ((3, 4), (5, 8), (1, 2)):
            a = B * num // den
            if 0 < a < B and (a, B - a) not in cands:
                cands.append((a, B - a))
        return cands

    def _plan(self, B, S, x_dtype, device, x=None):
        key = (B, S, x_dtype, str(device))
        plan = self._plans.get(key)
        if plan is not None:
            if plan.versions == self._versions():
                return plan
            # parameters were updated in place: same buffers and cut, freshly packed weights
            for part in plan.parts:
                self._bind_weights(part)
            plan.versions = self._versions()
            return plan
        cands = self._candidate_splits(B, S)
        if len(cands) == 1:
            plan = self._build_plan(cands[0], S, x_dtype, device)
        else:
            # Which cut of the batch is fastest depends on how the tile counts of the layers
            # fall on the 256 CUs, so it is measured, once per input shape.  Results do not
            # depend on the cut (every image is computed by the same arithmetic).
            if x is None:
                x = torch.rand((B, S, S, 3), device=device).to(x_dtype)
            whole = self._build_plan(cands[0], S, x_dtype, device)
            best, best_ms = whole, self._time_plan(whole, x, S) * 0.98     # a cut must win by 2 %
            for _attempt in range(4):
                for sizes in cands[1:]:
                    cand = self._build_plan(sizes, S, x_dtype, device)
                    ms = self._time_plan(cand, x, S)
                    if ms < best_ms:
                        best, best_ms = cand, ms
                    del cand
                if best is not whole:
                    break
                # No cut won.  A side stream that the runtime placed on the hardware queue of
                # the current stream cannot overlap with it (measured: about one stream in
                # four, and stable for the life of the stream), so try with fresh ones.
                self._side_streams.pop(str(device), None)
            plan = best
        self._plans[key] = plan
        return plan

    def _time_plan(self, plan, x, S, warm=2, steps=4):
        for _ in range(warm):
            self._run(plan, x, S, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            self._run(plan, x, S, True)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps

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

    def _run_part(self, part, x_ptr, image_size, post):
        part.ops[0].x = x_ptr + part.x_offset
        _lib.launch("ymx_run_program", part.boxes, part.ops, part.n_ops)
        if post:
            self._postprocess_part(part, image_size)

    def _run(self, plan, x, image_size, post):
        """All sub-batches of the plan: the first on the current stream, the others on side
        streams that start after the work already queued on the current stream and are
        joined back into it before this returns."""
        parts = plan.parts
        x_ptr = x.data_ptr()
        if len(parts) > 1:
            main = torch.cuda.current_stream()
            plan.start_event.record(main)
            for part in parts[1:]:
                part.stream.wait_event(plan.start_event)
                with torch.cuda.stream(part.stream):
                    self._run_part(part, x_ptr, image_size, post)
                    part.done.record(part.stream)
        self._run_part(parts[0], x_ptr, image_size, post)
        for part in parts[1:]:
            torch.cuda.current_stream().wait_event(part.done)

    def prepare(self, batch, size, x_dtype=torch.float32, device=None):
        """Plans the program for [batch, size, size, 3] inputs now instead of inside the first
        forward: allocates the activation buffers (one per layer output, about 100 MB per
        416x416 image in bf16) and, with split="auto", times the candidate cuts of the batch
        (about 0.3 s at batch 64; every candidate but the whole-batch plan and the best so
        far is released as soon as it has been timed).  Returns the chosen sub-batch sizes
        and the bytes of device memory the plan keeps."""
        if size % 32:
            raise ValueError("image side must be a multiple of 32, got %d" % size)
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("yolo_amd: no CPU fallback; prepare() needs a GPU device")
        plan = self._plan(int(batch), int(size), x_dtype, device)
        kept = sum(t.numel() * t.element_size() for part in plan.parts for t in part.keep if torch.is_tensor(t))
        return {"sub_batches": [part.B for part in plan.parts], "bytes": int(kept)}

    def forward_heads(self, x):
        """images [B,S,S,3] -> (sbbox, mbbox, lbbox) raw conv maps (plan-owned buffers)."""
        x = self._check_input(x)
        plan = self._plan(x.shape[0], x.shape[1], x.dtype, x.device, x)
        self._run(plan, x, x.shape[1], False)
        return plan.heads

    def _postprocess_part(self, part, image_size):
        self._decode_part(part, image_size)
        self._nms_part(part)

    def _decode_part(self, part, image_size):
        sc = part.scales
        _lib.launch("ymx_decode_filter_strided", part.boxes, sc.ptrs, sc.cell_ld, sc.n, sc.S, part.B, self.num_classes,
                    _lib.dtype_code(part.head_dtype), sc.anchors, sc.strides, sc.xyscale, 1, float(self.score_threshold),
                    float(image_size), _lib.ptr(part.boxes), _lib.ptr(part.scores_cm), _lib.ptr(part.src_idx),
                    _lib.ptr(part.counts), part.cap, _lib.ptr(part.dec_ws), part.dec_ws_bytes)

    def _nms_part(self, part):
        out_boxes, out_scores, out_classes, out_valid, out_keep = part.out
        _lib.launch("ymx_nms_combined", part.boxes, _lib.ptr(part.boxes), _lib.ptr(part.scores_cm), _lib.ptr(part.counts),
                    _lib.ptr(part.src_idx), part.B, part.cap, self.num_classes, int(self.max_outputs_per_class),
                    int(self.max_outputs), float(self.iou_threshold), float(self.score_threshold),
                    _lib.ptr(out_boxes), _lib.ptr(out_scores), _lib.ptr(out_classes), _lib.ptr(out_valid),
                    _lib.ptr(out_keep), _lib.ptr(part.nms_ws), part.nms_ws_bytes)

    def check_status(self):
        """Synchronises and raises if any decode+filter launch of this detector's plans left
        its status word set (a workgroup gave up waiting for the segments before it, which a
        healthy device never does).  The steady-state path never synchronises, so this is
        for callers that want the check at a point of their choosing."""
        words = [part.dec_ws[:4].view(torch.int32) for plan in self._plans.values() for part in plan.parts]
        if words and bool(torch.cat(words).any().item()):
            raise RuntimeError("yolo_amd: a decode+filter launch timed out waiting for a predecessor "
                               "segment (device fault?); its detections are invalid")

    def postprocess(self, plan, image_size):
        """decode + filter + combined NMS of the plan's head maps, on the current stream."""
        for part in plan.parts:
            self._postprocess_part(part, image_size)
        return _utils.Detections(plan.out_boxes, plan.out_scores, plan.out_classes, plan.out_valid, plan.out_keep)

    def forward(self, x, image_size=None):
        """images -> (boxes [B,T,4], scores [B,T], classes [B,T], valid [B]) with .keep_idx.

        The returned tensors are the plan's output buffers: they are overwritten by
        the next call with the same input shape."""
        x = self._check_input(x)
        plan = self._plan(x.shape[0], x.shape[1], x.dtype, x.device, x)
        self._run(plan, x, x.shape[1] if image_size is None else image_size, True)
        return _utils.Detections(plan.out_boxes, plan.out_scores, plan.out_classes, plan.out_valid, plan.out_keep)
