"""GPU parity of the backward pass of the convolution blocks (yolo_amd.training) against
torch's automatic differentiation of the CPU oracle (fp32 arithmetic on the same
bf16-rounded inputs and weights).  bf16 tolerance: 1e-2 of each gradient tensor's scale for
one block, looser bounds (stated per test) where several bf16 blocks are chained."""
import numpy as np
import pytest
import torch

from oracle import net_torch as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scaled_err(got, want):
    return float((got.double() - want.double()).abs().max() / want.double().abs().max().clamp_min(1e-30))


CASES = [   # B, H, W, Cin, Cout, k, downsample, act, shortcut
    (2, 13, 11, 64, 128, 3, False, "mish", True),
    (3, 16, 16, 32, 64, 3, True, "mish", False),          # stride 2: 16 -> 8
    (2, 10, 12, 128, 64, 1, False, "leaky", False),
    (1, 26, 26, 256, 255, 1, False, None, False),         # a head's output convolution: 255 channels, linear
    (2, 9, 7, 16, 24, 3, False, "mish", True),            # partial tiles everywhere
    (1, 40, 40, 64, 192, 3, False, "leaky", True),        # several pixel slices
    (2, 15, 15, 64, 32, 3, True, "mish", False),          # stride 2 on an odd size: 15 -> 7
    (2, 52, 52, 64, 128, 3, False, "mish", False),        # all-taps weight gradient: many pixel slices, 256-row ring
    (1, 104, 104, 32, 72, 3, False, "leaky", False),      # 512-row ring, partial channel tiles
    (1, 13, 13, 256, 512, 3, False, "leaky", False),      # a slice shorter than the ring's run-in
    (1, 208, 208, 8, 16, 3, False, "leaky", False),       # 1024-row ring (one workgroup per CU)
    (1, 500, 500, 8, 16, 3, False, "leaky", False),       # too wide for the ring: per-tap kernel
    (1, 52, 52, 64, 128, 3, True, "leaky", False),        # stride 2 at a layer's real width
    (2, 26, 30, 40, 72, 3, True, "mish", False),          # the same with channel counts off the tile sizes
    (2, 12, 8, 16, 255, 3, True, None, False),            # and an output channel count that is padded (255 -> 256)
    (2, 21, 19, 24, 40, 3, True, "mish", False),           # all-taps stride-2 weight gradient: odd sizes, partial channel tiles
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_conv_block_gradients(case):
    from yolo_amd import training as T
    from yolo_amd.layers import YOLOConv
    B, H, W, cin, cout, k, down, act, use_sc = CASES[case]
    g = torch.Generator().manual_seed(500 + case)
    x = N.bf16_round(torch.randn(B, H, W, cin, generator=g))
    p = {"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5),
         "bias": N.bf16_round(0.1 * torch.randn(cout, generator=g))}
    # oracle: fp32 autograd
    xo = x.clone().requires_grad_(True)
    po = {n: v.clone().requires_grad_(True) for n, v in p.items()}
    yo = N.yolo_conv(xo, po, down, act)
    sc = N.bf16_round(torch.randn(*yo.shape, generator=g)) if use_sc else None
    sco = None
    if use_sc:
        sco = sc.clone().requires_grad_(True)
        yo = yo + sco
    G = N.bf16_round(torch.randn(*yo.shape, generator=g))
    (yo * G).sum().backward()
    # GPU
    layer = YOLOConv((k, k, cout), downsample=down, activation=act, dtype=torch.bfloat16)
    layer.load_reference_params(p)
    layer.to(DEV)
    xg = x.to(DEV).bfloat16().requires_grad_(True)
    scg = sc.to(DEV).bfloat16().requires_grad_(True) if use_sc else None
    with T.backprop():
        yg = layer(xg, residual=scg)
    assert yg.requires_grad and yg.dtype == torch.bfloat16
    assert _scaled_err(yg.detach().float().cpu(), yo.detach()) < 1e-2
    (yg.float() * G.to(DEV)).sum().backward()
    assert layer.kernel.grad.shape == p["kernel"].shape and layer.kernel.grad.dtype == torch.float32
    errs = {"kernel": _scaled_err(layer.kernel.grad.cpu(), po["kernel"].grad), "bias": _scaled_err(layer.bias.grad.cpu(), po["bias"].grad),
            "x": _scaled_err(xg.grad.float().cpu(), xo.grad)}
    if use_sc:
        errs["shortcut"] = _scaled_err(scg.grad.float().cpu(), sco.grad)
    print("conv block gradient errors (of each tensor's scale):", errs)
    for name, e in errs.items():
        assert e < 1e-2, (name, e)
    # the other weight-gradient kernels (1: register-staged, taken for tensors beyond the 2 GB that
    # LDS-DMA addresses; 2: per-tap LDS-DMA, taken where the all-taps kernel does not apply) give
    # the same gradient
    from yolo_amd import _lib
    g_auto = layer.kernel.grad.clone()
    # (backward() runs on autograd's own thread and libymx options are per thread, so the variant
    # goes through training.options, which the backward step applies where it launches; the
    # trace shows which kernel each run really took)
    for variant, label in ((1, "wgrad_reg"), (2, "wgrad_dma")):
        layer.kernel.grad = None
        T.trace = []
        try:
            T.options["wgrad_variant"] = variant
            with T.backprop():
                y2 = layer(xg.detach(), residual=None if scg is None else scg.detach())
            (y2.float() * G.to(DEV)).sum().backward()
            assert T.trace == [label], T.trace
        finally:
            T.options["wgrad_variant"] = 0
            T.trace = None
        assert _scaled_err(layer.kernel.grad.cpu(), po["kernel"].grad) < 1e-2
        assert _scaled_err(layer.kernel.grad.cpu(), g_auto.cpu()) < 1e-5          # same products, another summation order
    # the weight gradient on a second stream beside the input gradient: the same bits
    layer.kernel.grad = None
    try:
        T.overlap_wgrad = True
        xs = xg.detach().requires_grad_(True)
        with T.backprop():
            y3 = layer(xs, residual=None if scg is None else scg.detach())
        (y3.float() * G.to(DEV)).sum().backward()
    finally:
        T.overlap_wgrad = False
    torch.cuda.synchronize()
    assert torch.equal(layer.kernel.grad, g_auto) and torch.equal(xs.grad, xg.grad)
    # outside the context nothing is recorded and the fused inference kernels run
    y_inf = layer(xg.detach(), residual=None if scg is None else scg.detach())
    assert not y_inf.requires_grad


FULL_LAYERS = [   # B, H, Cin, Cout, k, stride: the detector's own layer shapes at 416 x 416
    (1, 52, 512, 1024, 3, 1),      # the head's widest layer: 169 32-pixel steps, 4 pixel slices
    (2, 26, 512, 1024, 3, 1),
    (3, 13, 512, 1024, 3, 1),      # image borders every 13 pixels, three images in one ring
    (1, 104, 64, 128, 3, 1),       # 512-row ring
    (1, 208, 32, 64, 3, 1),        # 1024-row ring
    (1, 416, 8, 32, 3, 1),         # first layer (channels padded 3 -> 8): one live wave per workgroup
    (1, 416, 32, 64, 3, 2),        # stride 2, 32 input channels: the all-taps stride-2 kernel
    (2, 37, 24, 40, 3, 2),         # the same on an odd size with channel counts off its tiles
    (1, 52, 256, 512, 3, 2),
    (2, 52, 256, 128, 1, 1),
]


@pytest.mark.parametrize("case", range(len(FULL_LAYERS)))
def test_weight_gradient_at_full_layer_sizes(case):
    """ymx_conv_wgrad through the C ABI on whole layers of the detector, all three kernels,
    against the fp32 CPU derivative of the oracle's convolution (same bf16 inputs: the
    products are exact in fp32, only the order of the sum differs)."""
    from yolo_amd import _lib
    B, H, cin, cout, k, stride = FULL_LAYERS[case]
    g = torch.Generator().manual_seed(900 + case)
    oh = H if stride == 1 else (H + 1 - k) // 2 + 1
    x = N.bf16_round(torch.randn(B, H, H, cin, generator=g))
    dz = N.bf16_round(torch.randn(B, oh, oh, cout, generator=g))
    if cin == 8:
        x[..., 3:] = 0                                   # the padding channels of the first layer
    xc, dzc = x.permute(0, 3, 1, 2).contiguous(), dz.permute(0, 3, 1, 2).contiguous()
    if stride == 2:                                      # one zero row / column on the top and left, then VALID (layers.py:46-54)
        xc = torch.nn.functional.pad(xc, (1, 0, 1, 0))
        want = torch.nn.grad.conv2d_weight(xc, (cout, cin, k, k), dzc, stride=2, padding=0)
    else:
        want = torch.nn.grad.conv2d_weight(xc, (cout, cin, k, k), dzc, stride=1, padding=(k - 1) // 2)
    want = want.permute(2, 3, 1, 0).contiguous()         # OIHW -> HWIO
    L = _lib.lib()
    xg, dzg = x.to(DEV).bfloat16(), dz.to(DEV).bfloat16()
    got = {}
    for variant in (0, 1, 2):
        dw = torch.full((k, k, cin, cout), float("nan"), device=DEV)
        try:
            _lib.check(L.ymx_set_option(b"wgrad_variant", variant))
            nb = L.ymx_conv_wgrad_workspace_bytes(B, H, H, cin, cout, k, stride)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
            _lib.launch("ymx_conv_wgrad", xg, _lib.ptr(xg), 0, _lib.ptr(dzg), 0, _lib.ptr(dw), B, H, H, cin, cout, k, stride, _lib.ptr(ws), nb)
        finally:
            _lib.check(L.ymx_set_option(b"wgrad_variant", 0))
        got[variant] = dw.cpu()
        e = _scaled_err(got[variant], want)
        print("wgrad variant %d, %s: scaled error %.2e" % (variant, FULL_LAYERS[case], e))
        assert e < 1e-4, (variant, e)
    # launching again gives the same bits (ordered partial sums, no float atomics)
    dw2 = torch.empty((k, k, cin, cout), device=DEV)
    nb = L.ymx_conv_wgrad_workspace_bytes(B, H, H, cin, cout, k, stride)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    _lib.launch("ymx_conv_wgrad", xg, _lib.ptr(xg), 0, _lib.ptr(dzg), 0, _lib.ptr(dw2), B, H, H, cin, cout, k, stride, _lib.ptr(ws), nb)
    assert torch.equal(dw2.cpu(), got[0])


def test_residial_block_gradients():
    """x + conv3x3(conv1x1(x)) (layers.py:92-106): the gradient of x is the sum of the path through
    the two convolutions and of the shortcut; on the GPU the second term is added inside the
    first block's input-gradient convolution instead of by autograd."""
    from yolo_amd import training as T
    from yolo_amd.layers import ResidialBlock
    B, H, W, c1, c2 = 2, 12, 10, 32, 64
    g = torch.Generator().manual_seed(77)
    x = N.bf16_round(torch.randn(B, H, W, c2, generator=g))
    p1 = {"kernel": N.bf16_round(torch.randn(1, 1, c2, c1, generator=g) * (2.0 / c2) ** 0.5), "bias": N.bf16_round(0.1 * torch.randn(c1, generator=g))}
    p2 = {"kernel": N.bf16_round(torch.randn(3, 3, c1, c2, generator=g) * (2.0 / (9 * c1)) ** 0.5), "bias": N.bf16_round(0.1 * torch.randn(c2, generator=g))}
    G = N.bf16_round(torch.randn(B, H, W, c2, generator=g))
    # a layer in front, so that x is not a leaf (the usual place of a block in the network) and
    # the shortcut's gradient has to pass through two blocks' backward functions
    xo = x.clone().requires_grad_(True)
    po = [{n: v.clone().requires_grad_(True) for n, v in p.items()} for p in (p1, p2)]
    yo = xo * 1.0
    yo = yo + N.yolo_conv(N.yolo_conv(yo, po[0], False, "mish"), po[1], False, "mish")
    (yo * G).sum().backward()
    blk = ResidialBlock(c1, c2, activation="mish", dtype=torch.bfloat16)
    blk.layer_stack[0].load_reference_params(p1)
    blk.layer_stack[1].load_reference_params(p2)
    blk.to(DEV)
    xg = x.to(DEV).bfloat16().requires_grad_(True)
    with T.backprop():
        yg = blk(xg * 1.0)
    assert _scaled_err(yg.detach().float().cpu(), yo.detach()) < 1e-2
    (yg.float() * G.to(DEV)).sum().backward()
    errs = {"x": _scaled_err(xg.grad.float().cpu(), xo.grad)}
    for i, layer in enumerate(blk.layer_stack):
        errs["kernel%d" % i] = _scaled_err(layer.kernel.grad.cpu(), po[i]["kernel"].grad)
        errs["bias%d" % i] = _scaled_err(layer.bias.grad.cpu(), po[i]["bias"].grad)
    print("ResidialBlock gradient errors:", errs)
    for name, e in errs.items():
        assert e < 1e-2, (name, e)
    # a block whose input needs no gradient (the shortcut is a constant) still trains its kernels
    for layer in blk.layer_stack:
        layer.kernel.grad = None
    with T.backprop():
        y2 = blk(x.to(DEV).bfloat16())
    (y2.float() * G.to(DEV)).sum().backward()
    assert _scaled_err(blk.layer_stack[0].kernel.grad.cpu(), po[0]["kernel"].grad) < 1e-2


def test_upsample_concat_gradient():
    from yolo_amd import training as T
    from yolo_amd.layers import upsample2x_concat
    g = torch.Generator().manual_seed(9)
    for (B, H, W, c1, c2) in ((2, 5, 7, 16, 24), (1, 13, 13, 0, 8), (1, 1, 1, 8, 8), (2, 26, 26, 256, 128)):
        x = N.bf16_round(torch.randn(B, H, W, c2, generator=g))
        skip = N.bf16_round(torch.randn(B, 2 * H, 2 * W, c1, generator=g)) if c1 else None
        G = N.bf16_round(torch.randn(B, 2 * H, 2 * W, c1 + c2, generator=g))
        xo = x.clone().requires_grad_(True)
        up = N.upsample2x(xo)
        so = None
        if c1:
            so = skip.clone().requires_grad_(True)
            up = torch.cat([so, up], dim=-1)
        (up * G).sum().backward()
        for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
            xg = x.to(DEV).to(dt).requires_grad_(True)
            sg = skip.to(DEV).to(dt).requires_grad_(True) if c1 else None
            with T.backprop():
                y = upsample2x_concat(sg, xg)
            (y.float() * G.to(DEV)).sum().backward()
            assert _scaled_err(xg.grad.float().cpu(), xo.grad) < tol
            if c1:
                assert _scaled_err(sg.grad.float().cpu(), so.grad) < tol


def test_small_network_gradients_and_descent():
    """A stride-2 conv, a ResidialBlock, an upsample + concat and a linear output conv chained:
    every parameter gradient against the oracle's autograd (3e-2 of its scale: three to four
    bf16 blocks deep), and a few steps of gradient descent on the GPU lower the loss."""
    from yolo_amd import training as T
    from yolo_amd.layers import YOLOConv, ResidialBlock, upsample2x_concat
    g = torch.Generator().manual_seed(77)
    specs = [(3, 16, 32, True, "mish"), (1, 32, 16, False, "mish"), (3, 16, 32, False, "mish"), (1, 48, 24, False, None)]
    params = []
    for k, cin, cout, down, act in specs:
        params.append({"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5),
                       "bias": N.bf16_round(0.05 * torch.randn(cout, generator=g))})
    x = N.bf16_round(torch.randn(2, 12, 12, 16, generator=g))
    target = torch.randn(2, 12, 12, 24, generator=g)

    def oracle():
        ps = [{n: v.clone().requires_grad_(True) for n, v in p.items()} for p in params]
        a = N.yolo_conv(x, ps[0], True, "mish")                                  # 12 -> 6
        b = a + N.yolo_conv(N.yolo_conv(a, ps[1], False, "mish"), ps[2], False, "mish")
        c = torch.cat([x, N.upsample2x(b)], dim=-1)                             # 16 + 32 channels at 12 x 12
        out = N.yolo_conv(c, ps[3], False, None)
        loss = ((out - target) ** 2).mean()
        loss.backward()
        return float(loss.detach()), ps

    want_loss, ps = oracle()
    first = YOLOConv((3, 3, 32), downsample=True, dtype=torch.bfloat16)
    block = ResidialBlock(16, 32, dtype=torch.bfloat16)
    last = YOLOConv((1, 1, 24), activation=None, dtype=torch.bfloat16)
    layers = [first, block.layer_stack[0], block.layer_stack[1], last]
    for layer, p in zip(layers, params):
        layer.load_reference_params(p)
        layer.to(DEV)
    xg = x.to(DEV).bfloat16()
    tg = target.to(DEV)

    def step():
        with T.backprop():
            a = first(xg)
            b = block(a)
            out = last(upsample2x_concat(xg, b))
        loss = ((out.float() - tg) ** 2).mean()
        for layer in layers:
            layer.kernel.grad = None
            layer.bias.grad = None
        loss.backward()
        return float(loss.detach())

    got_loss = step()
    assert abs(got_loss - want_loss) < 2e-2 * abs(want_loss)
    for layer, p in zip(layers, ps):
        for name in ("kernel", "bias"):
            e = _scaled_err(getattr(layer, name).grad.cpu(), p[name].grad)
            assert e < 3e-2, (name, tuple(p[name].shape), e)
    losses = [got_loss]
    for _ in range(5):
        with torch.no_grad():
            for layer in layers:
                layer.kernel -= 0.05 * layer.kernel.grad
                layer.bias -= 0.05 * layer.bias.grad
        losses.append(step())
    assert losses[-1] < 0.9 * losses[0], losses


def test_detector_training_step_vs_oracle():
    """Darknet53 + YoloV3 + the YOLO loss at the three scales plus the L2 kernel regulariser of
    every convolution, 64 x 64 inputs: the gradient of every one of the 146 parameter tensors
    after back-propagation through 75 bf16 blocks against the oracle (same bf16 storage roundings in the forward pass, fp32 autograd
    backward).  Per tensor: cosine similarity and error relative to the tensor's scale.
    Measured: cosine >= 0.9997 on every tensor, scaled error median 1.3e-2, worst 3.6e-2 (what
    75 chained bf16 blocks leave; the single-block bound of 1e-2 is in test_conv_block_gradients)."""
    import yolo_amd as Y
    from yolo_amd import Detector, dataset as DS, losses as L, training as T
    from oracle import loss_torch as LT
    from oracle import postproc_np as P
    C, B, S = 3, 2, 64
    cfg = Y.config
    bp, hp = N.calibrated_params(C, seed=1, side=S)          # activations O(1) in every layer, head logits of std 2
    x = torch.rand(B, S, S, 3, generator=torch.Generator().manual_seed(3))
    rng = np.random.default_rng(5)
    boxes = np.zeros((B, 4, 5), np.float32)
    for b in range(B):
        for j in range(4):
            x0, y0 = rng.uniform(0, 40, 2)
            w, h = rng.uniform(6, 22, 2)
            boxes[b, j] = [np.floor(x0), np.floor(y0), np.floor(min(x0 + w, 63)), np.floor(min(y0 + h, 63)), rng.integers(0, C)]
    labels = DS.preprocess_true_boxes(torch.from_numpy(boxes).to(DEV), None, S, C)
    targets = DS.pack_targets(*labels)
    # oracle
    rbp, rhp = N.round_params_bf16(bp), N.round_params_bf16(hp)
    leaves = []
    for plist in (rbp, rhp):
        for p in plist:
            for name in ("kernel", "bias"):
                p[name] = p[name].clone().requires_grad_(True)
                leaves.append(p[name])
    heads = N.detector_forward(rbp, rhp, x, C, round_fn=N.bf16_round)
    total = 0
    for i, h in enumerate(heads):
        Si = h.shape[1]
        D = 5 + C
        conv5 = N.bf16_round(h).double().reshape(B, Si, Si, 3, D)          # the training path stores the head maps in bf16
        tgt = targets[i].cpu().double()
        split = Si * Si * 3 * D
        pred = LT.decode_train(conv5, i, Si, P.STRIDES, P.XYSCALE, P.ANCHORS)
        g, c, p = LT.yolo_loss(pred, conv5, tgt[:, :split].reshape(B, Si, Si, 3, D), tgt[:, split:].reshape(B, 128, 4),
                               Si * P.STRIDES[i], 0.5)
        total = total + g + c + p
    kernels_o = [p["kernel"] for plist in (rbp, rhp) for p in plist]
    penalty = LT.l2_penalty(kernels_o)                     # kernel_regularizer of every convolution (layers.py:61)
    (total + penalty).backward()
    # GPU
    det = Detector(num_classes=C, dtype=torch.bfloat16)
    det.load_reference_params(bp, hp)
    det.to(DEV)
    fns = L.get_yolo_loss_at_scales([0, 1, 2], C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"])
    with T.backprop():
        s, m, l = det.head(*det.backbone(x.to(DEV)))
    loss = fns[2](targets[0], s) + fns[1](targets[1], m) + fns[0](targets[2], l)
    loss.backward()
    reg = T.L2Regulariser(det)
    reg.add_gradients_()
    assert abs(float(loss.detach()) - float(total.detach())) < 3e-2 * abs(float(total.detach()))
    assert abs(float(reg.penalty()) - float(penalty.detach())) < 1e-3 * float(penalty.detach())    # fp32 master weights here, their bf16 roundings there
    convs = det.backbone.conv_layers() + det.head.conv_layers()
    got = []
    for conv in convs:
        got += [conv.kernel.grad, conv.bias.grad]
    assert len(got) == len(leaves) == 146
    cos, err = [], []
    for gq, w in zip(got, leaves):
        a, b = gq.double().cpu().reshape(-1), w.grad.double().reshape(-1)
        cos.append(float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300)))
        err.append(float((a - b).abs().max() / b.abs().max().clamp_min(1e-300)))
    print("detector gradients vs oracle: cosine min %.4f median %.4f; scaled error max %.3e median %.3e" %
          (min(cos), float(np.median(cos)), max(err), float(np.median(err))))
    assert min(cos) > 0.999
    assert float(np.median(err)) < 2e-2 and max(err) < 6e-2


Z_CASES = [   # B, H, W, Cin, Cout, k, stride, act, shortcut, the kernel family the shape is routed to
    (64, 26, 26, 512, 1024, 3, 1, 2, False, "halo_mixed"),   # resident-slab kernel, 256- and 192-pixel tiles in one launch
    (48, 26, 26, 512, 1024, 3, 1, 2, False, "halo"),      # 256-pixel tiles
    (16, 52, 52, 128, 256, 3, 1, 2, True, "halo"),        # the same with the shortcut
    (64, 13, 13, 512, 1024, 3, 1, 1, False, "halo192"),   # 192-pixel tiles
    (8, 52, 52, 1024, 512, 1, 1, 2, False, "dma4"),       # phase-staggered kernel
    (32, 52, 52, 256, 512, 3, 2, 1, False, "dma4"),       # stride 2
    (4, 208, 208, 64, 32, 1, 1, 2, False, "dma2"),        # narrow LDS-DMA kernel
    (4, 208, 208, 32, 64, 3, 1, 2, True, "slab2"),        # narrow input-slab kernel, shortcut
    (4, 104, 104, 128, 64, 1, 1, 1, False, "dma2"),
    (1, 13, 13, 256, 512, 3, 1, 2, True, "igemm"),        # too few tiles: register-staged kernel, two launches
    (2, 20, 24, 40, 72, 3, 1, 2, False, "igemm"),         # channel counts off the tile sizes
    (2, 64, 64, 3, 32, 3, 1, 2, False, "direct"),         # first layer (fp32 images): two launches
]


@pytest.mark.parametrize("case", range(len(Z_CASES)))
def test_training_forward_two_outputs(case):
    """ymx_conv2d_fused_z through the C ABI: y carries the bits of the inference convolution
    (ymx_conv2d_fused_ex with the activation and the shortcut), z those of the same
    convolution without activation, for every kernel family and for the shapes that run as
    two launches."""
    from yolo_amd import _lib
    B, H, W, cin, cout, k, stride, act, use_sc, family = Z_CASES[case]
    L = _lib.lib()
    g = torch.Generator().manual_seed(900 + case)
    first = cin % 8 != 0
    x = torch.randn(B, H, W, cin, generator=g).to(DEV)
    if not first:
        x = x.bfloat16()
    wt = (torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(DEV)
    shift = (0.1 * torch.randn(cout, generator=g)).to(DEV)
    code = _lib.BF16
    wp = torch.empty(L.ymx_packed_weight_bytes(k, cin, cout, code), dtype=torch.uint8, device=DEV)
    _lib.launch("ymx_pack_weights", wp, _lib.ptr(wt), _lib.ptr(wp), k, cin, cout, code)
    oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
    sc = torch.randn(B, oh, ow, cout, generator=g).to(DEV).bfloat16() if use_sc else None
    xd = _lib.dtype_code(x.dtype)

    def conv(a, res):
        y = torch.empty(B, oh, ow, cout, device=DEV, dtype=torch.bfloat16)
        _lib.launch("ymx_conv2d_fused_ex", x, _lib.ptr(x), _lib.ptr(wp), None, _lib.ptr(shift), _lib.ptr(res), _lib.ptr(y), B, H, W, cin, cout,
                    k, stride, a, code, xd, code, 0, 0, 0)
        return y

    want_y, want_z = conv(act, sc), conv(_lib.ACT_NONE, None)
    y = torch.full((B, oh, ow, cout), float("nan"), device=DEV, dtype=torch.bfloat16)
    z = torch.full_like(y, float("nan"))
    _lib.launch("ymx_conv2d_fused_z", x, _lib.ptr(x), _lib.ptr(wp), None, _lib.ptr(shift), _lib.ptr(sc), _lib.ptr(y), _lib.ptr(z),
                B, H, W, cin, cout, k, stride, act, xd)
    torch.cuda.synchronize()
    assert torch.equal(z.view(torch.int16), want_z.view(torch.int16))
    op = _lib.ymx_op()
    op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, B, H, W, cin, cout, k, stride
    op.dtype = op.out_dtype = code
    op.x_dtype = xd
    op.x, op.w, op.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
    import ctypes
    label = L.ymx_conv_kernel_label(ctypes.byref(op)).decode()
    print("kernel family:", label)
    assert label.startswith(family) or (family == "direct" and label == "first_rows")
    if label in ("igemm", "direct", "first_rows"):
        # two launches: y = act(z rounded to bf16) + shortcut, one rounding more than the fused epilogue
        ref = want_y.float()
        assert float((y.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    else:
        assert torch.equal(y.view(torch.int16), want_y.view(torch.int16))


def test_weight_packs_of_a_training_step_in_one_launch():
    """ymx_pack_weights_pair = ymx_pack_weights + ymx_pack_weights_dgrad, byte for byte."""
    from yolo_amd import _lib
    L = _lib.lib()
    code = _lib.BF16
    for k, cin, cout in ((3, 64, 128), (1, 256, 255), (3, 40, 72), (1, 1024, 512), (3, 8, 16)):
        wt = torch.randn(k, k, cin, cout, generator=torch.Generator().manual_seed(k + cin)).to(DEV)
        coutp = -(-cout // 8) * 8
        nf, nb = L.ymx_packed_weight_bytes(k, cin, cout, code), L.ymx_packed_weight_bytes(k, coutp, cin, code)
        f0, b0, f1, b1 = (torch.full((n,), 0x5a, dtype=torch.uint8, device=DEV) for n in (nf, nb, nf, nb))
        _lib.launch("ymx_pack_weights", f0, _lib.ptr(wt), _lib.ptr(f0), k, cin, cout, code)
        _lib.launch("ymx_pack_weights_dgrad", b0, _lib.ptr(wt), _lib.ptr(b0), k, cin, cout, coutp, code)
        _lib.launch("ymx_pack_weights_pair", f1, _lib.ptr(wt), _lib.ptr(f1), _lib.ptr(b1), k, cin, cout, coutp, 1, code)
        assert torch.equal(f0, f1) and torch.equal(b0, b1), (k, cin, cout)


def test_gradient_buckets_receive_the_block_gradients_in_place():
    """With yolo_amd.dist.GradientBuckets the convolution blocks write weight and bias gradients
    straight into the buckets (padded layers through one strided copy) and report them ready:
    the same numbers as the gradients autograd would have been handed, every bucket started
    during backward() in bucket order, and the L2 regulariser adds 2 * 0.0005 * kernel."""
    from yolo_amd import dist as D, training as T
    from yolo_amd.layers import YOLOConv, ResidialBlock
    torch.manual_seed(5)
    specs = [YOLOConv((3, 3, 32), downsample=True, dtype=torch.bfloat16, in_channels=3),
             ResidialBlock(16, 32, dtype=torch.bfloat16),
             YOLOConv((1, 1, 255), activation=None, dtype=torch.bfloat16, in_channels=32)]
    x = torch.rand(2, 24, 24, 3).to(DEV)
    block = specs[1]
    block.layer_stack[0].build(32)
    block.layer_stack[1].build(16)
    net = torch.nn.ModuleList(specs).to(DEV)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn_like(p) * (0.2 if p.dim() == 4 else 0.05))
    params = list(net.parameters())

    def run():
        with T.backprop():
            out = specs[2](block(specs[0](x)))
        out.float().square().mean().backward()

    run()
    want = [p.grad.clone() for p in params]
    gb = D.GradientBuckets(params, bucket_bytes=16 << 10)
    assert len(gb.buckets) >= 3
    for step in range(2):
        gb.begin()
        run()
        torch.cuda.synchronize()
        assert gb.launch_order == list(range(len(gb.buckets)))
        gb.wait()
        for p, w in zip(params, want):
            assert p.grad.data_ptr() == p._grad_bucket[1].data_ptr()
            assert torch.equal(p.grad, w), tuple(p.shape)
    reg = T.L2Regulariser(net)
    assert len(reg.kernels) == 4
    reg.add_gradients_()
    for m in net.modules():
        if isinstance(m, YOLOConv):
            i = [id(p) for p in params].index(id(m.kernel))
            assert torch.allclose(m.kernel.grad, want[i] + 0.001 * m.kernel.detach(), rtol=1e-6, atol=1e-9)
            j = [id(p) for p in params].index(id(m.bias))
            assert torch.equal(m.bias.grad, want[j])
    # back to plain gradients: the blocks hand them to autograd again, the buckets are left alone
    snapshot = [bk.clone() for bk in gb.buckets]
    for p in params:
        p.grad = None
    run()
    assert all(torch.equal(p.grad, w) for p, w in zip(params, want))
    assert all(torch.equal(a_, b_) for a_, b_ in zip(gb.buckets, snapshot))
    gb.close()
    assert not any(hasattr(p, "_grad_bucket") for p in params)
    pen = float(reg.penalty())
    assert abs(pen - 0.0005 * sum(float(k.detach().double().square().sum()) for k in reg.kernels)) < 1e-6 * pen


S2_CASES = [   # B, H, W, Cin, Cout -> kernel family of the 2x2 depth-to-space convolution
    (4, 208, 208, 64, 128, "dma4"),       # 4 * 64 = 256 output channels: phase-staggered kernel
    (4, 416, 416, 32, 64, "dma2"),        # 128 output channels: narrow LDS-DMA kernel
    (16, 52, 52, 256, 512, "dma4"),
    (16, 208, 208, 64, 128, "dma4"),      # 676 tiles: the 256-pixel form of that kernel (the two above take 192-pixel tiles)
    (2, 30, 26, 16, 40, "igemm"),         # register-staged kernel; dz stored with 40 channels
    (1, 15, 13, 8, 255, "igemm"),         # odd sizes (last row / column without gradient), 255 -> 256 dz channels
]


@pytest.mark.parametrize("case", range(len(S2_CASES)))
def test_stride2_input_gradient_depth_to_space(case):
    """ymx_pack_weights_dgrad_s2 + ymx_conv2d_dgrad_s2 through the C ABI against the fp32 CPU
    derivative of the oracle's stride-2 convolution with respect to its input (same bf16
    inputs; 1e-2 of the gradient's scale: bf16 output rounding)."""
    import ctypes
    from yolo_amd import _lib
    B, H, W, cin, cout, family = S2_CASES[case]
    L = _lib.lib()
    g = torch.Generator().manual_seed(1200 + case)
    wt = N.bf16_round(torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5)
    xo = torch.zeros(B, H, W, cin, requires_grad=True)
    yo = N.yolo_conv(xo, {"kernel": wt, "bias": torch.zeros(cout)}, True, None)
    oh, ow = yo.shape[1], yo.shape[2]
    dz = N.bf16_round(torch.randn(B, oh, ow, cout, generator=g))
    (yo * dz).sum().backward()
    coutp = -(-cout // 8) * 8
    dzp = torch.zeros(B, oh, ow, coutp, dtype=torch.bfloat16, device=DEV)
    dzp[..., :cout] = dz.to(DEV)
    code = _lib.BF16
    wg = wt.to(DEV)
    wp = torch.empty(L.ymx_packed_weight_bytes(2, coutp, 4 * cin, code), dtype=torch.uint8, device=DEV)
    _lib.launch("ymx_pack_weights_dgrad_s2", wp, _lib.ptr(wg), _lib.ptr(wp), cin, cout, coutp, code)
    # the pair pack writes the same bytes
    wf = torch.empty(L.ymx_packed_weight_bytes(3, cin, cout, code), dtype=torch.uint8, device=DEV)
    wp2 = torch.empty_like(wp)
    _lib.launch("ymx_pack_weights_pair", wp2, _lib.ptr(wg), _lib.ptr(wf), _lib.ptr(wp2), 3, cin, cout, coutp, 2, code)
    assert torch.equal(wp, wp2)
    dx = torch.zeros(B, H, W, cin, dtype=torch.bfloat16, device=DEV)
    dx[:, :2 * oh, :2 * ow] = float("nan")                  # every element of the covered region must be written
    _lib.launch("ymx_conv2d_dgrad_s2", dzp, _lib.ptr(dzp), _lib.ptr(wp), _lib.ptr(dx), B, oh, ow, coutp, cin, H, W)
    torch.cuda.synchronize()
    op = _lib.ymx_op()
    op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, B, oh, ow, coutp, 4 * cin, 1, 1
    op.dtype = op.out_dtype = op.x_dtype = code
    op.x, op.w, op.y = dzp.data_ptr(), wp.data_ptr(), dx.data_ptr()
    label = L.ymx_conv_kernel_label(ctypes.byref(op)).decode()      # (asked for a 1x1 of the same size: the same routing)
    assert label == family or (family == "dma4" and label in ("dma4_192", "dma4_mixed")), label
    e = _scaled_err(dx.float().cpu(), xo.grad)
    print("stride-2 input gradient, %s: error %.2e of the gradient's scale" % (family, e))
    assert e < 1e-2


FIRST_CASES = [   # B, H, W, Cin, Cout, fp32 image
    (2, 416, 416, 3, 32, True),        # the detector's first layer
    (3, 37, 45, 3, 32, False),         # width off the 32-pixel steps, bf16 input
    (1, 64, 100, 1, 64, True),         # one input channel, four output blocks
    (2, 20, 33, 2, 24, True),          # output channels off the block size
]


@pytest.mark.parametrize("case", range(len(FIRST_CASES)))
def test_first_layer_weight_gradient(case):
    """ymx_conv_wgrad_first through the C ABI against the fp32 CPU derivative of the oracle's
    convolution with respect to its kernel (same inputs rounded to bf16, as the kernel rounds the
    image): 1e-4 of the gradient's scale, bit-identical between two launches."""
    from yolo_amd import _lib
    B, H, W, cin, cout, f32 = FIRST_CASES[case]
    L = _lib.lib()
    g = torch.Generator().manual_seed(1500 + case)
    x = N.bf16_round(torch.rand(B, H, W, cin, generator=g))
    dz = N.bf16_round(torch.randn(B, H, W, cout, generator=g))
    wo = torch.zeros(3, 3, cin, cout, requires_grad=True)
    yo = N.yolo_conv(x, {"kernel": wo, "bias": torch.zeros(cout)}, False, None)
    (yo * dz).sum().backward()
    xg = x.to(DEV) if f32 else x.to(DEV).bfloat16()
    dzg = dz.to(DEV).bfloat16()
    nb = L.ymx_conv_wgrad_first_workspace_bytes(B, H, W, cin, cout)
    assert nb > 0
    outs = []
    for _ in range(2):
        ws = torch.full((nb,), 0xff, dtype=torch.uint8, device=DEV)
        dw = torch.full((3, 3, cin, cout), float("nan"), device=DEV)
        _lib.launch("ymx_conv_wgrad_first", xg, _lib.ptr(xg), _lib.dtype_code(xg.dtype), _lib.ptr(dzg), 0, _lib.ptr(dw), B, H, W, cin, cout,
                    _lib.ptr(ws), nb)
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    e = _scaled_err(outs[0].cpu(), wo.grad)
    print("first-layer weight gradient: error %.2e of the gradient's scale" % e)
    assert e < 1e-4
    assert L.ymx_conv_wgrad_first_workspace_bytes(B, H, W, 8, cout) == 0          # other shapes: the general kernels
