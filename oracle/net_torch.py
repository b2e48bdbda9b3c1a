"""torch-CPU fp32 restatement of the reference's conv blocks, backbone and head.

Test infrastructure only (see ``oracle/__init__.py``); parity unpinned: the
convolution, batch-norm and bilinear-resize arithmetic of the reference is
TensorFlow's, which cannot run here; this file restates their published
definitions.

Follows
  /root/reference/layers.py:17-22    Mish
  /root/reference/layers.py:33-39    Upsample (tf.image.resize bilinear, TF2
                                     half-pixel centres, no antialias)
  /root/reference/layers.py:41-90    YOLOConv
  /root/reference/layers.py:92-106   ResidialBlock
  /root/reference/backbones.py:6-60  Darknet53
  /root/reference/detectors.py:4-77  YoloV3

Tensors are NHWC and kernels HWIO at this module's interface, as in the
reference; torch's NCHW/OIHW is used only inside ``yolo_conv``.

A network is described by a flat list of *conv specs* in execution order, the
same order the product package uses for its parameters, so one list of
parameter dicts drives both.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # Keras BatchNormalization default epsilon (layers.py:4)
LEAKY_ALPHA = 0.1      # layers.py:76


def mish(x):
    """x * tanh(softplus(x)); layers.py:22."""
    return x * torch.tanh(F.softplus(x))


def activation(x, act):
    if act in (None, False, "", "none"):
        return x
    if act == "mish":
        return mish(x)
    if act == "leaky":
        return F.leaky_relu(x, LEAKY_ALPHA)
    if callable(act):
        return act(x)
    raise ValueError("Activation should either be a function or one of \"leaky\" or \"mish\".")


BN_MOMENTUM = 0.99      # Keras BatchNormalization default, used unchanged at layers.py:4


def batch_norm_train(y_nchw, p, update=None):
    """BatchNormalization.call with training=True and a trainable layer (layers.py:11-15 hands
    over to Keras): normalise with the batch's own mean and biased variance over N, H, W, then
    gamma / beta.  `update`, if a dict, receives the moving statistics Keras would keep
    afterwards (momentum 0.99; the fused kernel Keras takes for NHWC inputs feeds the moving
    variance with the unbiased estimate, n / (n - 1) times the biased one)."""
    mean = y_nchw.mean(dim=(0, 2, 3))
    var = y_nchw.var(dim=(0, 2, 3), unbiased=False)
    if update is not None:
        n = y_nchw.numel() // y_nchw.shape[1]
        with torch.no_grad():
            update["mean"] = p["mean"] * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
            update["var"] = p["var"] * BN_MOMENTUM + var.detach() * (n / max(n - 1, 1)) * (1 - BN_MOMENTUM)
    xhat = (y_nchw - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
    return xhat * p["gamma"].view(1, -1, 1, 1) + p["beta"].view(1, -1, 1, 1)


def yolo_conv(x, p, downsample=False, act="mish", bn_training=False, bn_update=None, bn_round=None):
    """One YOLOConv block on an NHWC tensor; layers.py:46-81.

    p: dict with 'kernel' [kh,kw,Cin,Cout] and either 'bias' [Cout] or the
    batch-norm tensors 'gamma','beta','mean','var' (use_bias = not
    batchNormalization, layers.py:61).  downsample: pad 1 row on top and 1
    column on the left only, then stride-2 VALID (layers.py:46-54); otherwise
    stride-1 SAME (layers.py:55-57).
    """
    w = p["kernel"].permute(3, 2, 0, 1).contiguous()          # HWIO -> OIHW
    xn = x.permute(0, 3, 1, 2)
    kh, kw = p["kernel"].shape[0], p["kernel"].shape[1]
    if downsample:
        xn = F.pad(xn, (1, 0, 1, 0))                           # left, right, top, bottom
        y = F.conv2d(xn, w, p.get("bias"), stride=2, padding=0)
    else:
        # TF SAME at stride 1 with an odd kernel pads (k-1)/2 on every side.
        y = F.conv2d(xn, w, p.get("bias"), stride=1, padding=((kh - 1) // 2, (kw - 1) // 2))
    if "gamma" in p and bn_round is not None:
        y = bn_round(y)              # what a reduced-precision path stores between the convolution and the normalisation
    if "gamma" in p and bn_training:
        y = batch_norm_train(y, p, bn_update)
    elif "gamma" in p:
        scale = p["gamma"] / torch.sqrt(p["var"] + BN_EPS)
        shift = p["beta"] - p["mean"] * scale
        y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    y = activation(y, act)
    return y.permute(0, 2, 3, 1).contiguous()


def upsample2x(x):
    """Bilinear x2 on NHWC, half-pixel centres, edge clamp; layers.py:37-39."""
    y = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    return y.permute(0, 2, 3, 1).contiguous()


# ---------------------------------------------------------------------------
# Network description: flat conv-spec lists in execution order.
# Each spec: (kh, cin, cout, downsample, act)
# ---------------------------------------------------------------------------

def darknet53_specs():
    """52 conv specs of Darknet53 in call order; backbones.py:10-42."""
    specs = [(3, 3, 32, False, "mish"), (3, 32, 64, True, "mish")]

    def res(n, f1, f2):
        for _ in range(n):
            specs.append((1, f2, f1, False, "mish"))
            specs.append((3, f1, f2, False, "mish"))

    res(1, 32, 64)
    specs.append((3, 64, 128, True, "mish"))
    res(2, 64, 128)
    specs.append((3, 128, 256, True, "mish"))
    res(8, 128, 256)
    specs.append((3, 256, 512, True, "mish"))
    res(8, 256, 512)
    specs.append((3, 512, 1024, True, "mish"))
    res(4, 512, 1024)
    return specs


def yolov3_specs(num_classes):
    """21 conv specs of YoloV3 in *attribute* order (layer_stack_1, lbbox,
    layer_stack_2, mbbox, sbbox); detectors.py:10-44."""
    nout = 3 * (num_classes + 5)
    s = []
    cin = 1024
    for k, c in ((1, 512), (3, 1024), (1, 512), (3, 1024), (1, 512)):
        s.append((k, cin, c, False, "mish")); cin = c
    s.append((3, 512, 1024, False, "mish"))
    s.append((1, 1024, nout, False, None))
    cin = 512 + 512
    for k, c in ((1, 512), (3, 1024), (1, 512), (3, 1024), (1, 512)):
        s.append((k, cin, c, False, "mish")); cin = c
    s.append((3, 512, 1024, False, "mish"))
    s.append((1, 1024, nout, False, None))
    cin = 256 + 512
    for k, c in ((1, 512), (3, 1024), (1, 512), (3, 1024), (1, 512), (3, 1024)):
        s.append((k, cin, c, False, "mish")); cin = c
    s.append((1, 1024, nout, False, None))
    return s


def init_params(specs, seed=0, init="reference", batch_norm=False, bn_random=False):
    """Random parameters for a spec list, drawn on the CPU from one generator.

    init='reference': kernel ~ N(0, 0.01^2), bias 0 (layers.py:62-63).
    init='variance' : kernel ~ N(0, 2/K), K = kh*kw*Cin, bias ~ N(0, 0.1^2), so
                      that activations keep a usable scale through 70+ layers.
    batch_norm: replace the bias by BN tensors (identity statistics unless
    bn_random) for every spec that has an activation.
    """
    g = torch.Generator().manual_seed(seed)
    params = []
    for (k, cin, cout, _down, act) in specs:
        if init == "reference":
            std = 0.01
        else:
            std = math.sqrt(2.0 / (k * k * cin))
        p = {"kernel": torch.randn(k, k, cin, cout, generator=g) * std}
        if batch_norm and act is not None:
            if bn_random:
                p["gamma"] = 1.0 + 0.2 * torch.randn(cout, generator=g)
                p["beta"] = 0.1 * torch.randn(cout, generator=g)
                p["mean"] = 0.1 * torch.randn(cout, generator=g)
                p["var"] = 0.5 + torch.rand(cout, generator=g)
            else:
                p["gamma"] = torch.ones(cout); p["beta"] = torch.zeros(cout)
                p["mean"] = torch.zeros(cout); p["var"] = torch.ones(cout)
        else:
            p["bias"] = torch.zeros(cout) if init == "reference" else 0.1 * torch.randn(cout, generator=g)
        params.append(p)
    return params


def darknet53_forward(params, x, act="mish", round_fn=None, hook=None):
    """x [B,S,S,3] -> (scale_1 [B,S/8,S/8,256], scale_2 [..,512], scale_3 [..,1024]).

    backbones.py:44-60.  ResidialBlock adds the shortcut after the second
    activation (layers.py:100-106).  round_fn, if given, is applied to every
    block output (used to emulate bf16 activation storage).
    """
    specs = darknet53_specs()
    rf = round_fn if round_fn is not None else (lambda t: t)
    it = iter(zip(specs, params))

    def conv(x, kind="main"):
        (k, cin, cout, down, a), p = next(it)
        if hook is not None:
            hook(p, x, down, kind)
        return yolo_conv(x, p, down, act if a is not None else None)

    def res(x):
        y = rf(conv(x))
        y = conv(y, "res2")
        return rf(x + y)

    x = rf(conv(x)); x = rf(conv(x))
    x = res(x)
    x = rf(conv(x))
    for _ in range(2): x = res(x)
    x = rf(conv(x))
    for _ in range(8): x = res(x)
    s1 = x
    x = rf(conv(x))
    for _ in range(8): x = res(x)
    s2 = x
    x = rf(conv(x))
    for _ in range(4): x = res(x)
    s3 = x
    return s1, s2, s3


def yolov3_forward(params, s1, s2, s3, num_classes, act="mish", round_fn=None, hook=None):
    """(scale_1, scale_2, scale_3) -> (sbbox, mbbox, lbbox) raw conv maps.

    detectors.py:46-77: five convs at /32, lbbox branch, upsample + concat
    [scale_2, up], five convs, mbbox branch, upsample + concat [scale_1, up],
    seven convs ending in sbbox.  The three output convs have no activation.
    """
    specs = yolov3_specs(num_classes)
    rf = round_fn if round_fn is not None else (lambda t: t)
    it = iter(zip(specs, params))

    def conv(x, final=False):
        (k, cin, cout, down, a), p = next(it)
        if hook is not None:
            hook(p, x, down, "out" if final else "main")
        y = yolo_conv(x, p, down, act if a is not None else None)
        return y if final else rf(y)

    x = s3
    for _ in range(5): x = conv(x)
    l = conv(x); l = conv(l, final=True)
    x = rf(upsample2x(x))
    x = torch.cat([s2, x], dim=-1)
    for _ in range(5): x = conv(x)
    m = conv(x); m = conv(m, final=True)
    x = rf(upsample2x(x))
    s = torch.cat([s1, x], dim=-1)
    for _ in range(6): s = conv(s)
    s = conv(s, final=True)
    return s, m, l


def detector_forward(bparams, hparams, x, num_classes, act="mish", round_fn=None, hook=None):
    s1, s2, s3 = darknet53_forward(bparams, x, act, round_fn, hook)
    return yolov3_forward(hparams, s1, s2, s3, num_classes, act, round_fn, hook)


def calibrated_params(num_classes, seed=0, side=64, out_std=2.0, conf_bias=0.0):
    """Random parameters whose activations stay O(1) through all 73 layers.

    The reference's own init (N(0, 0.01^2), layers.py:62) collapses every activation
    towards 0, and a fan-in init explodes through the 23 shortcut adds, so neither
    exercises the kernels.  Here every kernel is drawn N(0, 1) and then rescaled, layer by
    layer in execution order on a seeded calibration image, so that its pre-activation has
    unit standard deviation (0.5 on the second conv of a residual block, out_std on the
    three linear output convs).  Deterministic for a given (num_classes, seed, side).
    Returns (backbone_params, head_params).
    """
    g = torch.Generator().manual_seed(seed)
    bspecs, hspecs = darknet53_specs(), yolov3_specs(num_classes)

    def draw(specs):
        out = []
        for (k, cin, cout, _d, _a) in specs:
            out.append({"kernel": torch.randn(k, k, cin, cout, generator=g) / math.sqrt(k * k * cin),
                        "bias": torch.zeros(cout)})
        return out

    bp, hp = draw(bspecs), draw(hspecs)
    x = torch.rand(2, side, side, 3, generator=g)

    def hook(p, xin, down, kind):
        target = {"main": 1.0, "res2": 0.5, "out": out_std}[kind]
        z = yolo_conv(xin, {"kernel": p["kernel"], "bias": None}, down, None)
        p["kernel"] = p["kernel"] * (target / float(z.std()))
        if kind == "out":
            b = torch.zeros(p["kernel"].shape[3])
            b.view(3, -1)[:, 4] = conf_bias
            p["bias"] = b
        else:
            p["bias"] = 0.05 * torch.randn(p["kernel"].shape[3], generator=g)

    detector_forward(bp, hp, x, num_classes, hook=hook)
    return bp, hp


def bf16_round(t):
    """Round-trip through bfloat16 (round to nearest even), staying fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


def round_params_bf16(params):
    """Kernel weights rounded to bf16 (what the bf16 HIP path stores); the
    epilogue vectors (bias / BN) stay fp32."""
    out = []
    for p in params:
        q = dict(p)
        q["kernel"] = bf16_round(p["kernel"])
        out.append(q)
    return out


# ---------------------------------------------------------------------------
# CSPDarknet53 (backbones.py:62-264), REPAIRED.  The reference's class cannot be
# constructed: it passes three positionals to ResidialBlock(filter_num1, filter_num2,
# activation) (layers.py:93), and its second conv has 54 filters where every consumer
# expects 64 (backbones.py:68).  The definition below is the evident intent:
# ResidialBlock(in, f1, f2) -> ResidialBlock(f1, f2), 54 -> 64.  There is no oracle for it
# even in principle; this restatement is what the product is checked against.
# Stack items: ("conv", k, cout, down) or ("res", f1, f2); stacks in attribute order.
# ---------------------------------------------------------------------------

def cspdarknet53_stacks():
    def csp(n, f_in, f_mid, f_res1):
        main = [("conv", 1, f_mid, False)] + [("res", f_res1, f_mid)] * n + [("conv", 1, f_mid, False)]
        return main
    return [
        ("layer_stack_1", [("conv", 3, 32, False), ("conv", 3, 64, True)]),
        ("layer_stack_2", [("conv", 1, 64, False)]),
        ("layer_stack_2_skip", csp(1, 64, 64, 32)),
        ("layer_stack_3", [("conv", 1, 64, False), ("conv", 3, 128, True)]),
        ("layer_stack_4", csp(2, 128, 64, 64)),
        ("layer_stack_4_skip", [("conv", 1, 64, False)]),
        ("layer_stack_5", [("conv", 1, 128, False), ("conv", 3, 256, True)]),
        ("layer_stack_6", csp(8, 256, 128, 128)),
        ("layer_stack_6_skip", [("conv", 1, 128, False)]),
        ("layer_stack_7", [("conv", 1, 256, False)]),
        ("layer_stack_8", [("conv", 3, 512, True)]),
        ("layer_stack_9", csp(8, 512, 256, 256)),
        ("layer_stack_9_skip", [("conv", 1, 256, False)]),
        ("layer_stack_10", [("conv", 1, 512, False)]),
        ("layer_stack_11", [("conv", 3, 1024, True)]),
        ("layer_stack_12", csp(4, 1024, 512, 512)),
        ("layer_stack_12_skip", [("conv", 1, 512, False)]),
        ("layer_stack_13", [("conv", 1, 1024, False), ("conv", 1, 512, False), ("conv", 3, 1024, False), ("conv", 1, 512, False)]),
        ("layer_stack_14", [("conv", 1, 512, False), ("conv", 3, 1024, False), ("conv", 1, 512, False)]),
    ]


# input channels of every stack (the data flow of CSPDarknet53.call, backbones.py:175-264)
_CSP_INPUT = {"layer_stack_1": 3, "layer_stack_2": 64, "layer_stack_2_skip": 64, "layer_stack_3": 128,
              "layer_stack_4": 128, "layer_stack_4_skip": 128, "layer_stack_5": 128, "layer_stack_6": 256,
              "layer_stack_6_skip": 256, "layer_stack_7": 256, "layer_stack_8": 256, "layer_stack_9": 512,
              "layer_stack_9_skip": 512, "layer_stack_10": 512, "layer_stack_11": 512, "layer_stack_12": 1024,
              "layer_stack_12_skip": 1024, "layer_stack_13": 1024, "layer_stack_14": 2048}


def cspdarknet53_specs():
    """Flat conv specs (k, cin, cout, down, act) in attribute order, residual blocks expanded
    (1x1 then 3x3), the order of the product's parameters."""
    specs = []
    for name, items in cspdarknet53_stacks():
        cin = _CSP_INPUT[name]
        for it in items:
            if it[0] == "conv":
                specs.append((it[1], cin, it[2], it[3], "mish"))
                cin = it[2]
            else:
                specs.append((1, cin, it[1], False, "mish"))
                specs.append((3, it[1], it[2], False, "mish"))
                cin = it[2]
    return specs


def cspdarknet53_forward(params, x, round_fn=None):
    """x [B,S,S,3] -> (route_1 [B,S/8,S/8,256], route_2 [B,S/16,S/16,512], x [B,S/32,S/32,512]);
    backbones.py:175-264 on the repaired definition.  SPP: stride-1 SAME max-pools of 13, 9,
    5 concatenated with their input (backbones.py:254-259; padding never wins a max)."""
    rf = round_fn if round_fn is not None else (lambda t: t)
    it = iter(params)
    stacks = dict(cspdarknet53_stacks())

    def run(name, t):
        for item in stacks[name]:
            if item[0] == "conv":
                t = rf(yolo_conv(t, next(it), item[3], "mish"))
            else:
                y = rf(yolo_conv(t, next(it), False, "mish"))
                y = yolo_conv(y, next(it), False, "mish")
                t = rf(t + y)
        return t

    # parameters are consumed in attribute order, so run the stacks in that order and keep
    # the results the data flow needs
    x = run("layer_stack_1", x)
    route = run("layer_stack_2", x)
    x = run("layer_stack_2_skip", x)
    x = torch.cat([x, route], dim=-1)
    x = run("layer_stack_3", x)
    route = x
    x = run("layer_stack_4", x)
    route = run("layer_stack_4_skip", route)
    x = torch.cat([x, route], dim=-1)
    x = run("layer_stack_5", x)
    route = x
    x = run("layer_stack_6", x)
    route = run("layer_stack_6_skip", route)
    x = torch.cat([x, route], dim=-1)
    x = run("layer_stack_7", x)
    route_1 = x
    x = run("layer_stack_8", x)
    route = x
    x = run("layer_stack_9", x)
    route = run("layer_stack_9_skip", route)
    x = torch.cat([x, route], dim=-1)
    x = run("layer_stack_10", x)
    route_2 = x
    x = run("layer_stack_11", x)
    route = x
    x = run("layer_stack_12", x)
    route = run("layer_stack_12_skip", route)
    x = torch.cat([x, route], dim=-1)
    x = run("layer_stack_13", x)
    xn = x.permute(0, 3, 1, 2)
    x = torch.cat([F.max_pool2d(xn, k, 1, k // 2) for k in (13, 9, 5)] + [xn], dim=1).permute(0, 2, 3, 1).contiguous()
    x = run("layer_stack_14", x)
    return route_1, route_2, x
