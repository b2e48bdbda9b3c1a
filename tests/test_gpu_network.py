"""GPU parity of the whole detector against the CPU oracle, plus size-independent
properties at the benchmark's full sizes (determinism, batch invariance).

The oracle's conv stack is torch-CPU fp32.  For bf16 the oracle is run with the same
storage roundings the HIP path has (weights and every stored activation rounded to
bf16, fp32 accumulation), so the comparison isolates kernel arithmetic; the distance to
the un-rounded fp32 oracle is reported as well.
"""
import numpy as np
import pytest
import torch

from oracle import cref
from oracle import net_torch as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def _params(C, init, seed=0, side=64):
    """'reference' = the reference's own N(0, 0.01^2) init (degenerate: activations
    collapse to 0); 'variance' = parameters calibrated on a seeded image of the given
    side so that every layer's activations stay O(1)."""
    if init == "variance":
        return N.calibrated_params(C, seed=seed, side=side)
    return (N.init_params(N.darknet53_specs(), seed=seed, init=init),
            N.init_params(N.yolov3_specs(C), seed=seed + 1, init=init))


def _image(B, S, seed=1):
    return torch.rand(B, S, S, 3, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("init", ["variance", "reference"])
def test_modules_fp32_vs_oracle(init):
    """Darknet53 -> YoloV3 through the module API, fp32 MFMA path, 64x64 input."""
    from yolo_amd.backbones import Darknet53
    from yolo_amd.detectors import YoloV3
    C = 3
    bp, hp = _params(C, init)
    x = _image(2, 64)
    want_s = N.darknet53_forward(bp, x)
    want_h = N.yolov3_forward(hp, *want_s, C)
    bb, hd = Darknet53(), YoloV3(C)
    bb.load_reference_params(bp)
    hd.load_reference_params(hp)
    bb.to(DEV); hd.to(DEV)
    got_s = bb(x.to(DEV))
    got_h = hd(*got_s, training=False)
    for g, w in zip(got_s, want_s):
        assert g.shape == w.shape and g.dtype == torch.float32
        assert rel_err(g.cpu().numpy(), w.numpy()) < 1e-3
    for g, w in zip(got_h, want_h):
        assert g.shape == w.shape and g.dtype == torch.float32
        assert rel_err(g.cpu().numpy(), w.numpy()) < 1e-3


def test_bf16_every_op_vs_rounding_oracle_on_gpu_inputs():
    """Kernel error isolated from drift: each of the 75 ops of the bf16 network (73 convs,
    2 upsample+concat) is checked on the GPU's OWN input to that op -- the oracle computes
    the op in fp32 on exactly the bf16 values the kernel read (bf16-rounded weights, fp32
    accumulation, one final rounding) -- so differences cannot accumulate across layers.
    Every op must hold 1e-2 relative to its output scale (bf16 outputs) / 1e-3 (the three
    fp32 head maps)."""
    from yolo_amd import Detector
    from yolo_amd import detectors as D
    from yolo_amd.layers import YOLOConv
    C = 80
    bp, hp = _params(C, "variance", side=96)
    x = _image(2, 96)
    det = Detector(num_classes=C, dtype=torch.bfloat16)
    det.load_reference_params(bp, hp)
    det.to(DEV)
    rbp, rhp = N.round_params_bf16(bp), N.round_params_bf16(hp)
    pmap = {}
    for conv, p, (k, cin, cout, down, act) in zip(det.backbone.conv_layers() + det.head.conv_layers(), rbp + rhp,
                                                  N.darknet53_specs() + N.yolov3_specs(C)):
        pmap[conv] = (p, down, act)
    worst = {"conv_bf16": 0.0, "conv_f32": 0.0, "upcat": 0.0}
    count = {"conv": 0, "upcat": 0}

    def hook(mod, args, kwargs, out):
        p, down, act = pmap[mod]
        xin = args[0].float().cpu()
        if mod.in_channels == 3:
            xin = args[0].float().cpu()                   # the fp32 image, rounded to bf16 inside the kernel
            xin = N.bf16_round(xin)
        want = N.yolo_conv(xin, p, down, act)
        res = kwargs.get("residual")
        if res is not None:
            want = want + res.float().cpu()
        e = rel_err(out.float().cpu().numpy(), want.numpy())
        key = "conv_f32" if out.dtype == torch.float32 else "conv_bf16"
        worst[key] = max(worst[key], e)
        count["conv"] += 1
        assert e < (1e-3 if out.dtype == torch.float32 else 1e-2), (mod.name, tuple(out.shape), e)

    handles = [m.register_forward_hook(hook, with_kwargs=True) for m in det.modules() if isinstance(m, YOLOConv)]
    real_upcat = D.upsample2x_concat

    def checked_upcat(skip, xx):
        y = real_upcat(skip, xx)
        want = torch.cat([skip.float().cpu(), N.upsample2x(xx.float().cpu())], dim=-1)
        e = rel_err(y.float().cpu().numpy(), want.numpy())
        worst["upcat"] = max(worst["upcat"], e)
        count["upcat"] += 1
        assert e < 1e-2, e
        return y

    D.upsample2x_concat = checked_upcat
    try:
        s = det.backbone(x.to(DEV))
        heads = det.head(*s)
    finally:
        D.upsample2x_concat = real_upcat
        for h in handles:
            h.remove()
    assert count == {"conv": 73, "upcat": 2}
    print("bf16 per-op error vs rounding oracle on the GPU's own inputs:", worst)
    # the planned program computes the same bits as the module-by-module walk just checked
    for a, b in zip(heads, det.forward_heads(x.to(DEV))):
        assert torch.equal(a, b)


def _decoded_errors(heads_gpu, want_heads, C, B):
    """Decode GPU head maps on the GPU and oracle head maps on the CPU; per tensor the
    differences the north-star tolerance speaks of.  'scale' errors are |d| / max|want|,
    'elem' errors |d| / |want| element by element."""
    import yolo_amd as Y
    from yolo_amd import utils as U
    from oracle import postproc_np as P
    cfg = Y.config
    dec = U.decode_heads([h.reshape(B, h.shape[1], h.shape[2], 3, 5 + C) for h in heads_gpu], C, cfg["strides"],
                         cfg["xyscale"], U.default_anchors())
    n = dec["N"]
    g_box = dec["boxes"][:, :n].cpu().numpy().astype(np.float64)
    g_sc = dec["scores_cm"][:, :, :n].transpose(1, 2).cpu().numpy().astype(np.float64)
    w_box, w_sc = P.decode_all([h.numpy().reshape(B, h.shape[1], h.shape[2], 3, 5 + C) for h in want_heads], C)
    w_box = w_box.astype(np.float64); w_sc = w_sc.astype(np.float64)
    out = {}
    for name, g, w in (("xy", g_box[..., :2], w_box[..., :2]), ("wh", g_box[..., 2:], w_box[..., 2:]),
                       ("score", g_sc, w_sc)):
        d = np.abs(g - w)
        out[name] = {"scale_max": float(d.max() / np.abs(w).max()),
                     "scale_rms": float(np.sqrt((d ** 2).mean()) / np.sqrt((w ** 2).mean())),
                     "elem_max": float((d / np.maximum(np.abs(w), 1e-30)).max()),
                     "elem_rms": float(np.sqrt(((d / np.maximum(np.abs(w), 1e-30)) ** 2).mean())),
                     "elem_median": float(np.median(d / np.maximum(np.abs(w), 1e-30)))}
    return out


@pytest.mark.parametrize("side,batch", [(96, 2), (160, 2), (416, 1)])
def test_bf16_decoded_boxes_and_scores_vs_fp32_oracle(side, batch):
    """The north-star quantity: decoded boxes and scores of the bf16 detector against the
    fp32 CPU oracle on identical inputs and weights.

    (1) Under the reference's own random init (N(0, 0.01^2) kernels, zero bias -- the
        configuration the 1e-2 tolerance is stated for) every decoded value holds 1e-2
        element by element, with a wide margin.
    (2) Under the activation-calibrated init, where the 75 layers carry O(1) signals and
        the head logits have a standard deviation of 2, storing every activation in bf16
        (relative rounding error up to 2^-9 per stored value, 75 times in a row) leaves the
        logits with an RMS error of about 1 % of their scale however the kernels are
        written; test_bf16_every_op_vs_rounding_oracle_on_gpu_inputs holds each kernel to
        1e-2 on its own inputs.  Decoding maps a logit error dt to dt * s'(t) * stride for
        box centres and sigmoid products (bounded by dt / 4) and to a RELATIVE error dt on
        exp(t_wh).  Measured at 96^2 / 160^2 / 416^2 (identical within 10 %): logit RMS
        error 1.25-1.32e-2 of the logit RMS, worst logit 0.12-0.16 absolute;
          centres  worst element 1.3e-3 ... 5.0e-3 of the image scale      -> 1e-2 asserted
          scores   RMS 8.4e-3 ... 9.3e-3 of the score RMS                  -> 1e-2 asserted
                   worst element 2.4e-2 ... 3.5e-2 (absolute, scores <= 1) -> bounded by dt_max / 2
          sizes    relative error = |dt|: median 1.3e-2 ... 1.5e-2, RMS 2.2e-2 ... 2.4e-2,
                   worst 9e-2 ... 1.2e-1                                   -> 2e-2 / 3e-2 / expm1(dt_max)
        The size and worst-score figures exceed 1e-2 because of bf16 activation storage, not
        because of any kernel: the fp32 path holds 1e-3 on the same weights.
    """
    from yolo_amd import Detector
    C = 80
    x = _image(batch, side)
    # (1) reference init
    bp, hp = _params(C, "reference")
    det = Detector(num_classes=C, dtype=torch.bfloat16)
    det.load_reference_params(bp, hp)
    det.to(DEV)
    heads = [h.clone() for h in det.forward_heads(x.to(DEV))]
    want = N.detector_forward(bp, hp, x, C)
    err = _decoded_errors(heads, want, C, batch)
    print("bf16 decoded vs fp32 oracle, reference init, %d^2:" % side, err)
    for name in ("xy", "wh", "score"):
        assert err[name]["elem_max"] < 1e-2, (name, err[name])
    # (2) calibrated init
    bp, hp = _params(C, "variance", side=side)
    det = Detector(num_classes=C, dtype=torch.bfloat16)
    det.load_reference_params(bp, hp)
    det.to(DEV)
    heads = [h.clone() for h in det.forward_heads(x.to(DEV))]
    want = N.detector_forward(bp, hp, x, C)
    dt_max = max(float((g.cpu() - w).abs().max()) for g, w in zip(heads, want))
    dt_rms = max(float((g.cpu() - w).pow(2).mean().sqrt() / w.pow(2).mean().sqrt()) for g, w in zip(heads, want))
    err = _decoded_errors(heads, want, C, batch)
    print("bf16 decoded vs fp32 oracle, calibrated init, %d^2: logit max |dt| %.3e, rms dt / rms t %.3e" %
          (side, dt_max, dt_rms), err)
    assert dt_rms < 2e-2
    assert err["xy"]["scale_max"] < 1e-2 and err["xy"]["elem_rms"] < 1e-2
    assert err["score"]["scale_rms"] < 1e-2 and err["score"]["scale_max"] <= 0.25 * 2 * dt_max + 1e-6
    assert err["wh"]["elem_median"] < 2e-2
    assert err["wh"]["elem_max"] <= float(np.expm1(dt_max)) * 1.001 + 1e-6
    assert err["wh"]["elem_rms"] < 3e-2


def test_detector_bf16_vs_rounding_oracle():
    """bf16 network (planned program) end to end vs the oracle with identical storage
    roundings.  Two bf16 pipelines that differ only in fp32 summation order still drift
    apart: a sum that lands on the other side of a bf16 rounding boundary flips one stored
    activation by an ulp (0.4 %), and 75 layers of that leave about 1 % RMS at the heads.
    The kernels themselves are held to 1e-2 per op on identical inputs by
    test_bf16_every_op_vs_rounding_oracle_on_gpu_inputs; this test bounds the accumulated
    drift (RMS and worst element) and prints it."""
    from yolo_amd import Detector
    C = 80
    bp, hp = _params(C, "variance", side=96)
    x = _image(2, 96)
    det = Detector(num_classes=C, dtype=torch.bfloat16)
    det.load_reference_params(bp, hp)
    det.to(DEV)
    heads = [h.clone() for h in det.forward_heads(x.to(DEV))]
    assert all(h.dtype == torch.float32 for h in heads)
    # the image itself is consumed in fp32 by the first layer; everything stored is bf16
    want = N.detector_forward(N.round_params_bf16(bp), N.round_params_bf16(hp), x, C, round_fn=N.bf16_round)
    pure = N.detector_forward(bp, hp, x, C)
    for g, w, p in zip(heads, want, pure):
        e = rel_err(g.cpu().numpy(), w.numpy())
        e32 = rel_err(g.cpu().numpy(), p.numpy())
        rms = float((g.cpu() - w).pow(2).mean().sqrt() / w.pow(2).mean().sqrt())
        print("bf16 head %s: vs rounding oracle max %.3e rms %.3e, vs fp32 oracle max %.3e" % (tuple(g.shape), e, rms, e32))
        assert rms < 2e-2          # measured 1.1e-2 ... 1.4e-2
        assert e < 2e-2            # measured 1.1e-2 ... 1.5e-2
        assert e32 < 5e-2
    # module-by-module path gives the same bits as the planned program
    s = det.backbone(x.to(DEV))
    h2 = det.head(*s)
    for a, b in zip(heads, h2):
        assert torch.equal(a, b)


def test_plan_follows_parameter_updates():
    """A planned program holds pointers to packed weights; in-place parameter updates after
    the first forward (load_state_dict, copy_) must reach the next forward, and moving the
    module drops the plans."""
    from yolo_amd import Detector
    C = 3
    x = _image(2, 64).to(DEV)
    det = Detector(num_classes=C, dtype=torch.bfloat16, seed=11).to(DEV)
    det.calibrate_(x)
    h0 = [h.clone() for h in det.forward_heads(x)]
    other = Detector(num_classes=C, dtype=torch.bfloat16, seed=12).to(DEV)
    other.calibrate_(x)
    want = [h.clone() for h in other.forward_heads(x)]
    assert not torch.equal(h0[0], want[0])
    plan = det._plan(2, 64, x.dtype, x.device)
    det.load_state_dict(other.state_dict())
    got = det.forward_heads(x)
    assert det._plan(2, 64, x.dtype, x.device) is plan          # same buffers, re-bound weights
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    # one in-place edit of one bias
    with torch.no_grad():
        det.head.sbbox_layer_stack[-1].bias.add_(1.0)
    got = det.forward_heads(x)
    assert torch.allclose(got[0], want[0] + 1.0, atol=1e-5) and torch.equal(got[1], want[1])
    # the module-by-module path sees the same parameters
    s = det.backbone(x)
    for a, b in zip(det.head(*s), got):
        assert torch.equal(a, b)


def test_detector_fp32_end_to_end_detections():
    """Config 1 geometry (416x416, batch 1, fp32): heads within 1e-3 of the oracle, and
    decode+filter+NMS of the GPU's heads identical to the oracle's post-processing of
    those same heads."""
    from yolo_amd import Detector
    C = 80
    for init in ("reference", "variance"):
        bp, hp = _params(C, init, side=416)
        x = _image(1, 416)
        det = Detector(num_classes=C, dtype=torch.float32)
        det.load_reference_params(bp, hp)
        det.to(DEV)
        heads = [h.clone() for h in det.forward_heads(x.to(DEV))]
        want = N.detector_forward(bp, hp, x, C)
        for g, w in zip(heads, want):
            assert rel_err(g.cpu().numpy(), w.numpy()) < 1e-3
        out = det(x.to(DEV))
        got = [t.cpu().numpy() for t in out] + [out.keep_idx.cpu().numpy()]
        heads5 = [h.cpu().numpy().reshape(1, h.shape[1], h.shape[2], 3, 5 + C) for h in heads]
        if init == "variance":
            # decoded values differ from the CPU's by ulps; exactness is asserted on identical
            # decoded inputs in test_gpu_kernels, here the whole chain must agree on the result
            ref = cref.bbox_nms(heads5, 416.0, C)
            np.testing.assert_array_equal(got[3], ref[3])
            np.testing.assert_array_equal(got[4], ref[4])
            np.testing.assert_allclose(got[0], ref[0], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(got[1], ref[1], rtol=1e-4, atol=1e-7)
            np.testing.assert_array_equal(got[2], ref[2])
        else:
            # reference init: every score ties at ~0.25 = the threshold, so survivors are
            # decided by the last ulp of a sigmoid; only structural properties are stable
            assert 0 <= got[3][0] <= 32 and got[0].min() >= 0 and got[0].max() <= 1


def test_full_size_determinism_and_batch_invariance():
    """416x416 bf16: two runs give identical bits; an image's detections do not depend on
    the batch it is in (what makes the batch-sharded multi-GPU result equal the 1-GPU one)."""
    from yolo_amd import Detector
    C = 80
    bp, hp = _params(C, "variance", side=416)
    det = Detector(num_classes=C, dtype=torch.bfloat16)
    det.load_reference_params(bp, hp)
    det.to(DEV)
    x = _image(8, 416).to(DEV)
    a = det(x)
    a = [t.clone() for t in a] + [a.keep_idx.clone()]
    heads_a = [h.clone() for h in det.forward_heads(x)]
    b = det(x)
    b = [t.clone() for t in b] + [b.keep_idx.clone()]
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert int(a[3].sum()) > 0
    for i in (0, 5):
        one = det(x[i:i + 1].contiguous())
        one = [t.clone() for t in one] + [one.keep_idx.clone()]
        for u, v in zip(a, one):
            assert torch.equal(u[i:i + 1], v)
        h1 = det.forward_heads(x[i:i + 1].contiguous())
        for u, v in zip(heads_a, h1):
            assert torch.equal(u[i:i + 1], v)


def test_608_geometry_batch_invariance():
    """608x608 (grids 76/38/19: the widest rows go to the kernels without a resident slab):
    finite outputs, and batch 3 equals three runs of batch 1 bit for bit."""
    from yolo_amd import Detector
    C = 80
    det = Detector(num_classes=C, dtype=torch.bfloat16, seed=3).to(DEV)
    x = _image(3, 608).to(DEV)
    det.calibrate_(x[:2].contiguous())
    out = det(x)
    full = [t.clone() for t in out] + [out.keep_idx.clone()]
    heads = [h.clone() for h in det.forward_heads(x)]
    assert [tuple(h.shape) for h in heads] == [(3, 76, 76, 255), (3, 38, 38, 255), (3, 19, 19, 255)]
    assert all(bool(torch.isfinite(h).all()) for h in heads)
    assert int(full[3].min()) > 0
    for i in range(3):
        one = det(x[i:i + 1].contiguous())
        one = [t.clone() for t in one] + [one.keep_idx.clone()]
        for u, v in zip(full, one):
            assert torch.equal(u[i:i + 1], v)


def test_608_fp32_vs_oracle():
    """Config 3 geometry against the oracle: 608x608, fp32, batch 1, calibrated init.  Head
    maps within 1e-3 of the CPU network; detections equal to the oracle's post-processing
    of those head maps (grids 76/38/19: the wide 3x3 layers at W = 76 run on the streaming
    large-tile kernel instead of the resident-slab one)."""
    from yolo_amd import Detector
    C = 80
    bp, hp = _params(C, "variance", side=64)
    x = _image(1, 608, seed=9)
    det = Detector(num_classes=C, dtype=torch.float32)
    det.load_reference_params(bp, hp)
    det.to(DEV)
    heads = [h.clone() for h in det.forward_heads(x.to(DEV))]
    assert [tuple(h.shape) for h in heads] == [(1, 76, 76, 255), (1, 38, 38, 255), (1, 19, 19, 255)]
    want = N.detector_forward(bp, hp, x, C)
    for g, w in zip(heads, want):
        assert rel_err(g.cpu().numpy(), w.numpy()) < 1e-3
    out = det(x.to(DEV))
    got = [t.cpu().numpy() for t in out] + [out.keep_idx.cpu().numpy()]
    heads5 = [h.cpu().numpy().reshape(1, h.shape[1], h.shape[2], 3, 5 + C) for h in heads]
    ref = cref.bbox_nms(heads5, 608.0, C)
    assert int(ref[3][0]) > 0
    np.testing.assert_array_equal(got[3], ref[3])
    np.testing.assert_array_equal(got[4], ref[4])
    np.testing.assert_array_equal(got[2], ref[2])
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-4, atol=1e-7)


def test_608_bf16_wide_layers_vs_rounding_oracle():
    """The bf16 kernels that 608x608 routes differently from 416x416, at their full layer
    sizes (W = 76 and 38, batch 1), each against the fp32 oracle on bf16-rounded operands."""
    from yolo_amd import _lib
    from yolo_amd.layers import YOLOConv
    shapes = [(1, 76, 76, 512, 1024, 3, False, True), (1, 76, 76, 1024, 512, 1, False, False),
              (1, 76, 76, 128, 256, 3, False, True), (1, 152, 152, 128, 256, 3, True, False),
              (1, 38, 38, 512, 1024, 3, False, True), (1, 152, 152, 64, 128, 3, False, True)]
    L = _lib.lib()
    try:
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 1))
        for i, (B, H, W, cin, cout, k, down, use_res) in enumerate(shapes):
            g = torch.Generator().manual_seed(600 + i)
            xx = N.bf16_round(torch.randn(B, H, W, cin, generator=g))
            p = {"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5),
                 "bias": 0.1 * torch.randn(cout, generator=g)}
            want = N.yolo_conv(xx, p, down, "mish")
            res = None
            if use_res:
                res = N.bf16_round(torch.randn(*want.shape, generator=g))
                want = want + res
            layer = YOLOConv((k, k, cout), downsample=down, activation="mish", dtype=torch.bfloat16)
            layer.load_reference_params(p)
            layer.to(DEV)
            y = layer(xx.to(DEV).bfloat16(), residual=None if res is None else res.to(DEV).bfloat16(),
                      out_dtype=torch.float32)
            assert rel_err(y.cpu().numpy(), want.numpy()) < 1e-3, shapes[i]
    finally:
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 160))


def test_concurrent_sub_batches_equal_whole_batch():
    """A batch cut into sub-batches that run concurrently on separate streams gives the
    same bits as the whole batch as one program: heads, detections and keep indices; twice
    in a row (buffers and streams are reused), then with a different cut."""
    from yolo_amd import Detector
    C = 80
    x = _image(7, 160).to(DEV)
    whole = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split="off").to(DEV)
    whole.calibrate_(x[:2].contiguous())
    want = whole(x)
    want = [t.clone() for t in want] + [want.keep_idx.clone()]
    want_heads = [h.clone() for h in whole.forward_heads(x)]
    assert int(want[3].min()) > 0
    for cut in ((5, 2), (3, 3, 1), (1, 6)):
        det = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split=cut).to(DEV)
        det.calibrate_(x[:2].contiguous())
        for _ in range(2):
            got = det(x)
            plan = det._plan(7, 160, x.dtype, x.device)
            assert plan.sizes == cut and len(plan.parts) == len(cut)
            got = [t.clone() for t in got] + [got.keep_idx.clone()]
            for u, v in zip(want, got):
                assert torch.equal(u, v)
        for u, v in zip(want_heads, det.forward_heads(x)):
            assert torch.equal(u, v)
        # a batch the cut does not sum to runs as one program
        one = det(x[:2].contiguous())
        assert det._plan(2, 160, x.dtype, x.device).sizes == (2,)
        for u, v in zip(want, list(one) + [one.keep_idx]):
            assert torch.equal(u[:2], v)


def test_fused_stem_equals_two_launches_in_the_detector():
    """The planned path runs the backbone's first two layers as one launch (Detector(fuse_stem=True),
    the default): same heads and detections, bit for bit, as with the two separate launches, whole
    batch and concurrent sub-batches, and the 32-channel map between them is never written."""
    from yolo_amd import Detector
    C = 80
    for side, batch, cut in ((160, 3, "off"), (224, 5, (3, 2))):
        x = _image(batch, side).to(DEV)
        outs = []
        for fuse in (False, True):
            det = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split=cut, fuse_stem=fuse).to(DEV)
            det.calibrate_(x[:2].contiguous())
            plan = det._plan(batch, side, x.dtype, x.device)
            assert all(int(part.ops[0].C1) == int(fuse) for part in plan.parts)
            mids = [part.keep[0] for part in plan.parts]        # the first layer's output buffers
            for m in mids:
                assert tuple(m.shape[1:]) == (side, side, 32)
                m.fill_(float("nan"))
            got = det(x)
            heads = det.forward_heads(x)
            torch.cuda.synchronize()
            for m in mids:
                assert bool(torch.isnan(m.float()).all()) == fuse
            outs.append([t.clone() for t in got] + [got.keep_idx.clone()] + [h.clone() for h in heads])
        assert int(outs[0][3].min()) > 0
        for u, v in zip(*outs):
            assert torch.equal(u, v)


def test_decode_beside_matrix_kernels_screen():
    """Bounded screen for results that depend on what else runs on the GPU.  The decode+filter
    launch is repeated on one stream while a second stream keeps the matrix pipes busy (a
    hipBLASLt matmul and small convolutions, which leave room on their CUs for decode
    workgroups); every launch must give the bits of a launch that ran alone.  Then 40 forwards
    of a batch cut into concurrent sub-batches against the whole-batch program.  (A build of
    the decode kernel with hipcc's packed-FMA box corners failed 112 of 300 such launches and
    about one forward in one.)"""
    from yolo_amd import Detector
    from yolo_amd.layers import YOLOConv
    C = 80
    x = _image(7, 160).to(DEV)
    whole = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split="off").to(DEV)
    whole.calibrate_(x[:2].contiguous())
    want = whole(x)
    want = [t.clone() for t in want] + [want.keep_idx.clone()]
    part = whole._plan(7, 160, x.dtype, x.device).parts[0]
    part.boxes.zero_()
    whole._decode_part(part, 160)
    torch.cuda.synchronize()
    ref_boxes, ref_counts, ref_src = part.boxes.clone().view(torch.int32), part.counts.clone(), part.src_idx.clone()
    ma = torch.randn(2048, 2048, device=DEV).bfloat16()
    conv = YOLOConv((3, 3, 1024), dtype=torch.bfloat16, in_channels=512).to(DEV)
    cx = torch.randn(4, 20, 20, 512, device=DEV).bfloat16()
    conv(cx)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = torch.zeros((), device=DEV, dtype=torch.int32)
    n = int(ref_counts.max())
    for i in range(150):
        with torch.cuda.stream(sb):
            for _ in range(2):
                torch.matmul(ma, ma) if i % 2 else conv(cx)
        with torch.cuda.stream(sa):
            whole._decode_part(part, 160)
            bad += ((part.boxes.view(torch.int32) != ref_boxes).any() | (part.counts != ref_counts).any() |
                    (part.src_idx[:, :n] != ref_src[:, :n]).any()).int()
    torch.cuda.synchronize()
    assert int(bad) == 0
    for cut in ((4, 3), (5, 2)):
        det = Detector(num_classes=C, dtype=torch.bfloat16, seed=5, split=cut).to(DEV)
        det.calibrate_(x[:2].contiguous())
        for _ in range(20):
            got = det(x)
            for u, v in zip(want, list(got) + [got.keep_idx]):
                assert torch.equal(u, v)


def test_auto_cut_is_timed_once_and_exact():
    """split="auto" on a batch large enough to be cut: whichever cut the timing picks, the
    results equal the whole-batch program, and the choice is cached per input shape."""
    from yolo_amd import Detector
    C = 80
    x = _image(16, 416).to(DEV)
    det = Detector(num_classes=C, dtype=torch.bfloat16, seed=6).to(DEV)
    det.calibrate_(x[:2].contiguous())
    got = det(x)
    got = [t.clone() for t in got] + [got.keep_idx.clone()]
    plan = det._plan(16, 416, x.dtype, x.device)
    assert sum(plan.sizes) == 16 and plan is det._plan(16, 416, x.dtype, x.device)
    whole = Detector(num_classes=C, dtype=torch.bfloat16, seed=6, split="off").to(DEV)
    whole.calibrate_(x[:2].contiguous())
    want = whole(x)
    for u, v in zip(list(want) + [want.keep_idx], got):
        assert torch.equal(u, v)
    with pytest.raises(ValueError):
        Detector(num_classes=C, split=(4, 0))


def test_prepare_plans_ahead_of_the_first_forward():
    """Detector.prepare does the planning (buffers, timed cut) that the first forward would
    do, reports what it kept, and the forward that follows reuses that plan."""
    from yolo_amd import Detector
    det = Detector(num_classes=3, dtype=torch.bfloat16, seed=2).to(DEV)
    info = det.prepare(4, 96)
    assert info["sub_batches"] == [4] and info["bytes"] > 4 * 96 * 96 * 32 * 2
    plans = dict(det._plans)
    out = det(_image(4, 96).to(DEV))
    assert out[0].shape == (4, 32, 4) and dict(det._plans) == plans
    with pytest.raises(ValueError):
        det.prepare(4, 100)


@pytest.mark.parametrize("S,B", [(416, 64), (608, 32)])
def test_benchmark_batch_properties(S, B):
    """The benchmark configurations themselves (416x416 batch 64 and 608x608 batch 32 per
    GPU, bf16, concurrent sub-batches as the engine chooses them): properties that hold for
    any input size.  Scores sorted,
    padding zero, boxes clipped, every output row is the decoded candidate its keep index
    names, kept boxes of one class do not overlap beyond the threshold, and the C oracle's
    post-processing of the GPU's head maps gives the same keep indices."""
    import yolo_amd as Y
    from yolo_amd import Detector
    from yolo_amd import utils as U
    C = 80
    x = _image(B, S, seed=4).to(DEV)
    det = Detector(num_classes=C, dtype=torch.bfloat16, seed=2).to(DEV)
    det.calibrate_(x[:2].contiguous())
    out = det(x)
    boxes, scores, classes, valid = [t.cpu().numpy() for t in out]
    keep = out.keep_idx.cpu().numpy()
    heads = [h.clone() for h in det.forward_heads(x)]
    assert sum(det._plan(B, S, x.dtype, x.device).sizes) == B
    det.check_status()
    assert valid.min() > 0 and valid.max() <= 32
    cfg = Y.config
    dec = U.decode_heads([h.reshape(B, h.shape[1], h.shape[2], 3, 5 + C) for h in heads], C, cfg["strides"],
                         cfg["xyscale"], U.default_anchors())
    n = dec["N"]
    assert n == {416: 10647, 608: 22743}[S]
    d_boxes = dec["boxes"][:, :n].cpu().numpy()
    d_scores = dec["scores_cm"][:, :, :n].cpu().numpy()              # [B, C, N]
    for b in range(B):
        v = int(valid[b])
        assert np.all(np.diff(scores[b, :v]) <= 0) and np.all(scores[b, :v] > 0.25)
        assert not scores[b, v:].any() and not boxes[b, v:].any() and np.all(keep[b, v:] == -1)
        assert boxes[b].min() >= 0.0 and boxes[b].max() <= 1.0
        k = keep[b, :v]
        c = classes[b, :v].astype(np.int64)
        assert np.array_equal(scores[b, :v], d_scores[b, c, k])
        cx, cy, w, h = d_boxes[b, k].T
        want = np.clip(np.stack([cy - h * 0.5, cx - w * 0.5, cy + h * 0.5, cx + w * 0.5], 1) / np.float32(S), 0, 1)
        np.testing.assert_allclose(boxes[b, :v], want, rtol=0, atol=1e-6)
    # greedy invariant on a few images: same-class kept boxes have IoU <= threshold (unclipped boxes)
    for b in (0, 17, B - 1):
        v = int(valid[b])
        cx, cy, w, h = d_boxes[b, keep[b, :v]].T
        y0, x0, y1, x1 = cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2
        for i in range(v):
            for j in range(i):
                if classes[b, i] == classes[b, j]:
                    iw = max(min(x1[i], x1[j]) - max(x0[i], x0[j]), 0.0)
                    ih = max(min(y1[i], y1[j]) - max(y0[i], y0[j]), 0.0)
                    inter = iw * ih
                    assert inter / (w[i] * h[i] + w[j] * h[j] - inter) <= 0.45 + 1e-5
    # the oracle's filter + NMS on the GPU's decode: identical keep indices (first 4 images)
    sub = slice(0, 4)
    fb, fs, counts, src = cref.filter_bboxes(d_boxes[sub], np.ascontiguousarray(d_scores[sub].transpose(0, 2, 1)), 0.25, float(S))
    want = cref.combined_nms(fb, fs, counts, 32, 32, 0.45, 0.25, src_idx=src)
    np.testing.assert_array_equal(keep[sub], want[4])
    np.testing.assert_array_equal(valid[sub], want[3])
    np.testing.assert_array_equal(classes[sub], want[2])


def test_stress_geometry_scaled():
    """NMS stress geometry (1280^2-equivalent heads, 100 800 candidates, 80 classes,
    threshold 0.001) at batch 2: decode+filter+NMS vs the C oracle on the GPU's decode."""
    import yolo_amd as Y
    from yolo_amd import utils as U
    C, B = 80, 2
    gen = torch.Generator().manual_seed(10)
    heads = []
    for S in (160, 80, 40):
        h = torch.randn(B, S, S, 3, 5 + C, generator=gen) * 2
        h[..., 4] += 6.0
        heads.append(h.to(DEV))
    cfg = Y.config
    dec = U.decode_heads(heads, C, cfg["strides"], cfg["xyscale"], U.default_anchors())
    n = dec["N"]
    assert n == 100800
    d_boxes = dec["boxes"][:, :n].cpu().numpy()
    d_scores = dec["scores_cm"][:, :, :n].transpose(1, 2).cpu().numpy()
    fb, fs, counts, src = cref.filter_bboxes(d_boxes, d_scores, 0.001, 1280.0)
    assert counts.min() > 0.9 * n
    want = cref.combined_nms(fb, fs, counts, 32, 32, 0.45, 0.001, src_idx=src)
    raw = [h.reshape(B, h.shape[1], h.shape[2], -1) for h in heads]
    out = Y.bbox_nms(raw, 1280.0, 32, 32, 0.45, 0.001)
    got = [t.cpu().numpy() for t in out] + [out.keep_idx.cpu().numpy()]
    np.testing.assert_allclose(got[0], want[0], rtol=1e-6, atol=1e-7)
    for a, b in zip(got[1:], want[1:]):
        np.testing.assert_array_equal(a, b)
