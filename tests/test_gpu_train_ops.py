"""GPU parity of the kernels beyond plain inference: the YOLO loss (forward) and the SPP
max-pool block, called through the C ABI (via the Python mirrors), against the CPU oracle.
fp32 tolerance 1e-3 relative on every loss term; max-pooling is exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import loss_np as LN
from oracle import postproc_np as P
from tests_support import loss_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("S,C,scale,B", [(13, 80, 2, 3), (26, 80, 1, 2), (52, 80, 0, 2), (13, 1, 2, 4), (52, 1, 0, 1),
                                         (7, 3, 1, 2)])
def test_yolo_loss_vs_oracle(S, C, scale, B):
    import yolo_amd as Y
    from yolo_amd import losses as L
    conv, label, bboxes = loss_case(B, S, C, scale, seed=100 + S + C, n_gt=6)
    pred = P.decode_train(conv, scale, S, C)
    want = np.array(LN.yolo_loss(pred, conv, label, bboxes, S * P.STRIDES[scale], 0.5), np.float64)
    cfg = Y.config
    target = np.concatenate([label.reshape(B, -1), bboxes.reshape(B, -1)], axis=1)
    t_d = torch.from_numpy(target).to(DEV)
    o_d = torch.from_numpy(conv.reshape(B, S, S, -1)).to(DEV)
    # the closure path: decode fused into the kernel
    got = L.loss_at_scale(t_d, o_d, scale, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"], terms=True)
    np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), want, rtol=1e-3)
    fs = L.get_yolo_loss_at_scales([0, 1, 2], C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"])
    total = fs[2 - scale](t_d, o_d)                      # list order is scales[::-1]
    np.testing.assert_allclose(float(total), want.sum(), rtol=1e-3)
    # the reference's own signature: pred handed in
    g, c, p = L.yolo_loss(torch.from_numpy(pred).to(DEV), torch.from_numpy(conv).to(DEV), torch.from_numpy(label).to(DEV),
                          torch.from_numpy(bboxes).to(DEV), S * P.STRIDES[scale], 0.5)
    np.testing.assert_allclose(np.array([float(g), float(c), float(p)]), want, rtol=1e-3)
    # reproducible bit for bit
    again = L.loss_at_scale(t_d, o_d, scale, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"], terms=True)
    assert torch.equal(got, again)


def test_yolo_loss_late_binding_and_errors():
    import yolo_amd as Y
    from yolo_amd import losses as L
    cfg = Y.config
    B, S, C = 2, 13, 4
    conv, label, bboxes = loss_case(B, S, C, 2, seed=5)
    target = torch.from_numpy(np.concatenate([label.reshape(B, -1), bboxes.reshape(B, -1)], axis=1)).to(DEV)
    out = torch.from_numpy(conv.reshape(B, S, S, -1)).to(DEV)
    late = L.get_yolo_loss_at_scales([0, 1, 2], C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"], late_binding=True)
    want0 = LN.loss_at_scale(target.cpu().numpy(), conv.reshape(B, S, S, -1), 0, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)[0]
    for f in late:                                       # the reference's closures all use scales[0]
        np.testing.assert_allclose(float(f(target, out)), float(want0), rtol=1e-3)
    with pytest.raises(ValueError):
        L.loss_at_scale(target[:, :-4], out, 2, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.loss_at_scale(target.cpu(), out.cpu(), 2, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 13, 13, 512), (1, 19, 19, 64), (3, 7, 5, 24), (1, 40, 40, 16), (2, 3, 4, 8)])
def test_spp_concat_exact(shape, dtype):
    from yolo_amd.layers import spp_concat
    B, H, W, C = shape
    x = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(H * W + C)).to(dtype)
    xn = x.float().permute(0, 3, 1, 2)
    want = torch.cat([F.max_pool2d(xn, k, 1, k // 2) for k in (13, 9, 5)] + [xn], dim=1).permute(0, 2, 3, 1)
    got = spp_concat(x.to(DEV))
    assert got.shape == (B, H, W, 4 * C) and got.dtype == dtype
    assert torch.equal(got.float().cpu(), want)


def test_cspdarknet53_fp32_vs_oracle():
    """The repaired CSPDarknet53 (module API, zero-copy concats, fused SPP) against the torch-CPU
    restatement of the same repaired definition, fp32, 64x64 and 96x96 inputs."""
    from oracle import net_torch as N
    from yolo_amd.backbones import CSPDarknet53
    params = N.init_params(N.cspdarknet53_specs(), seed=3, init="variance")
    net = CSPDarknet53()
    net.load_reference_params(params)
    net.to(DEV)
    for side, batch in ((64, 2), (96, 1)):
        x = torch.rand(batch, side, side, 3, generator=torch.Generator().manual_seed(side))
        want = N.cspdarknet53_forward(params, x)
        got = net(x.to(DEV))
        assert [tuple(t.shape) for t in got] == [(batch, side // 8, side // 8, 256), (batch, side // 16, side // 16, 512),
                                                 (batch, side // 32, side // 32, 512)]
        for g, w in zip(got, want):
            assert g.dtype == torch.float32
            err = float((g.cpu() - w).abs().max() / w.abs().max())
            assert err < 1e-3, err


def test_cspdarknet53_bf16_vs_rounding_oracle():
    from oracle import net_torch as N
    from yolo_amd.backbones import CSPDarknet53
    params = N.init_params(N.cspdarknet53_specs(), seed=4, init="variance")
    net = CSPDarknet53(dtype=torch.bfloat16)
    net.load_reference_params(params)
    net.to(DEV)
    x = torch.rand(2, 64, 64, 3, generator=torch.Generator().manual_seed(7))
    want = N.cspdarknet53_forward(N.round_params_bf16(params), x, round_fn=N.bf16_round)
    got = net(x.to(DEV))
    for g, w in zip(got, want):
        assert g.dtype == torch.bfloat16
        rms = float((g.float().cpu() - w).pow(2).mean().sqrt() / w.pow(2).mean().sqrt())
        assert rms < 2e-2, rms


def test_conv_writes_into_a_channel_slice():
    """YOLOConv(out=...): the output lands in a slice of a wider NHWC buffer, nothing else is touched."""
    from oracle import net_torch as N
    from yolo_amd.layers import YOLOConv, ResidialBlock
    g = torch.Generator().manual_seed(11)
    x = N.bf16_round(torch.randn(2, 9, 7, 16, generator=g))
    p = {"kernel": N.bf16_round(torch.randn(1, 1, 16, 24, generator=g) * 0.2), "bias": torch.randn(24, generator=g)}
    layer = YOLOConv((1, 1, 24), dtype=torch.bfloat16)
    layer.load_reference_params(p)
    layer.to(DEV)
    want = layer(x.to(DEV).bfloat16())
    cat = torch.full((2, 9, 7, 40), 3.0, device=DEV, dtype=torch.bfloat16)
    got = layer(x.to(DEV).bfloat16(), out=cat[..., 8:32])
    assert got.data_ptr() == cat[..., 8:32].data_ptr()
    assert torch.equal(cat[..., 8:32], want) and bool((cat[..., :8] == 3).all()) and bool((cat[..., 32:] == 3).all())
    with pytest.raises(ValueError):
        layer(x.to(DEV).bfloat16(), out=cat[..., 8:30])
    blk = ResidialBlock(8, 16, dtype=torch.bfloat16)
    for c, cin in zip(blk.layer_stack, (16, 8)):
        c.build(cin, generator=g)
    blk.to(DEV)
    want = blk(x.to(DEV).bfloat16())
    blk(x.to(DEV).bfloat16(), out=cat[..., 0:16])
    assert torch.equal(cat[..., 0:16], want)


def _random_boxes(rng, n, size, C):
    wh = np.exp(rng.normal(3.5, 1.0, (n, 2))).clip(4, size * 0.9)
    ctr = rng.uniform(0, size, (n, 2))
    x0y0 = np.clip(ctr - wh / 2, 0, size - 2)
    x1y1 = np.clip(ctr + wh / 2, x0y0 + 1, size - 1)
    cls = rng.integers(0, C, (n, 1))
    return np.floor(np.concatenate([x0y0, x1y1, cls], 1))      # integer pixel corners, as a dataset holds them


@pytest.mark.parametrize("C,size,literal", [(1, 416, False), (80, 416, False), (3, 608, False), (1, 416, True)])
def test_assign_labels_vs_oracle(C, size, literal):
    from oracle import labels_np as LB
    from yolo_amd import dataset as DS
    import yolo_amd as Y
    rng = np.random.default_rng(C + size)
    B, nmax = 5, 40
    counts = np.array([40, 0, 1, 17, 33], np.int32)
    boxes = np.zeros((B, nmax, 5), np.float32)
    for b in range(B):
        boxes[b, :counts[b]] = _random_boxes(rng, counts[b], size, C)
    boxes[0, 5] = boxes[0, 4]                                  # an exact duplicate: the later one overwrites
    boxes[3, 2] = [size + 40, 10, size + 90, 60, 0]            # centre right of the image: skipped whole (no label, no ring slot)
    boxes[3, 7] = [-80, 20, -10, 90, 0]                        # centre left of the image
    boxes[4, 1, 4] = C                                         # class out of range: skipped
    cfg = Y.config
    ag = DS.scale_anchors(literal=literal)
    np.testing.assert_array_equal(ag, LB.scale_anchors(cfg["anchors"], cfg["strides"], literal).astype(np.float32))
    got = DS.preprocess_true_boxes(torch.from_numpy(boxes).to(DEV), torch.from_numpy(counts).to(DEV), size, C,
                                   anchors_grid=ag)
    for b in range(B):
        want = LB.preprocess_true_boxes(boxes[b, :counts[b]], size, C, cfg["strides"], ag.astype(np.float64))
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g[b].cpu().numpy(), w.astype(np.float32))
    assert float(got[0][0].sum()) > 0 and float(got[0][1].abs().sum()) == 0
    tg = DS.pack_targets(*got)
    assert [tuple(t.shape) for t in tg] == [(B, (size // s) ** 2 * 3 * (5 + C) + 512) for s in cfg["strides"]]
    # the targets feed the loss as they are
    from yolo_amd import losses as L
    S = size // 32
    out = torch.randn(B, S, S, 3 * (5 + C), generator=torch.Generator().manual_seed(1)).to(DEV)
    val = L.loss_at_scale(tg[2], out, 2, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"], terms=True)
    want = LN.loss_at_scale(tg[2].cpu().numpy(), out.cpu().numpy(), 2, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)[1]
    np.testing.assert_allclose(val.cpu().numpy().astype(np.float64), np.array(want, np.float64), rtol=1e-3)


@pytest.mark.parametrize("shape,target,literal", [((37, 53), 64, False), ((37, 53), 64, True), ((480, 640), 416, False),
                                                  ((300, 200), 416, True), ((416, 416), 416, False), ((5, 3), 32, False)])
def test_letterbox_bit_exact_vs_oracle(shape, target, literal):
    from oracle import image_np as IM
    from yolo_amd import utils as U
    rng = np.random.default_rng(shape[0] * 7 + target)
    B = 2
    img8 = rng.integers(0, 256, (B, shape[0], shape[1], 3)).astype(np.uint8)
    imgf = rng.uniform(0, 255, (B, shape[0], shape[1], 3)).astype(np.float32)
    boxes = np.array([[3.0, 4.0, 20.0, 30.0, 1.0]], np.float64)
    for src in (img8, imgf):
        want = np.stack([IM.letterbox(src[b], target, literal) for b in range(B)])
        got, bx = U.image_preprocess(torch.from_numpy(src).to(DEV), boxes, target, literal=literal)
        assert got.shape == (B, target, target, 3) and got.dtype == torch.float32
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        g16, _ = U.image_preprocess(torch.from_numpy(src).to(DEV), None, target, literal=literal, dtype=torch.bfloat16)
        assert torch.equal(g16.cpu(), torch.from_numpy(want).to(torch.bfloat16))
    # one image, and the box half of the function (utils.py:29-30)
    one, bx = U.image_preprocess(torch.from_numpy(img8[0]).to(DEV), boxes, target, literal=literal)
    np.testing.assert_array_equal(one.cpu().numpy(), IM.letterbox(img8[0], target, literal))
    scale = min(target / shape[1], target / shape[0])
    pad_h, pad_w = (target - int(scale * shape[0])) // 2, (target - int(scale * shape[1])) // 2
    np.testing.assert_allclose(bx[0], [3 * scale + pad_w, 4 * scale + pad_h, 20 * scale + pad_w, 30 * scale + pad_h, 1.0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        U.image_preprocess(torch.from_numpy(img8), None, target)


@pytest.mark.parametrize("thickness", [1, 2, 3, None])
def test_draw_boxes_vs_oracle(thickness):
    from oracle import image_np as IM
    from yolo_amd import utils as U
    rng = np.random.default_rng(3)
    B, H, W, M, C = 3, 96, 128, 12, 5
    imgs = rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)
    lo = rng.uniform(0, 0.7, (B, M, 2)).astype(np.float32)
    hi = np.minimum(lo + rng.uniform(0.02, 0.5, (B, M, 2)).astype(np.float32), 1.0).astype(np.float32)
    boxes = np.concatenate([lo, hi], -1)                       # (ymin, xmin, ymax, xmax), overlapping freely
    boxes[0, 1] = boxes[0, 0]                                  # identical boxes of different classes: the later wins
    classes = rng.integers(0, C, (B, M)).astype(np.float32)
    classes[1, 2] = C                                          # out of range: skipped
    valid = np.array([M, 0, 7], np.int32)
    t = int(0.6 * (H + W) / 600) if thickness is None else thickness
    want = np.stack([IM.draw_boxes(imgs[b], boxes[b], classes[b], valid[b], C, t) for b in range(B)])
    d = torch.from_numpy(imgs).to(DEV)
    out = U.draw_boxes(d, torch.from_numpy(boxes).to(DEV), torch.from_numpy(classes).to(DEV), torch.from_numpy(valid).to(DEV),
                       C, thickness)
    assert out.data_ptr() == d.data_ptr()
    np.testing.assert_array_equal(d.cpu().numpy(), want)
    assert (want[1] == imgs[1]).all() and (want[0] != imgs[0]).any()


@pytest.mark.parametrize("scale", [1, 2, 3])
def test_draw_boxes_with_labels_vs_oracle(scale):
    """Labels (utils.py:233-240): filled rectangle then black text after each outline, later
    detections over earlier ones, clipped at the image border, exact against the oracle."""
    from oracle import image_np as IM
    from yolo_amd import utils as U
    rng = np.random.default_rng(11 + scale)
    B, H, W, M, C = 3, 120, 160, 10, 6
    imgs = rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)
    lo = rng.uniform(0, 0.7, (B, M, 2)).astype(np.float32)
    hi = np.minimum(lo + rng.uniform(0.02, 0.5, (B, M, 2)).astype(np.float32), 1.0).astype(np.float32)
    boxes = np.concatenate([lo, hi], -1)
    boxes[0, 0] = [0.0, 0.0, 0.5, 0.5]                         # label above the top edge: clipped away
    boxes[0, 1] = [0.3, 0.9, 0.6, 1.0]                         # label runs off the right edge
    boxes[0, 3] = boxes[0, 2]                                  # same corner: the later label wins
    classes = rng.integers(0, C, (B, M)).astype(np.float32)
    classes[2, 1] = C                                          # skipped, label and all
    valid = np.array([M, 0, 6], np.int32)
    names = ["face", "Dog_2", "x", "a~b{c}", "p\xe9", "0123456789"]
    scores = rng.uniform(0, 1, (B, M)).astype(np.float32)
    labels = [[IM.label_text(names[int(c) % C], sc) for c, sc in zip(classes[b][:valid[b]], scores[b])] for b in range(B)]
    labels[0][4] = ""                                          # no label for this one
    labels[2][2] = "W" * 60                                    # cut at LABEL_BYTES by the host side
    want_labels = [[t[:U.LABEL_BYTES] for t in row] for row in labels]
    want = np.stack([IM.draw_boxes(imgs[b], boxes[b], classes[b], valid[b], C, 2, want_labels[b] + [""] * M, scale)
                     for b in range(B)])
    d = torch.from_numpy(imgs).to(DEV)
    U.draw_boxes(d, torch.from_numpy(boxes).to(DEV), torch.from_numpy(classes).to(DEV), torch.from_numpy(valid).to(DEV),
                 C, 2, labels=labels, text_scale=scale)
    np.testing.assert_array_equal(d.cpu().numpy(), want)
    plain = np.stack([IM.draw_boxes(imgs[b], boxes[b], classes[b], valid[b], C, 2) for b in range(B)])
    assert (want != plain).any() and (want[1] == imgs[1]).all()
    assert (want == 0).all(-1).sum() > (plain == 0).all(-1).sum()      # black text pixels appeared


def test_draw_bbox_end_to_end():
    """draw_bbox = bbox_nms on the model output + outlines (utils.py:197-242): the drawn images
    equal the oracle's drawing of the detections the GPU NMS returns."""
    from oracle import image_np as IM
    from yolo_amd import utils as U
    C, B = 3, 2
    g = torch.Generator().manual_seed(0)
    heads = [(torch.randn(B, s, s, 3 * (5 + C), generator=g) * 2).to(DEV) for s in (8, 4, 2)]
    imgs = torch.zeros(B, 64, 64, 3, dtype=torch.uint8, device=DEV)
    det = U.bbox_nms(heads, 64, num_classes=C)
    outs = U.draw_bbox(imgs, heads, 64, C, show_label=False)
    assert len(outs) == B and int(det[3].sum()) > 0
    for b in range(B):
        want = IM.draw_boxes(np.zeros((64, 64, 3), np.uint8), det[0][b].cpu().numpy(), det[2][b].cpu().numpy(),
                             int(det[3][b]), C, int(0.6 * 128 / 600))
        np.testing.assert_array_equal(outs[b].cpu().numpy(), want)
    # with captions '<name>: <score>' (utils.py:234), names defaulting to the class indices
    names = ["ant", "bee", "cat"]
    for cls_names, shown in ((None, ["0", "1", "2"]), (names, names)):
        imgs2 = torch.full((B, 64, 64, 3), 200, dtype=torch.uint8, device=DEV)
        outs2 = U.draw_bbox(imgs2, heads, 64, C, classes=cls_names, text_scale=1)
        for b in range(B):
            n = int(det[3][b])
            cl, sc = det[2][b].cpu().numpy(), det[1][b].float().cpu().numpy()
            labels = [IM.label_text(shown[int(cl[j])], sc[j]) for j in range(n)]
            want = IM.draw_boxes(np.full((64, 64, 3), 200, np.uint8), det[0][b].cpu().numpy(), cl, n, C,
                                 int(0.6 * 128 / 600), labels, 1)
            np.testing.assert_array_equal(outs2[b].cpu().numpy(), want)


@pytest.mark.parametrize("S,C,scale,B", [(13, 80, 2, 3), (26, 80, 1, 2), (52, 3, 0, 2), (7, 1, 1, 2)])
def test_yolo_loss_gradient_vs_autograd_oracle(S, C, scale, B):
    """d loss / d head output from the fused kernel (through torch.autograd) against the
    automatic gradient of the torch restatement, fp32 tolerance 1e-3 of the gradient's scale
    per tensor and 2e-3 element by element on the responsible rows; per-term upstream weights."""
    import yolo_amd as Y
    from yolo_amd import losses as L
    from oracle import loss_torch as LT
    cfg = Y.config
    conv, label, bboxes = loss_case(B, S, C, scale, seed=300 + S + C, n_gt=6)
    target = np.concatenate([label.reshape(B, -1), bboxes.reshape(B, -1)], axis=1)
    out = conv.reshape(B, S, S, -1)
    t_d = torch.from_numpy(target).to(DEV)
    for weights in ((1.0, 1.0, 1.0), (0.5, 2.0, 3.0)):
        o_d = torch.from_numpy(out).to(DEV).requires_grad_(True)
        terms = L.loss_at_scale(t_d, o_d, scale, C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"], terms=True)
        (terms * torch.tensor(weights, device=DEV)).sum().backward()
        (g, c, p), want = LT.loss_and_grad(target, out, scale, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS, weights)
        np.testing.assert_allclose(terms.detach().cpu().numpy().astype(np.float64), [g, c, p], rtol=1e-3)
        got = o_d.grad.cpu().numpy().astype(np.float64)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()
        D = 5 + C
        rows = label[..., 4].reshape(B, -1) > 0
        gr, wr = got.reshape(B, -1, D)[rows], want.reshape(B, -1, D)[rows]
        assert rows.sum() > 0 and np.abs(gr - wr).max() <= 2e-3 * np.abs(wr).max()
        # nothing depends on the class and box logits of rows that are not responsible
        assert (got.reshape(B, -1, D)[~rows][:, [0, 1, 2, 3] + list(range(5, D))] == 0).all()
    # the summed closure is differentiable too, and the gradient is reproducible bit for bit
    fs = L.get_yolo_loss_at_scales([0, 1, 2], C, cfg["strides"], 0.5, cfg["xyscale"], cfg["anchors"])
    grads = []
    for _ in range(2):
        o_d = torch.from_numpy(out).to(DEV).requires_grad_(True)
        fs[2 - scale](t_d, o_d).backward()
        grads.append(o_d.grad.clone())
    assert torch.equal(grads[0], grads[1])
    (g, c, p), want = LT.loss_and_grad(target, out, scale, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)
    assert np.abs(grads[0].cpu().numpy() - want).max() <= 1e-3 * np.abs(want).max()
