"""CPU tests of the oracle itself: hand-derived known answers, the numpy and C
restatements against each other, and both against the committed golden fixtures.

The reference has no tests or golden vectors (it cannot be imported here either:
TensorFlow is absent), so nothing below pins TensorFlow's arithmetic: parity unpinned.
"""
import glob
import math
import os

import numpy as np
import pytest
import torch

from oracle import cref
from oracle import net_torch as N
from oracle import postproc_np as P

F = np.float32


# ----------------------------------------------------------------------------- decode
def test_decode_known_answers():
    """utils.py:107-126 by hand on a 2x2 grid, 2 classes, scale 0 (stride 8, xyscale 1.2,
    anchors (12,16),(19,36),(40,28))."""
    t = np.zeros((1, 2, 2, 3, 7), F)
    # cell (row 0, col 1), anchor 2: t_x = ln 3 -> sigmoid 0.75 ; t_w = ln 2 ; conf logit ln 3
    t[0, 0, 1, 2, 0] = math.log(3.0)
    t[0, 0, 1, 2, 2] = math.log(2.0)
    t[0, 0, 1, 2, 4] = math.log(3.0)
    t[0, 0, 1, 2, 6] = -math.log(3.0)
    xywh, conf, prob = P.decode(t, 0, 2, 2)
    # all-zero logits: sigmoid 0.5 -> (0.5*1.2 - 0.1 + g)*8 = (0.5+g)*8 ; wh = anchor
    np.testing.assert_allclose(xywh[0, 1, 0, 0], [4.0, 12.0, 12.0, 16.0], rtol=1e-6)   # row 1, col 0
    np.testing.assert_allclose(xywh[0, 1, 1, 1], [12.0, 12.0, 19.0, 36.0], rtol=1e-6)
    # the edited cell: x = (0.75*1.2 - 0.1 + 1)*8 = 14.4 ; y = (0.5+0)*8 = 4 ; w = 2*40 ; h = 28
    np.testing.assert_allclose(xywh[0, 0, 1, 2], [14.4, 4.0, 80.0, 28.0], rtol=1e-6)
    np.testing.assert_allclose(conf[0, 0, 1, 2, 0], 0.75, rtol=1e-6)
    np.testing.assert_allclose(prob[0, 0, 1, 2], [0.5, 0.25], rtol=1e-6)
    bx, sc = P.decode_predict(t, 0, 2, 2)
    assert bx.shape == (1, 12, 4) and sc.shape == (1, 12, 2)
    # flatten order (row, col, anchor): row 0, col 1, anchor 2 -> index 5
    np.testing.assert_allclose(sc[0, 5], [0.375, 0.1875], rtol=1e-6)
    np.testing.assert_allclose(sc[0, 0], [0.25, 0.25], rtol=1e-6)
    tr = P.decode_train(t, 0, 2, 2)
    assert tr.shape == (1, 2, 2, 3, 7)
    np.testing.assert_allclose(tr[0, 0, 1, 2], [14.4, 4.0, 80.0, 28.0, 0.75, 0.5, 0.25], rtol=1e-6)


def test_decode_scales_use_their_own_constants():
    t = np.zeros((1, 1, 1, 3, 6), F)
    for s, (stride, anc) in enumerate(((8, (12, 16)), (16, (36, 75)), (32, (142, 110)))):
        xywh, _, _ = P.decode(t, s, 1, 1)
        np.testing.assert_allclose(xywh[0, 0, 0, 0], [0.5 * stride, 0.5 * stride, anc[0], anc[1]], rtol=1e-6)


def test_filter_known_answers():
    """utils.py:142-167: >= threshold, order kept, (cy-h/2, cx-w/2, cy+h/2, cx+w/2)/size."""
    boxes = np.array([[[10, 20, 4, 8], [30, 40, 10, 20], [50, 60, 2, 2], [5, 5, 2, 4]]], F)
    scores = np.array([[[0.1, 0.25], [0.2, 0.1], [0.9, 0.0], [0.25, 0.3]]], F)
    fb, fs, counts, src = P.filter_bboxes(boxes, scores, 0.25, 100.0)
    assert counts.tolist() == [3] and src.tolist() == [[0, 2, 3]]
    np.testing.assert_allclose(fb[0, 0], [0.16, 0.08, 0.24, 0.12], rtol=1e-6)
    np.testing.assert_allclose(fb[0, 1], [0.59, 0.49, 0.61, 0.51], rtol=1e-6)
    np.testing.assert_array_equal(fs[0], scores[0, [0, 2, 3]])
    # per-image counts differ, padding is zero / -1
    boxes2 = np.concatenate([boxes, boxes], 0)
    scores2 = np.concatenate([scores, scores * 0.5], 0)
    fb, fs, counts, src = P.filter_bboxes(boxes2, scores2, 0.25, 100.0)
    assert counts.tolist() == [3, 1] and src[1].tolist() == [2, -1, -1]
    assert np.all(fb[1, 1:] == 0) and np.all(fs[1, 1:] == 0)


# ----------------------------------------------------------------------------- NMS
def _nms(boxes, scores, **kw):
    args = dict(max_output_size_per_class=32, max_total_size=32, iou_threshold=0.45, score_threshold=0.25)
    args.update(kw)
    b = np.asarray(boxes, F)[None]
    s = np.asarray(scores, F)[None]
    a = P.combined_nms(b, s, **args)
    c = cref.combined_nms(b, s, **args)
    for x, y in zip(a, c):
        np.testing.assert_array_equal(x, y)
    return a


def test_nms_identical_boxes_keep_first():
    ob, os_, oc, ov, ok = _nms([[0, 0, .5, .5]] * 3, [[.9], [.8], [.7]])
    assert ov[0] == 1 and ok[0, 0] == 0 and ok[0, 1] == -1 and os_[0, 0] == F(0.9)


def test_nms_iou_exactly_at_threshold_is_kept():
    # A = 0.5 x 1.0, B = 0.5 x 0.5 inside A: inter 0.25, union 0.5, IoU = 0.5 exactly
    boxes = [[0, 0, .5, 1.0], [0, 0, .5, .5]]
    assert P.nms_iou(np.array(boxes[0], F), np.array([boxes[1]], F))[0] == F(0.5)
    _, _, _, ov, ok = _nms(boxes, [[.9], [.8]], iou_threshold=0.5)
    assert ov[0] == 2 and ok[0, :2].tolist() == [0, 1]
    _, _, _, ov, _ = _nms(boxes, [[.9], [.8]], iou_threshold=0.49)
    assert ov[0] == 1


def test_nms_score_threshold_is_strict():
    _, _, _, ov, ok = _nms([[0, 0, .1, .1], [.5, .5, .6, .6]], [[.25], [.26]], score_threshold=0.25)
    assert ov[0] == 1 and ok[0, 0] == 1


def test_nms_ties_break_by_index_then_class():
    boxes = [[0, 0, .1, .1], [.5, .5, .6, .6], [.8, .8, .9, .9]]
    # same score everywhere: per class index order; merged by class asc then index asc
    scores = [[.5, .5], [.5, .5], [.5, .5]]
    _, _, oc, ov, ok = _nms(boxes, scores)
    assert ov[0] == 6
    assert oc[0, :6].tolist() == [0, 0, 0, 1, 1, 1] and ok[0, :6].tolist() == [0, 1, 2, 0, 1, 2]
    # within a class a tie is won by the lower index: box 1 overlaps box 0 and loses
    boxes = [[0, 0, .5, .5], [0, 0, .5, .45], [.6, .6, .9, .9]]
    _, _, _, ov, ok = _nms(boxes, [[.7], [.7], [.7]])
    assert ok[0, :2].tolist() == [0, 2] and ov[0] == 2


def test_nms_zero_area_and_flipped_boxes():
    # zero-area boxes never overlap anything; flipped corners are normalised for IoU only
    boxes = [[.2, .2, .2, .6], [.2, .2, .2, .6], [.5, .5, 0, 0], [0, 0, .5, .5]]
    ob, _, _, ov, ok = _nms(boxes, [[.9], [.8], [.7], [.6]])
    assert ov[0] == 3 and ok[0, :3].tolist() == [0, 1, 2]      # box 3 == box 2 flipped -> suppressed
    np.testing.assert_array_equal(ob[0, 2], np.array([.5, .5, 0, 0], F))   # emitted as given (clipped)


def test_nms_clips_and_caps():
    boxes = [[-.2, -.1, .3, 1.4]] + [[.1 * i, .1 * i, .1 * i + .05, .1 * i + .05] for i in range(1, 9)]
    scores = [[.95]] + [[.9 - .01 * i] for i in range(1, 9)]
    ob, os_, _, ov, ok = _nms(boxes, scores, max_output_size_per_class=5, max_total_size=3)
    assert ov[0] == 3 and ok[0].tolist() == [0, 1, 2]
    np.testing.assert_allclose(ob[0, 0], [0, 0, .3, 1.0])
    _, _, _, ov, _ = _nms(boxes, scores, max_output_size_per_class=2, max_total_size=8)
    assert ov[0] == 2


def test_nms_counts_and_src_idx():
    rng = np.random.default_rng(3)
    boxes = rng.random((2, 20, 4)).astype(F)
    scores = rng.random((2, 20, 3)).astype(F)
    counts = np.array([20, 7], np.int32)
    src = np.arange(40, dtype=np.int32).reshape(2, 20) * 3
    a = P.combined_nms(boxes, scores, counts, 4, 6, 0.3, 0.2, src_idx=src)
    c = cref.combined_nms(boxes, scores, counts, 4, 6, 0.3, 0.2, src_idx=src)
    for x, y in zip(a, c):
        np.testing.assert_array_equal(x, y)
    assert np.all(a[4][1][a[4][1] >= 0] < 60 + 7 * 3)          # image 1 only uses its first 7
    empty = P.combined_nms(boxes, scores, np.zeros(2, np.int32), 4, 6, 0.3, 0.2)
    assert empty[3].tolist() == [0, 0] and np.all(empty[4] == -1) and np.all(empty[0] == 0)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_numpy_and_c_restatements_agree(seed):
    rng = np.random.default_rng(seed)
    C = [5, 80, 1][seed]
    heads = [rng.normal(0, 2, (2, S, S, 3, 5 + C)).astype(F) for S in (8, 4, 2)]
    b1, s1 = P.decode_all(heads, C)
    b2, s2 = cref.decode_all(heads, C)
    np.testing.assert_allclose(b1, b2, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(s1, s2, rtol=2e-6, atol=1e-7)
    f1 = P.filter_bboxes(b1, s1, 0.25, 64)
    f2 = cref.filter_bboxes(b1, s1, 0.25, 64)
    for x, y in zip(f1, f2):
        np.testing.assert_array_equal(x, y)
    n1 = P.combined_nms(f1[0], f1[1], f1[2], 8, 16, 0.45, 0.25, src_idx=f1[3])
    n2 = cref.combined_nms(f1[0], f1[1], f1[2], 8, 16, 0.45, 0.25, src_idx=f1[3])
    for x, y in zip(n1, n2):
        np.testing.assert_array_equal(x, y)


def test_bbox_iou_known_answers():
    a = np.array([[5, 5, 10, 10]], F)          # [0,10]^2
    b = np.array([[10, 5, 10, 10]], F)         # x in [5,15]: inter 50, union 150
    np.testing.assert_allclose(P.bbox_iou(a, b), [1 / 3], rtol=1e-6)
    np.testing.assert_allclose(P.bbox_giou(a, b), [1 / 3 - 0.0], rtol=1e-6)   # enclosing box == union hull
    z = np.zeros((1, 4), F)
    assert P.bbox_iou(z, z)[0] == 0 and P.bbox_giou(z, z)[0] == 0             # divide_no_nan
    c = np.array([[30, 5, 10, 10]], F)         # disjoint: iou 0, hull 35x10, union 200 -> giou = -(350-200)/350
    np.testing.assert_allclose(P.bbox_giou(a, c), [-150 / 350], rtol=1e-6)
    assert P.bbox_iou(a[:, None], np.stack([b[0], c[0]])[None]).shape == (1, 2)


# ----------------------------------------------------------------------------- conv net
def test_conv_padding_semantics():
    """stride 1 = SAME; downsample = one zero row/col on top/left, VALID, stride 2 (layers.py:46-57)."""
    x = torch.arange(16, dtype=torch.float32).reshape(1, 4, 4, 1)
    p = {"kernel": torch.ones(3, 3, 1, 1), "bias": torch.zeros(1)}
    y = N.yolo_conv(x, p, False, None)
    assert y.shape == (1, 4, 4, 1)
    assert y[0, 0, 0, 0] == 0 + 1 + 4 + 5 and y[0, 3, 3, 0] == 10 + 11 + 14 + 15
    y = N.yolo_conv(x, p, True, None)
    assert y.shape == (1, 2, 2, 1)
    # output (0,0) covers rows/cols -1..1 of x -> x[0:2,0:2]; output (1,1) covers rows/cols 1..3
    assert y[0, 0, 0, 0] == 0 + 1 + 4 + 5
    assert y[0, 1, 1, 0] == sum(float(x[0, r, c, 0]) for r in (1, 2, 3) for c in (1, 2, 3))


def test_activations_and_bn():
    v = torch.tensor([-30.0, -1.0, 0.0, 1.0, 30.0])
    m = N.mish(v)
    np.testing.assert_allclose(m[3].item(), 1.0 * math.tanh(math.log1p(math.e)), rtol=1e-6)
    assert abs(m[0].item()) < 1e-10 and m[4].item() == 30.0
    np.testing.assert_allclose(N.activation(v, "leaky").numpy(), [-3.0, -0.1, 0.0, 1.0, 30.0], rtol=1e-6)
    with pytest.raises(ValueError):
        N.activation(v, "relu")
    x = torch.ones(1, 1, 1, 2)
    p = {"kernel": torch.ones(1, 1, 2, 1), "gamma": torch.tensor([2.0]), "beta": torch.tensor([0.5]),
         "mean": torch.tensor([1.0]), "var": torch.tensor([4.0 - 1e-3])}
    y = N.yolo_conv(x, p, False, None)          # conv = 2 ; (2-1)/sqrt(4)*2 + 0.5 = 1.5
    np.testing.assert_allclose(y.item(), 1.5, rtol=1e-6)


def test_upsample_half_pixel():
    x = torch.tensor([0.0, 4.0]).reshape(1, 1, 2, 1)
    y = N.upsample2x(x)
    np.testing.assert_allclose(y[0, :, :, 0].numpy(), [[0, 1, 3, 4], [0, 1, 3, 4]], rtol=1e-6)


def test_network_shapes_and_sizes():
    specs = N.darknet53_specs() + N.yolov3_specs(80)
    assert len(N.darknet53_specs()) == 52 and len(N.yolov3_specs(80)) == 21
    assert sum(k * k * ci * co + co for k, ci, co, _, _ in specs) == 88419869
    bp = N.init_params(N.darknet53_specs(), seed=0)
    hp = N.init_params(N.yolov3_specs(2), seed=1)
    x = torch.rand(1, 64, 64, 3, generator=torch.Generator().manual_seed(1))
    s1, s2, s3 = N.darknet53_forward(bp, x)
    assert s1.shape == (1, 8, 8, 256) and s2.shape == (1, 4, 4, 512) and s3.shape == (1, 2, 2, 1024)
    s, m, l = N.yolov3_forward(hp, s1, s2, s3, 2)
    assert s.shape == (1, 8, 8, 21) and m.shape == (1, 4, 4, 21) and l.shape == (1, 2, 2, 21)


# ----------------------------------------------------------------------------- goldens
def test_postproc_goldens(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "postproc_*.npz")))
    assert len(files) == 3
    for f in files:
        g = np.load(f)
        C = int(g["C"])
        heads = [g["head%d" % i] for i in range(3)]
        for impl in (P, cref):
            xywh, sc = impl.decode_all(heads, C)
            np.testing.assert_allclose(xywh, g["xywh"], rtol=3e-6, atol=1e-6)
            np.testing.assert_allclose(sc, g["scores"], rtol=3e-6, atol=1e-7)
            fb, fs, counts, src = impl.filter_bboxes(g["xywh"], g["scores"], float(g["thr"]), float(g["image_size"]))
            np.testing.assert_array_equal(counts, g["f_counts"])
            np.testing.assert_array_equal(src, g["f_src"])
            np.testing.assert_array_equal(fb, g["f_boxes"])
            np.testing.assert_array_equal(fs, g["f_scores"])
            out = impl.combined_nms(g["f_boxes"], g["f_scores"], g["f_counts"], int(g["per_class"]),
                                    int(g["total"]), float(g["iou"]), float(g["thr"]), src_idx=g["f_src"])
            for got, key in zip(out, ("n_boxes", "n_scores", "n_classes", "n_valid", "n_keep")):
                np.testing.assert_array_equal(got, g[key])


def test_nms_goldens(golden_dir):
    for name in ("nms_dense", "nms_ties"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        for impl in (P, cref):
            out = impl.combined_nms(g["boxes"], g["scores"], g["counts"], int(g["per_class"]), int(g["total"]),
                                    float(g["iou"]), float(g["thr"]))
            for got, key in zip(out, ("n_boxes", "n_scores", "n_classes", "n_valid", "n_keep")):
                np.testing.assert_array_equal(got, g[key])
        assert g["n_valid"].min() > 0


def test_conv_goldens(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "conv_*.npz")))
    assert len(files) == 5
    for f in files:
        g = np.load(f)
        p = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p_")}
        act = str(g["act"])
        y = N.yolo_conv(torch.from_numpy(g["x"]), p, bool(g["down"]), None if act == "none" else act)
        if "residual" in g.files:
            y = y + torch.from_numpy(g["residual"])
        np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=1e-5)


# ----------------------------------------------------------------------------- loss
def _scalar_loss_terms(conv, label, bboxes, scale, S, C, strides, xyscale, anchors, thr):
    """The loss of /root/reference/losses.py:10-66 spelled out candidate by candidate in
    plain Python floats (double precision): an independent restatement for small cases."""
    import math
    sig = lambda v: 1.0 / (1.0 + math.exp(-v))
    sce = lambda z, x: max(x, 0.0) - x * z + math.log1p(math.exp(-abs(x)))

    def iou_giou(b1, b2):
        a1, a2 = b1[2] * b1[3], b2[2] * b2[3]
        l1, t1, r1, d1 = b1[0] - b1[2] / 2, b1[1] - b1[3] / 2, b1[0] + b1[2] / 2, b1[1] + b1[3] / 2
        l2, t2, r2, d2 = b2[0] - b2[2] / 2, b2[1] - b2[3] / 2, b2[0] + b2[2] / 2, b2[1] + b2[3] / 2
        iw, ih = max(min(r1, r2) - max(l1, l2), 0.0), max(min(d1, d2) - max(t1, t2), 0.0)
        inter = iw * ih
        union = a1 + a2 - inter
        iou = inter / union if union != 0 else 0.0
        ew, eh = max(r1, r2) - min(l1, l2), max(d1, d2) - min(t1, t2)
        enc = ew * eh
        return iou, iou - ((enc - union) / enc if enc != 0 else 0.0)

    B = conv.shape[0]
    input_size = S * strides[scale]
    tot = [0.0, 0.0, 0.0]
    for b in range(B):
        for r in range(S):
            for c in range(S):
                for a in range(3):
                    t = [float(v) for v in conv[b, r, c, a]]
                    lab = [float(v) for v in label[b, r, c, a]]
                    xs, st = xyscale[scale], strides[scale]
                    px = (sig(t[0]) * xs - 0.5 * (xs - 1) + c) * st
                    py = (sig(t[1]) * xs - 0.5 * (xs - 1) + r) * st
                    pw = math.exp(t[2]) * float(anchors[scale][a][0])
                    ph = math.exp(t[3]) * float(anchors[scale][a][1])
                    pconf = sig(t[4])
                    respond = lab[4]
                    _, giou = iou_giou((px, py, pw, ph), lab[:4])
                    tot[0] += respond * (2.0 - lab[2] * lab[3] / input_size ** 2) * (1 - giou)
                    best = max(iou_giou((px, py, pw, ph), [float(v) for v in bboxes[b, j]])[0] for j in range(bboxes.shape[1]))
                    bgd = (1 - respond) * (1.0 if best < thr else 0.0)
                    bce = sce(respond, t[4])
                    tot[1] += (respond - pconf) ** 2 * (respond * bce + bgd * bce)
                    tot[2] += respond * sum(sce(lab[5 + k], t[5 + k]) for k in range(C))
    return [v / B for v in tot]


from tests_support import loss_case as _loss_case      # noqa: E402


def test_loss_oracle_matches_scalar_restatement():
    from oracle import loss_np as LN
    for (B, S, C, scale, seed) in ((2, 3, 2, 0, 1), (1, 2, 5, 2, 2), (2, 4, 1, 1, 3)):
        conv, label, bboxes = _loss_case(B, S, C, scale, seed)
        pred = P.decode_train(conv, scale, S, C)
        got = LN.yolo_loss(pred, conv, label, bboxes, S * P.STRIDES[scale], 0.5)
        want = _scalar_loss_terms(conv, label, bboxes, scale, S, C, P.STRIDES, P.XYSCALE, P.ANCHORS, 0.5)
        np.testing.assert_allclose(np.array(got, np.float64), np.array(want), rtol=2e-5)
        assert all(np.isfinite(v) for v in want) and want[1] > 0 and want[2] > 0
        # the closure body on the flat target layout gives the same total
        target = np.concatenate([label.reshape(B, -1), bboxes.reshape(B, -1)], axis=1)
        tot, terms = LN.loss_at_scale(target, conv.reshape(B, S, S, -1), scale, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)
        np.testing.assert_allclose(float(tot), sum(want), rtol=2e-5)


def test_loss_known_answers():
    """Hand-derived: one cell, zero logits, no responsible anchor -> only the background
    objectness term: each of the 3 anchors gives (0 - 0.5)^2 * log 2; with a ground-truth box
    overlapping an anchor's prediction by IoU >= 0.5 that anchor drops out."""
    from oracle import loss_np as LN
    C, S, scale = 1, 1, 2
    conv = np.zeros((1, 1, 1, 3, 6), np.float32)
    label = np.zeros_like(conv)
    bboxes = np.zeros((1, 128, 4), np.float32)
    pred = P.decode_train(conv, scale, S, C)
    g, c, p = LN.yolo_loss(pred, conv, label, bboxes, 32, 0.5)
    assert g == 0 and p == 0
    np.testing.assert_allclose(c, 3 * 0.25 * np.log(2.0), rtol=1e-6)
    bboxes[0, 0] = pred[0, 0, 0, 1, :4]                 # exactly anchor 1's predicted box: IoU 1 with it
    g, c, p = LN.yolo_loss(pred, conv, label, bboxes, 32, 0.5)
    np.testing.assert_allclose(c, 2 * 0.25 * np.log(2.0), rtol=1e-6)
    # a responsible anchor whose prediction equals its label: GIoU 1 -> no box loss, objectness (1-0.5)^2 log 2
    label[0, 0, 0, 1, :4] = pred[0, 0, 0, 1, :4]
    label[0, 0, 0, 1, 4] = 1.0
    label[0, 0, 0, 1, 5] = 1.0
    g, c, p = LN.yolo_loss(pred, conv, label, bboxes, 32, 0.5)
    np.testing.assert_allclose(g, 0.0, atol=1e-6)
    np.testing.assert_allclose(p, np.log(2.0), rtol=1e-6)
    # late binding: every closure of the reference uses scales[0]
    fs = LN.get_yolo_loss_at_scales([0, 1, 2], C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)
    fs2 = LN.get_yolo_loss_at_scales([0, 1, 2], C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS, late_binding=False)
    target = np.concatenate([label.reshape(1, -1), bboxes.reshape(1, -1)], axis=1)
    out = conv.reshape(1, 1, 1, -1)
    assert fs[0](target, out) == fs[1](target, out) == fs[2](target, out) == fs2[2](target, out)
    assert fs2[0](target, out) != fs2[2](target, out)


def test_label_assignment_known_answers():
    """Hand-derived: a 64x64 box centred at (100, 60) in a 416 image.  Grid cells: stride 8 ->
    (12, 7), 16 -> (6, 3), 32 -> (3, 1).  In grid units the box is 8x8 / 4x4 / 2x2; against the
    stride-16 anchors (36,75)/16, (76,55)/16, (72,146)/16 = 2.25x4.69, 4.75x3.44, 4.5x9.1 the
    IoUs are inter/union with the anchor centred on the cell centre."""
    from oracle import labels_np as LB
    ag = LB.scale_anchors(P.ANCHORS.reshape(-1), P.STRIDES)
    box = np.array([[68, 28, 132, 92, 0]], np.float64)
    ls, lm, ll, bs, bm, bl = LB.preprocess_true_boxes(box, 416, 2, P.STRIDES, ag)
    assert ls.shape == (52, 52, 3, 7) and lm.shape == (26, 26, 3, 7) and ll.shape == (13, 13, 3, 7)
    # stride 16: box (6.25, 3.75, 4, 4) vs anchor 1 (4.75 x 3.4375) centred (6.5, 3.5)
    iw = min(6.25 + 2, 6.5 + 2.375) - max(6.25 - 2, 6.5 - 2.375)
    ih = min(3.75 + 2, 3.5 + 1.71875) - max(3.75 - 2, 3.5 - 1.71875)
    iou = iw * ih / (16 + 4.75 * 3.4375 - iw * ih)
    assert iou > 0.3
    np.testing.assert_allclose(lm[3, 6, 1], [100, 60, 64, 64, 1, 0.995, 0.005])
    np.testing.assert_allclose(bm[0], [100, 60, 64, 64])
    assert lm[..., 4].sum() >= 1 and bm[1:].sum() == 0
    # every responsible row carries the same box and smoothed one-hot
    for lab in (ls, lm, ll):
        rows = lab[lab[..., 4] == 1]
        assert np.all(rows[:, :4] == [100, 60, 64, 64]) and np.allclose(rows[:, 5:], [0.995, 0.005])
    # a tiny box matches no anchor above 0.3: exactly one row overall, at the best anchor
    tiny = np.array([[200, 200, 203, 203, 1]], np.float64)
    out = LB.preprocess_true_boxes(tiny, 416, 2, P.STRIDES, ag)
    assert sum(int(l[..., 4].sum()) for l in out[:3]) == 1 and int(out[0][..., 4].sum()) == 1
    assert out[0][25, 25, 0, 4] == 1                    # stride 8, cell (25, 25), the smallest anchor
    # the literal reading of dataset.py:51,440: one number per scale
    lit = LB.scale_anchors(P.ANCHORS.reshape(-1), P.STRIDES, literal=True)
    assert lit[0].tolist() == [[12, 12]] * 3 and lit[2].tolist() == [[19, 19]] * 3
    # letterbox bookkeeping (utils.py:19-31): 200 x 400 (h x w) into 416: scale 1.04, pad_y 104, pad_x 0
    bx, scale, px, py = LB.letterbox_boxes(np.array([[10., 20., 110., 120., 0.]]), 200, 400, 416)
    assert abs(scale - 1.04) < 1e-12 and (px, py) == (0, 104)
    np.testing.assert_allclose(bx[0, :4], [10.4, 124.8, 114.4, 228.8])


# ---- image utilities (utils.py:5-31, 197-242) -----------------------------------------------------
def test_letterbox_known_answers():
    from oracle import image_np as IM
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (4, 4, 3)).astype(np.uint8)
    np.testing.assert_array_equal(IM.letterbox(img, 4), img.astype(np.float32) / 256)      # same size: untouched
    # 2 -> 4 along one axis: the weights of half-pixel-centre bilinear resize with replicated edges
    a, b = 10.0, 50.0
    line = np.array([[[a] * 3, [b] * 3]] * 2, np.float32)                               # [2,2,3], constant down the rows
    out = IM.resize_linear(line, 4, 4)
    np.testing.assert_allclose(out[0, :, 0], [a, 0.75 * a + 0.25 * b, 0.25 * a + 0.75 * b, b])
    assert np.ptp(out[:, 1, 0]) == 0
    # a 2 x 4 image in a 4 x 4 square: scale 1, one padding row above and below, value 128 / 256
    wide = rng.integers(0, 256, (2, 4, 3)).astype(np.uint8)
    o = IM.letterbox(wide, 4)
    np.testing.assert_array_equal(o[1:3], wide.astype(np.float32) / 256)
    assert (o[0] == 0.5).all() and (o[3] == 0.5).all()
    # the reference's argument order: the resized image has new_width rows and new_height columns
    q = IM.letterbox(wide, 4, literal=True)
    assert (q[:, 0] == 0.5).all() and (q[:, 3] == 0.5).all() and not (q[:, 1:3] == 0.5).all()
    np.testing.assert_array_equal(q[:, 1:3], IM.resize_linear(wide, 4, 2) / np.float32(256))
    # square images do not see the difference
    sq = rng.integers(0, 256, (6, 6, 3)).astype(np.uint8)
    np.testing.assert_array_equal(IM.letterbox(sq, 8), IM.letterbox(sq, 8, literal=True))


def test_draw_boxes_known_answer():
    from oracle import image_np as IM
    img = np.zeros((8, 8, 3), np.uint8)
    boxes = np.array([[0.25, 0.25, 0.75, 0.75], [0.0, 0.0, 0.5, 0.5]], np.float32)
    out = IM.draw_boxes(img, boxes, np.array([0.0, 2.0]), 1, 4, 1)
    ring = np.zeros((8, 8), bool)
    ring[2, 2:7] = ring[6, 2:7] = ring[2:7, 2] = ring[2:7, 6] = True
    assert ring.sum() == 16
    assert (out[ring] == [255, 0, 0]).all() and (out[~ring] == 0).all()                  # class 0 of 4: hue 0 = red
    two = IM.draw_boxes(img, boxes, np.array([0.0, 2.0]), 2, 4, 1)
    assert (two[4, 2] == [0, 255, 255]).all() and (two[2, 4] == [0, 255, 255]).all()    # the later box paints over the earlier
    assert (two[6, 6] == [255, 0, 0]).all()
    skipped = IM.draw_boxes(img, boxes, np.array([7.0, -1.0]), 2, 4, 1)
    assert (skipped == 0).all()                                                         # classes outside [0, C) are not drawn
    thick = IM.draw_boxes(img, boxes[:1], np.array([1.0]), 1, 4, 3)
    assert (thick[1:4, 1:8] != 0).any(axis=-1).all() and (thick[4, 4] == 0).all()


def test_draw_labels_known_answer():
    """The label of utils.py:233-240 with the 5x7 typeface: rectangle extent, text rows and
    one glyph spelled out pixel by pixel."""
    from oracle import image_np as IM
    img = np.full((40, 60, 3), 9, np.uint8)
    boxes = np.array([[0.5, 0.25, 0.9, 0.75]], np.float32)          # y1 = 20, x1 = 15
    out = IM.draw_boxes(img, boxes, np.array([0.0]), 1, 4, 1, ["T1"], 1)
    red = np.array([255, 0, 0], np.uint8)
    # rectangle: x 15 .. 15 + 12, y 20 - 7 - 3 .. 20, both inclusive
    region = out[10:21, 15:28]
    assert ((region == red).all(-1) | (region == 0).all(-1)).all()
    assert (out[9, 15:28] == 9).all() and (out[10:20, 28] == 9).all() and (out[10:20, 14] == 9).all()
    black = (out == 0).all(-1)
    ys, xs = np.nonzero(black)
    assert ys.min() == 11 and ys.max() == 17                        # seven rows ending at y1 - 3
    # 'T': top bar over five columns and a centre stem
    assert black[11, 15:20].all() and black[12:18, 17].all() and not black[12:18, 15].any()
    # '1' in the next cell (x 21..25): stem in its middle column
    assert black[11:18, 23].all() and not black[:, 20].any() and not black[:, 26:].any()
    # scale 2 doubles every font pixel
    out2 = IM.draw_boxes(np.full((60, 80, 3), 9, np.uint8), np.array([[0.5, 0.25, 0.9, 0.75]], np.float32),
                         np.array([0.0]), 1, 4, 1, ["T"], 2)
    b2 = (out2 == 0).all(-1)
    y1, x1 = 30, 20
    assert b2[y1 - 2 - 14:y1 - 2 - 12, x1:x1 + 10].all() and b2[y1 - 16:y1 - 2, x1 + 4:x1 + 6].all()
    assert IM.label_text("face", np.float32(0.987)) == "face: 0.99"
    # no label when the string is empty; unknown codes draw the hollow box
    assert (IM.draw_boxes(img, boxes, np.array([0.0]), 1, 4, 1, [""], 1) ==
            IM.draw_boxes(img, boxes, np.array([0.0]), 1, 4, 1)).all()
    hollow = (IM.draw_boxes(img, boxes, np.array([0.0]), 1, 4, 1, ["\x01"], 1) == 0).all(-1)
    assert hollow[11:18, 15].all() and hollow[11:18, 19].all() and hollow[11, 15:20].all() and not hollow[13, 16:19].any()


def test_loss_torch_restatement_matches_numpy_and_finite_differences():
    """The torch restatement used for gradients gives the numpy restatement's loss values, and
    its automatic gradient agrees with central finite differences of its own forward."""
    from oracle import loss_torch as LT
    from oracle import loss_np as LN
    from tests_support import loss_case
    B, S, C, scale = 2, 7, 3, 1
    conv, label, bboxes = loss_case(B, S, C, scale, seed=21, n_gt=5)
    target = np.concatenate([label.reshape(B, -1), bboxes.reshape(B, -1)], axis=1)
    out = conv.reshape(B, S, S, -1)
    (g, c, p), grad = LT.loss_and_grad(target, out, scale, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)
    want = LN.loss_at_scale(target, out, scale, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)[1]
    np.testing.assert_allclose([g, c, p], np.array(want, np.float64), rtol=2e-5)
    assert grad.shape == out.shape and np.isfinite(grad).all() and np.abs(grad).max() > 0
    rng = np.random.default_rng(0)
    flat = out.astype(np.float64).reshape(-1)
    resp = np.argwhere(label[..., 4].reshape(-1) > 0).reshape(-1)
    D = 5 + C
    picks = list(rng.integers(0, flat.size, 12)) + [int(resp[0]) * D + k for k in range(D)]
    for i in picks:
        h = 1e-5
        vals = []
        for sgn in (1, -1):
            pert = flat.copy()
            pert[i] += sgn * h
            (g2, c2, p2), _ = LT.loss_and_grad(target, pert.reshape(out.shape), scale, C, P.STRIDES, 0.5, P.XYSCALE, P.ANCHORS)
            vals.append(g2 + c2 + p2)
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - grad.reshape(-1)[i]) <= 1e-5 + 1e-4 * abs(fd), (i, fd, grad.reshape(-1)[i])


def test_labels_oracle_skips_boxes_without_a_cell():
    """Build-defined: a box centred outside the image, or of a class outside [0, C), leaves no
    label row and takes no slot of the ground-truth ring (the reference raises or wraps)."""
    from oracle import labels_np as LB
    from oracle import postproc_np as P
    ag = LB.scale_anchors(P.ANCHORS, P.STRIDES, False)
    inside = np.array([[100.0, 120.0, 180.0, 200.0, 1.0]])
    want = LB.preprocess_true_boxes(inside, 416, 3, P.STRIDES, ag)
    mixed = np.array([[430.0, 10.0, 470.0, 60.0, 0.0], [100.0, 120.0, 180.0, 200.0, 1.0], [-90.0, 5.0, -10.0, 40.0, 2.0],
                      [10.0, 10.0, 50.0, 50.0, 3.0], [10.0, 10.0, 50.0, 50.0, -1.0]])
    got = LB.preprocess_true_boxes(mixed, 416, 3, P.STRIDES, ag)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    assert sum(float(np.abs(w).sum()) for w in want) > 0


def test_l2_penalty_of_the_kernels():
    """Keras l2(c): c * sum(w^2) per kernel (layers.py:61), so each kernel's gradient gains 2 c w."""
    import torch
    from oracle import loss_torch as LT
    ks = [torch.randn(3, 3, 4, 5, generator=torch.Generator().manual_seed(i)).requires_grad_(True) for i in range(3)]
    pen = LT.l2_penalty(ks)
    assert abs(float(pen) - 0.0005 * sum(float((k.double() ** 2).sum()) for k in ks)) < 1e-12
    pen.backward()
    for k in ks:
        assert torch.allclose(k.grad, 0.001 * k.detach(), rtol=1e-6, atol=0)
