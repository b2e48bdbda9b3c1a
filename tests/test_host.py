"""CPU tests of the host side: the C-ABI library loads and exports every symbol the
header declares, the Python modules mirror the reference's interface, the product path
refuses to run without a GPU, and the multi-GPU shard/gather logic (gloo, 2 ranks)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ymx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ymx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from yolo_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(_lib.SIGNATURES) == declared          # the binding covers the header exactly
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().ymx_version() == 100
    assert ctypes.sizeof(_lib.ymx_op) == 17 * 4 + 4 + 6 * 8      # 17 ints, padding, 6 pointers


def test_argument_validation_without_gpu():
    """Entry points validate before touching the device, so these run on a CPU-only box."""
    from yolo_amd import _lib
    L = _lib.lib()
    assert L.ymx_packed_weight_bytes(3, 512, 1024, _lib.BF16) == 1024 * 4608 * 2
    assert L.ymx_packed_weight_bytes(1, 1024, 255, _lib.BF16) == 256 * 1024 * 2
    assert L.ymx_packed_weight_bytes(3, 32, 64, _lib.F32) == 128 * 320 * 4      # K 288 -> 320
    assert L.ymx_packed_weight_bytes(3, 3, 32, _lib.BF16) == 27 * 32 * 4        # direct path keeps HWIO fp32
    assert L.ymx_nms_workspace_bytes(2, 80, 32) == 2 * 80 * 32 * 8 + 640
    rc = L.ymx_conv2d_fused(None, None, None, None, None, None, 1, 8, 8, 8, 8, 3, 1, 0, 0, None)
    assert rc == -1 and b"null" in L.ymx_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    one = ctypes.c_void_p(16)
    rc = L.ymx_conv2d_fused(one, one, None, None, None, one, 1, 8, 8, 8, 8, 4, 1, 0, 0, None)
    assert rc == -2 and b"kernel size" in L.ymx_last_error()
    rc = L.ymx_nms_combined(one, one, None, None, 1, 128, 4, 100, 32, 0.5, 0.1, one, one, one, one, one, one, 1 << 20, None)
    assert rc == -2
    rc = L.ymx_nms_combined(one, one, None, None, 1, 128, 4, 32, 32, 0.5, 0.1, one, one, one, one, one, one, 8, None)
    assert rc == -3
    with pytest.raises(RuntimeError):
        _lib.check(rc)
    # a program op with a negative tile threshold is rejected, with the op's index in the message
    assert ctypes.sizeof(_lib.ymx_op) == 18 * 4 + 6 * 8
    op = _lib.ymx_op()
    op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, 1, 8, 8, 8, 8, 3, 1
    op.dtype = op.x_dtype = op.out_dtype = _lib.BF16
    op.x = op.w = op.y = 16
    op.min_tiles = -1
    prog = (_lib.ymx_op * 1)(op)
    rc = L.ymx_run_program(prog, 1, None)
    assert rc == -1 and b"op 0" in L.ymx_last_error() and b"min_tiles" in L.ymx_last_error()
    prog[0].kind = 7
    assert L.ymx_run_program(prog, 1, None) == -1 and b"unknown kind" in L.ymx_last_error()
    # the backward-pass entry points
    assert L.ymx_conv_wgrad(None, 0, one, 0, one, 1, 8, 8, 8, 8, 3, 1, None, 0, None) == -1
    assert L.ymx_conv_wgrad(one, 0, one, 0, one, 1, 8, 8, 12, 8, 3, 1, None, 0, None) == -2 and b"multiples of 8" in L.ymx_last_error()
    assert L.ymx_conv_wgrad(one, 0, one, 0, one, 1, 8, 8, 8, 8, 5, 1, None, 0, None) == -2
    nb = L.ymx_conv_wgrad_workspace_bytes(64, 52, 52, 512, 1024, 3, 1)
    assert nb > 0 and nb % (9 * 512 * 1024 * 4) == 0                     # whole partial gradients, one per pixel slice
    assert L.ymx_conv_wgrad(one, 0, one, 0, one, 64, 52, 52, 512, 1024, 3, 1, one, 16, None) == -3
    assert L.ymx_pack_weights_dgrad(one, one, 3, 64, 255, 248, _lib.BF16, None) == -1 and b"dz_channels" in L.ymx_last_error()
    assert L.ymx_pack_weights_dgrad(None, one, 3, 64, 256, 256, _lib.BF16, None) == -1
    assert L.ymx_set_option(b"wgrad_variant", 3) == -1
    assert L.ymx_set_option(b"wgrad_variant", 2) == 0 and L.ymx_set_option(b"wgrad_variant", 0) == 0


def test_label_encoding_and_labeled_draw_validation():
    """Host half of the captions of draw_bbox (utils.py:234): string formatting, the
    zero-padded byte table the library takes, and the entry point's argument checks."""
    import numpy as np
    from yolo_amd import _lib, utils as U
    assert U.label_text("face", np.float32(0.125)) == "face: 0.12" and U.label_text("7", 1.0) == "7: 1.00"
    tab = U.encode_labels([["ab", None, "\u20ac\xe9"], []], 4, label_bytes=3)
    assert tab.shape == (2, 4, 3) and tab.dtype == np.uint8
    assert tab[0, 0].tolist() == [97, 98, 0] and tab[0, 1].tolist() == [0, 0, 0] and tab[0, 2].tolist() == [127, 0xE9, 0]
    assert not tab[1].any()
    assert U.encode_labels([["abcdef"]], 1, label_bytes=4)[0, 0].tolist() == [97, 98, 99, 100]      # cut, no terminator
    with pytest.raises(ValueError, match="detection slots"):
        U.encode_labels([["a", "b"]], 1)
    L = _lib.lib()
    one = ctypes.c_void_p(16)
    assert L.ymx_draw_boxes_labeled(one, 1, 8, 8, one, one, one, 4, one, 3, 1, None, 16, 2, None) == -1 and b"null" in L.ymx_last_error()
    assert L.ymx_draw_boxes_labeled(one, 1, 8, 8, one, one, one, 4, one, 3, 1, one, 0, 2, None) == -1 and b"label_bytes" in L.ymx_last_error()
    assert L.ymx_draw_boxes_labeled(one, 1, 8, 8, one, one, one, 4, one, 3, 1, one, 257, 2, None) == -1
    assert L.ymx_draw_boxes_labeled(one, 1, 8, 8, one, one, one, 4, one, 3, 1, one, 16, 0, None) == -1 and b"text_scale" in L.ymx_last_error()
    assert L.ymx_draw_boxes_labeled(one, 1, 8, 8, one, one, one, 4, one, 3, 1, one, 16, 17, None) == -1
    assert L.ymx_draw_boxes_labeled(one, 0, 8, 8, one, one, one, 4, one, 3, 1, one, 16, 2, None) == -1


def test_fused_stem_entry_point_validates_without_gpu():
    from yolo_amd import _lib
    L = _lib.lib()
    one = ctypes.c_void_p(256)
    args = lambda **kw: [kw.get("x", one), one, None, one, kw.get("w1", one), None, one, kw.get("y", one),
                         kw.get("B", 2), kw.get("H", 64), kw.get("W", 64), kw.get("act", 2), kw.get("y_ld", 0), None]
    assert L.ymx_conv_stem_fused(*args(x=None)) == -1 and b"null" in L.ymx_last_error()
    assert L.ymx_conv_stem_fused(*args(H=1)) == -1
    assert L.ymx_conv_stem_fused(*args(y_ld=60)) == -1 and b"y_ld" in L.ymx_last_error()
    assert L.ymx_conv_stem_fused(*args(y_ld=66)) == -1
    assert L.ymx_conv_stem_fused(*args(w1=ctypes.c_void_p(264))) == -1 and b"aligned" in L.ymx_last_error()
    assert L.ymx_conv_stem_fused(*args(act=0)) == -2 and b"activation" in L.ymx_last_error()
    assert L.ymx_conv_stem_fused(*args(B=64, H=2048, W=2048)) == -2          # the image batch must stay under 2 GB


def test_fused_resblock_entry_point_validates_without_gpu():
    from yolo_amd import _lib
    L = _lib.lib()
    one = ctypes.c_void_p(256)
    args = lambda **kw: [kw.get("x", one), kw.get("w1", one), None, one, one, None, one, kw.get("y", ctypes.c_void_p(512)),
                         kw.get("B", 2), kw.get("H", 64), kw.get("W", 64), kw.get("C", 64), kw.get("act", 2),
                         kw.get("x_ld", 0), kw.get("y_ld", 0), None]
    assert L.ymx_conv_resblock_fused(*args(x=None)) == -1 and b"null" in L.ymx_last_error()
    assert L.ymx_conv_resblock_fused(*args(H=0)) == -1
    assert L.ymx_conv_resblock_fused(*args(C=96)) == -2 and b"channels" in L.ymx_last_error()
    assert L.ymx_conv_resblock_fused(*args(C=128, x_ld=64)) == -1 and b"x_ld" in L.ymx_last_error()
    assert L.ymx_conv_resblock_fused(*args(y_ld=68)) == -1
    assert L.ymx_conv_resblock_fused(*args(w1=ctypes.c_void_p(264))) == -1 and b"aligned" in L.ymx_last_error()
    assert L.ymx_conv_resblock_fused(*args(y=one)) == -1 and b"must not be x" in L.ymx_last_error()
    assert L.ymx_conv_resblock_fused(*args(act=0)) == -2 and b"activation" in L.ymx_last_error()
    assert L.ymx_conv_resblock_fused(*args(B=64, H=2048, W=2048)) == -2          # x must stay under 2 GB


def test_wide_tile_plan_labels_are_consistent():
    """The tile plan of the wide kernels, asked of the library for a sweep of batch sizes: a mixed
    launch is only chosen for launches of two rounds and more, and forcing a pure form always works."""
    from yolo_amd import _lib
    L = _lib.lib()
    try:
        for B in range(1, 130, 3):
            for hw, cin, cout, k in ((52, 512, 1024, 3), (52, 128, 256, 3), (26, 512, 1024, 3), (52, 1024, 512, 1), (13, 512, 1024, 3)):
                op = _conv_op(B, hw, hw, cin, cout, k, min_tiles=1)
                tiles = -(-B * hw * hw // 256) * -(-cout // 256)
                lab = _label(op)
                assert lab in ("halo", "halo192", "halo_mixed", "dma4", "dma4_192", "dma4_mixed"), (B, hw, lab)
                if lab.endswith("_mixed"):
                    assert tiles > 256, (B, hw, cin, cout, tiles)
                opt = b"halo_pixels" if k == 3 else b"dma4_pixels"
                for px, want in ((256, ""), (192, "192")):
                    _lib.check(L.ymx_set_option(opt, px))
                    assert _label(op) == ("halo" if k == 3 else "dma4") + ("_" if (want and k == 1) else "") + want, (B, hw, px)
                _lib.check(L.ymx_set_option(opt, 0))
    finally:
        _lib.check(L.ymx_set_option(b"halo_pixels", 0))
        _lib.check(L.ymx_set_option(b"dma4_pixels", 0))


def test_module_interface_matches_reference():
    import yolo_amd
    from yolo_amd import layers, backbones, detectors
    assert yolo_amd.config["strides"] == [8, 16, 32] and yolo_amd.config["xyscale"] == [1.2, 1.1, 1.05]
    assert len(yolo_amd.config["anchors"]) == 18 and yolo_amd.config["anchor_per_scale"] == 3
    for name in ("decode", "decode_train", "decode_predict", "filter_bboxes", "bbox_nms", "bbox_iou", "bbox_giou"):
        assert callable(getattr(yolo_amd, name))
    for name in ("BatchNormalization", "Mish", "RouteGroup", "Upsample", "YOLOConv", "ResidialBlock"):
        assert hasattr(layers, name)
    with pytest.raises(ValueError, match="Activation should either be a function"):
        layers.YOLOConv((3, 3, 8), activation="relu")
    c = layers.YOLOConv((3, 3, 8), in_channels=4)
    assert c.kernel.shape == (3, 3, 4, 8) and c.bias.shape == (8,) and c.act_code == 2 and c.bn is None
    assert abs(float(c.kernel.detach().std()) - 0.01) < 0.003 and float(c.bias.detach().abs().max()) == 0
    c = layers.YOLOConv((1, 1, 8), batchNormalization=True, activation="leaky", in_channels=4)
    assert not hasattr(c, "bias") and c.bn is not None and c.act_code == 1
    assert layers.YOLOConv((3, 3, 8), downsample=True).output_hw(416, 416) == (208, 208)

    bb = backbones.Darknet53()
    hd = detectors.YoloV3(80)
    assert len(bb.conv_layers()) == 52 and len(hd.conv_layers()) == 21
    assert [len(s) for s in (bb.layer_stack_1, bb.layer_stack_2, bb.layer_stack_3)] == [15, 9, 5]
    n = sum(p.numel() for p in bb.parameters()) + sum(p.numel() for p in hd.parameters())
    assert n == 88419869
    assert [c.name for c in hd.conv_layers() if c.name] == ["lbbox", "mbbox", "sbbox"]
    assert all(c.act_code == 0 and c.filters == 255 for c in hd.conv_layers() if c.name)
    from oracle import net_torch as N
    got = [(c.kernel_size, c.in_channels, c.filters, c.downsample) for c in bb.conv_layers() + hd.conv_layers()]
    want = [(k, ci, co, d) for k, ci, co, d, _ in N.darknet53_specs() + N.yolov3_specs(80)]
    assert got == want


def test_cspdarknet53_structure():
    """The repaired CSPDarknet53: attribute names of the reference (backbones.py:66-173), 78
    convolutions whose shapes follow the data flow of backbones.py:175-264."""
    from yolo_amd import backbones
    from oracle import net_torch as N
    m = backbones.CSPDarknet53(seed=0)
    names = [n for n, _ in N.cspdarknet53_stacks()]
    assert all(hasattr(m, n) for n in names) and len(names) == 19
    got = [(c.kernel_size, c.in_channels, c.filters, c.downsample) for c in m.conv_layers()]
    want = [(k, ci, co, d) for k, ci, co, d, _ in N.cspdarknet53_specs()]
    assert got == want and len(got) == 78
    assert [len(getattr(m, n)) for n in ("layer_stack_2_skip", "layer_stack_6", "layer_stack_12", "layer_stack_13")] == [3, 10, 6, 4]
    assert m.layer_stack_1[1].filters == 64 and m.layer_stack_14[0].in_channels == 2048


def test_no_cpu_fallback():
    from yolo_amd import layers, utils
    x = torch.zeros(1, 8, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layers.YOLOConv((3, 3, 8), in_channels=8)(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layers.Mish()(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        utils.bbox_iou(torch.zeros(2, 4), torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        utils.bbox_nms([torch.zeros(1, 2, 2, 21), torch.zeros(1, 1, 1, 21)], 16)
    assert layers.RouteGroup(2, 1)(torch.arange(8.).reshape(1, 1, 1, 8)).flatten().tolist() == [4, 5, 6, 7]


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "yolo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text, f


def test_shard_range_and_pack_roundtrip():
    from yolo_amd import dist as D
    from yolo_amd.utils import Detections
    for gb, world in ((256, 8), (10, 4), (3, 8), (64, 1)):
        spans = [D.shard_range(gb, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == gb
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    g = torch.Generator().manual_seed(0)
    det = Detections(torch.rand(3, 5, 4, generator=g), torch.rand(3, 5, generator=g),
                     torch.randint(0, 80, (3, 5), generator=g).float(),
                     torch.tensor([5, 0, 2], dtype=torch.int32),
                     torch.randint(-1, 10647, (3, 5), generator=g, dtype=torch.int32))
    rec = D.pack_detections(det, 4)
    assert rec.shape == (4, 36)
    back = D.unpack_detections(rec[:3], 5)
    for a, b in zip(tuple(det) + (det.keep_idx,), tuple(back) + (back.keep_idx,)):
        assert torch.equal(a, b)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from yolo_amd import dist as D
from yolo_amd.utils import Detections
WORLD = %(world)d
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=WORLD)
rank = dist.get_rank()
T = 4
ok = True
for GB in (2 * WORLD + 3, 4 * WORLD, 1):  # uneven shards, even shards, ranks with no image
    g = torch.Generator().manual_seed(7 + GB)
    full = Detections(torch.rand(GB, T, 4, generator=g), torch.rand(GB, T, generator=g),
                      torch.randint(0, 80, (GB, T), generator=g).float(),
                      torch.randint(0, T + 1, (GB,), generator=g, dtype=torch.int32),
                      torch.randint(-1, 999, (GB, T), generator=g, dtype=torch.int32))
    s, c = D.shard_range(GB, rank, WORLD)
    mine = Detections(*[t[s:s + c].clone() for t in full], full.keep_idx[s:s + c].clone())
    got = D.all_gather_detections(mine, GB)
    ok = ok and all(torch.equal(a, b) for a, b in zip(tuple(full) + (full.keep_idx,), tuple(got) + (got.keep_idx,)))
# data-parallel gradient averaging: several buckets, two dtypes, gradients stay views of the buckets
ps = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300, 7, 64)] + [torch.nn.Parameter(torch.zeros(3, 3, dtype=torch.float64))]
gb = D.GradientBuckets(ps, bucket_bytes=1024)
ok = ok and len(gb.buckets) >= 3
for step in range(2):
    gb.begin()
    for i, p in enumerate(ps):
        p.grad.add_(torch.full_like(p, float((rank + 1) * (i + 1) + step)))      # what a backward pass would accumulate
    gb.reduce().wait()
    for i, p in enumerate(ps):
        want = sum((r + 1) * (i + 1) + step for r in range(WORLD)) / WORLD
        ok = ok and torch.allclose(p.grad, torch.full_like(p, want)) and p.grad.data_ptr() >= min(b.data_ptr() for b in gb.buckets)
ok = ok and all(p.grad.data_ptr() %% 256 == b.data_ptr() %% 256 for p in ps for b in [gb.buckets[gb._where[id(p)]]])
# gradients reported ready in an order that differs from the buckets': collectives still start in bucket order, and a
# bucket starts as soon as it and every bucket before it are complete (not only at wait())
gb.begin()
for i, p in enumerate(ps):
    p.grad.add_(torch.full_like(p, float(rank + i)))
started = []
for p in ps:                                   # forward parameter order = reverse of the bucket order
    gb.ready(p)
    started.append(len(gb.launch_order))
ok = ok and gb.launch_order == list(range(len(gb.buckets))) and started[0] == 0 and started[-1] == len(gb.buckets)
gb.wait()
for i, p in enumerate(ps):
    ok = ok and torch.allclose(p.grad, torch.full_like(p, sum(r + i for r in range(WORLD)) / WORLD))
gb.remove_hooks()
# driven by autograd: a small network whose parameters receive gradients through the post-accumulate hooks; every bucket's
# collective has started by the time backward() returns, in bucket order, and the result is the mean of the ranks' gradients
torch.manual_seed(3)
net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40), torch.nn.Tanh(), torch.nn.Linear(40, 2))
nb = D.GradientBuckets(list(net.parameters()), bucket_bytes=2048)
xs = torch.randn(WORLD, 5, 6, generator=torch.Generator().manual_seed(11))
for step in range(2):
    nb.begin()
    net(xs[rank]).square().sum().backward()
    ok = ok and len(nb.buckets) >= 3 and nb.launch_order == list(range(len(nb.buckets)))      # all started inside backward()
    nb.wait()
    got = [p.grad.clone() for p in net.parameters()]
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(WORLD):
        gs = torch.autograd.grad(net(xs[r]).square().sum(), list(net.parameters()))
        for a, g_ in zip(ref, gs):
            a.add_(g_ / WORLD)
    ok = ok and all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(got, ref))
# rehearse=True: the collectives are issued in a group of one rank too
rp = torch.nn.Parameter(torch.zeros(10))
rb = D.GradientBuckets([rp], hooks=False, rehearse=True)
rb.begin(); rp.grad.add_(2.0); rb.reduce()
ok = ok and len(rb._handles) == 1
rb.wait()
ok = ok and torch.allclose(rp.grad, torch.full_like(rp, 2.0))
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_all_gather_gloo(tmp_path, world):
    """Shard -> pack -> all-gather -> unpack over gloo at every world size the benchmark is
    launched with: the gathered detections equal the unsharded ones bit for bit.  Then the
    bucketed gradient all-reduce of a training step: averages over the ranks, in place; the
    collectives start in bucket order as gradients are reported ready, from explicit ready()
    calls in a scrambled order and from autograd's hooks during backward()."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "port": port, "world": world})
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env) for r in range(world)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0] * world


def _conv_op(B, H, W, cin, cout, k, stride=1, min_tiles=0):
    from yolo_amd import _lib
    op = _lib.ymx_op()
    op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, B, H, W, cin, cout, k, stride
    op.dtype = op.x_dtype = op.out_dtype = _lib.BF16
    op.x = op.w = op.y = 256          # never dereferenced: the label call validates and selects only
    op.min_tiles = min_tiles
    return op


def _label(op):
    from yolo_amd import _lib
    r = _lib.lib().ymx_conv_kernel_label(ctypes.byref(op))
    return None if r is None else r.decode()


def test_kernel_selection_labels():
    """Which kernel family a layer is routed to, asked of the library (no GPU needed)."""
    assert _label(_conv_op(64, 52, 52, 512, 1024, 3)) == "halo_mixed"   # 2 704 tiles = 10.56 rounds: 2 560 tiles of 256 pixels + 192 short ones
    assert _label(_conv_op(48, 52, 52, 128, 256, 3)) == "halo"          # 507 tiles: two rounds either way
    assert _label(_conv_op(64, 76, 76, 512, 1024, 3)) == "halo192"    # rows too long for two slabs of 256-pixel tiles
    assert _label(_conv_op(32, 91, 91, 512, 1024, 3)) == "halo192"
    assert _label(_conv_op(32, 92, 92, 512, 1024, 3)).startswith("dma4")   # rows too long for two resident slabs of any tile
    assert _label(_conv_op(64, 13, 13, 512, 1024, 3)) == "halo192"    # 172 tiles of 256 pixels would idle a third of the CUs
    assert _label(_conv_op(64, 26, 26, 256, 512, 3)) == "halo192"     # 338 tiles: two rounds either way, shorter ones
    assert _label(_conv_op(64, 26, 26, 512, 1024, 3)) == "halo_mixed"  # 676 tiles: two rounds of 256-pixel tiles and one of 220 192-pixel tiles (2.72 against 3 or 4 x 0.72)
    assert _label(_conv_op(64, 52, 52, 128, 256, 3)) == "halo_mixed"   # 676 tiles again (one channel tile): 512 + 219 short ones
    assert _label(_conv_op(48, 26, 26, 512, 1024, 3)) == "halo"       # 2 rounds against 3
    assert _label(_conv_op(64, 52, 52, 1024, 512, 1)) == "dma4_mixed" # 1 352 tiles = 5.28 rounds: 5 rounds of 256-pixel tiles + 96 short ones
    assert _label(_conv_op(48, 52, 52, 1024, 512, 1)) == "dma4"       # 1 014 tiles: 4 rounds
    assert _label(_conv_op(64, 26, 26, 512, 1024, 3, stride=2)) == "dma4_192"   # 172 tiles of 256 pixels: a third of the CUs idle
    assert _label(_conv_op(64, 26, 26, 512, 256, 1)) == "dma4_192"    # 169 tiles against 226 shorter ones
    assert _label(_conv_op(64, 104, 104, 128, 256, 3, stride=2)) == "dma4_mixed"  # 676 tiles: 512 + 219 short ones
    assert _label(_conv_op(1, 13, 13, 512, 1024, 3)) == "igemm"       # too few tiles for the large-tile kernels
    assert _label(_conv_op(1, 13, 13, 512, 1024, 3, min_tiles=1)) == "halo192"
    assert _label(_conv_op(64, 208, 208, 32, 64, 3)) == "slab2"
    assert _label(_conv_op(64, 208, 208, 64, 32, 1)) == "dma2"
    # the input-slab kernel is chosen by its real LDS need: 32-channel rows up to 343 pixels,
    # 64-channel rows up to 123; longer rows fall through to the streaming kernel
    assert _label(_conv_op(16, 343, 343, 32, 64, 3)) == "slab2"
    for w in (344, 352, 368, 384, 400):                                 # image sides 688 ... 800
        assert _label(_conv_op(16, w, w, 32, 64, 3)) == "dma2", w
    assert _label(_conv_op(16, 123, 123, 64, 128, 3)) == "slab2"
    assert _label(_conv_op(16, 124, 124, 64, 128, 3)) == "dma2"
    from yolo_amd import _lib
    bad = _conv_op(1, 8, 8, 8, 8, 4)
    assert _label(bad) is None and b"kernel size" in _lib.lib().ymx_last_error()
    first = _conv_op(2, 64, 64, 3, 32, 3)
    first.x_dtype = _lib.F32
    assert _label(first) == "first_rows"


def test_options_are_per_thread():
    """ymx_set_option changes the calling thread only: two threads holding different options
    select different kernels for the same op, repeatedly and concurrently, and a fresh thread
    sees the defaults."""
    import threading
    from yolo_amd import _lib
    L = _lib.lib()
    op_args = (64, 52, 52, 512, 1024, 3)
    want = {1: "dma", 4: "dma4_mixed", 5: "halo_mixed"}
    errors = []
    start = threading.Barrier(3)

    def worker(variant):
        try:
            _lib.check(L.ymx_set_option(b"dma_variant", variant))
            start.wait()
            for _ in range(2000):
                got = _label(_conv_op(*op_args))
                if got != want[variant]:
                    errors.append((variant, got))
                    return
        except Exception as e:          # noqa: BLE001 - reported through the list
            errors.append((variant, repr(e)))

    threads = [threading.Thread(target=worker, args=(v,)) for v in (1, 4, 5)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    # this thread never set anything: defaults, whatever the workers did
    assert _label(_conv_op(*op_args)) == "halo_mixed"
    seen = []
    t = threading.Thread(target=lambda: seen.append(_label(_conv_op(*op_args))))
    t.start(); t.join()
    assert seen == ["halo_mixed"]
    # min_tiles too: a thread that disables the large-tile kernels does not affect this one
    def disable():
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 1 << 40))
        seen.append(_label(_conv_op(*op_args)))
    t = threading.Thread(target=disable)
    t.start(); t.join()
    assert seen[-1] == "igemm" and _label(_conv_op(*op_args)) == "halo_mixed"


# ---- TFRecord shards (dataset.py:56-126) ------------------------------------------------------------
def _example_classes():
    """tf.train.Example / Features / Feature and TensorProto built from their published .proto
    definitions with the protobuf runtime (an encoder independent of yolo_amd.tfrecord)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="ymx_example_test.proto", package="ymxt", syntax="proto3")

    def msg(name, fields, nested=()):
        m = fd.message_type.add(name=name)
        for fname, num, typ, label, tname, oneof in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
            if oneof is not None:
                f.oneof_index = oneof
        return m

    O, R = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("BytesList", [("value", 1, F.TYPE_BYTES, R, None, None)])
    msg("FloatList", [("value", 1, F.TYPE_FLOAT, R, None, None)])
    msg("Int64List", [("value", 1, F.TYPE_INT64, R, None, None)])
    m = msg("Feature", [("bytes_list", 1, F.TYPE_MESSAGE, O, ".ymxt.BytesList", 0), ("float_list", 2, F.TYPE_MESSAGE, O, ".ymxt.FloatList", 0),
                        ("int64_list", 3, F.TYPE_MESSAGE, O, ".ymxt.Int64List", 0)])
    m.oneof_decl.add(name="kind")
    m = msg("Features", [("feature", 1, F.TYPE_MESSAGE, R, ".ymxt.Features.FeatureEntry", None)])
    e = m.nested_type.add(name="FeatureEntry")
    e.field.add(name="key", number=1, type=F.TYPE_STRING, label=O)
    e.field.add(name="value", number=2, type=F.TYPE_MESSAGE, label=O, type_name=".ymxt.Feature")
    e.options.map_entry = True
    msg("Example", [("features", 1, F.TYPE_MESSAGE, O, ".ymxt.Features", None)])
    msg("Dim", [("size", 1, F.TYPE_INT64, O, None, None), ("name", 2, F.TYPE_STRING, O, None, None)])
    msg("TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, R, ".ymxt.Dim", None), ("unknown_rank", 3, F.TYPE_BOOL, O, None, None)])
    msg("TensorProto", [("dtype", 1, F.TYPE_INT32, O, None, None), ("tensor_shape", 2, F.TYPE_MESSAGE, O, ".ymxt.TensorShapeProto", None),
                        ("version_number", 3, F.TYPE_INT32, O, None, None), ("tensor_content", 4, F.TYPE_BYTES, O, None, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("ymxt." + n))
    return get("Example"), get("TensorProto")


def test_tfrecord_format():
    from yolo_amd import tfrecord as T
    # CRC-32C check values of RFC 3720 B.4 and the common "123456789" vector
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # the library's eight-bytes-per-step routine against the bytewise definition: every length
    # around its block size, an unaligned start, and a buffer fed in two pieces
    from yolo_amd import _lib
    blob = bytes(np.random.default_rng(0).integers(0, 256, 5000, dtype=np.uint8))
    for n in list(range(0, 40)) + [63, 64, 65, 1000, 4999, 5000]:
        assert int(_lib.lib().ymx_crc32c(blob[:n], n, 0)) == T._crc32c_bytewise(blob[:n]), n
        assert T.crc32c(blob[3:3 + n]) == T._crc32c_bytewise(blob[3:3 + n])
    part = int(_lib.lib().ymx_crc32c(blob[:1234], 1234, 0))
    assert int(_lib.lib().ymx_crc32c(blob[1234:], len(blob) - 1234, part)) == T._crc32c_bytewise(blob)
    c = T.crc32c(b"abc")
    assert T.masked_crc32c(b"abc") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    # wire encoding by hand: Example{features{feature{"a": float_list [1.0]}}}
    assert T.example({"a": T.floats_feature([1.0])}) == bytes.fromhex("0a0f0a0d0a0161120812060a040000803f")
    # ... and against the protobuf runtime on the messages' published definitions
    Example, TensorProto = _example_classes()
    rng = np.random.default_rng(0)
    img = rng.random((4, 4, 3)).astype(np.float16)
    tp = TensorProto(dtype=19, tensor_content=img.tobytes())
    for d in img.shape:
        tp.tensor_shape.dim.add(size=d)
    assert T.serialize_tensor(img) == tp.SerializeToString(deterministic=True)
    np.testing.assert_array_equal(T.parse_tensor(T.serialize_tensor(img)), img)
    for arr in (np.arange(6, dtype=np.int64).reshape(2, 3), np.float32(3.5), np.zeros((0, 2), np.float32), np.arange(5, dtype=np.uint8)):
        back = T.parse_tensor(T.serialize_tensor(arr))
        assert back.dtype == arr.dtype and back.shape == np.shape(arr) and (back == arr).all()
    vals = rng.random(7).astype(np.float16)
    ex = Example()
    ex.features.feature["image"].bytes_list.value.append(T.serialize_tensor(img))
    ex.features.feature["small_targets"].float_list.value.extend([float(v) for v in vals])
    ex.features.feature["count"].int64_list.value.extend([3, -1, 1 << 40])
    mine = T.example({"small_targets": T.floats_feature(vals), "image": T.bytes_feature(T.serialize_tensor(img)),
                      "count": T.int64_feature([3, -1, 1 << 40])})
    assert mine == ex.SerializeToString(deterministic=True)
    parsed = T.parse_example(mine)
    np.testing.assert_array_equal(parsed["small_targets"], vals.astype(np.float32))
    np.testing.assert_array_equal(parsed["count"], [3, -1, 1 << 40])
    np.testing.assert_array_equal(T.parse_tensor(parsed["image"]), img)


def test_to_tfrecord_shards(tmp_path):
    from yolo_amd import tfrecord as T
    rng = np.random.default_rng(1)
    examples = []
    for _ in range(7):
        image = rng.random((1, 8, 8, 3)).astype(np.float16)
        targets = tuple((rng.random((1, n)) * (rng.random((1, n)) < 0.2)).astype(np.float16) for n in (40, 20, 10))
        examples.append((image, targets))
    paths = T.to_tfrecord(examples, str(tmp_path / "train.tfrecord"), 3)
    assert [os.path.basename(p) for p in paths] == ["train.tfrecord.shard%dof3" % k for k in (1, 2, 3)]
    seen = 0
    for p in paths:
        recs = list(T.read_records(p))
        assert len(recs) == 2                                   # 7 // 3 per shard, the seventh example is dropped
        for r in recs:
            image, targets = examples[seen]
            f = T.parse_example(r)
            np.testing.assert_array_equal(T.parse_tensor(f["image"]), image[0])
            for key, t in zip(("small_targets", "medium_targets", "large_targets"), targets):
                np.testing.assert_array_equal(f[key], t[0].astype(np.float32))
            seen += 1
    # a flipped byte is caught by the checksums
    raw = bytearray(open(paths[0], "rb").read())
    raw[20] ^= 1
    bad = tmp_path / "bad"
    bad.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        list(T.read_records(str(bad)))
    # sparse form: the indices of the non-zero targets, as the reference stores them
    sp = T.to_tfrecord(examples[:2], str(tmp_path / "sp"), 1, use_sparse=True)
    f = T.parse_example(next(T.read_records(sp[0])))
    np.testing.assert_array_equal(T.parse_tensor(f["small_targets"]), np.argwhere(examples[0][1][0][0] != 0))
    with pytest.raises(ValueError):
        T.to_tfrecord(examples[:2], str(tmp_path / "x"), 3)


def test_library_code_has_no_packed_fma_of_the_hazardous_form():
    """gfx950 returns a packed fp32 FMA without its addend on lanes 32-63, while another wave's
    MFMAs occupy the SIMD, when the multiplier is a scalar pair and the low result's addend is
    taken from the high register (tools/isa_lint.py says how that was measured).  hipcc forms
    that instruction on its own from scalar code, so the built library's machine code is what is
    checked: every kernel, disassembled."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    from yolo_amd import _lib
    # the matcher itself, on the instruction as hipcc emitted it and on its harmless relatives
    bad = "v_pk_fma_f32 v[14:15], s[12:13], v[16:17], v[12:13] op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]"
    assert isa_lint.hazardous(bad)
    assert isa_lint.hazardous("v_pk_fma_f32 v[2:3], v[4:5], s[6:7], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,0,1]")
    assert not isa_lint.hazardous("v_pk_fma_f32 v[14:15], s[12:13], v[16:17], v[12:13] op_sel_hi:[0,1,1]")
    assert not isa_lint.hazardous("v_pk_fma_f32 v[14:15], v[20:21], v[16:17], v[12:13] op_sel:[0,0,1] op_sel_hi:[0,1,0]")
    assert not isa_lint.hazardous("v_pk_mul_f32 v[12:13], s[92:93], v[12:13] op_sel:[0,1]")
    assert not isa_lint.hazardous("v_fma_f32 v1, s2, v3, v4")
    found, functions = isa_lint.scan(_lib.LIB_PATH)
    assert functions >= 100, functions          # the disassembly really covered the kernels
    assert found == [], found


def test_l2_regulariser_and_bucket_close_on_cpu():
    """training.L2Regulariser is plain tensor arithmetic (runs anywhere): the gradient gains
    2 * c * kernel, biases are left alone; GradientBuckets.close() hands plain gradients back."""
    from yolo_amd import training as T, dist as D
    from yolo_amd.layers import YOLOConv
    net = torch.nn.ModuleList([YOLOConv((3, 3, 8), in_channels=4), YOLOConv((1, 1, 6), in_channels=8)])
    reg = T.L2Regulariser(net, coefficient=0.0005)
    assert len(reg.kernels) == 2
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    reg.add_gradients_()
    for m in net:
        assert torch.allclose(m.kernel.grad, 1.0 + 0.001 * m.kernel.detach())
        assert torch.equal(m.bias.grad, torch.ones_like(m.bias))
    want = 0.0005 * sum(float((m.kernel.detach().double() ** 2).sum()) for m in net)
    assert abs(float(reg.penalty()) - want) < 1e-12 + 1e-6 * want
    params = list(net.parameters())
    gb = D.GradientBuckets(params, bucket_bytes=256)
    assert all(p.grad.data_ptr() == p._grad_bucket[1].data_ptr() for p in params)
    assert T._slot(params[0]) is not None
    params[0].grad = None
    assert T._slot(params[0]) is None            # a caller who went back to plain gradients gets them from autograd
    gb.close()
    assert not any(hasattr(p, "_grad_bucket") for p in params)
    with pytest.raises(Exception):
        T.batch_norm_act(torch.nn.Module(), torch.zeros(1, 2, 2, 8))    # CPU tensors: no fallback


def test_training_entry_points_validate_without_gpu():
    """The training-side entry points reject bad shapes before any launch (CPU-only box)."""
    from yolo_amd import _lib
    L = _lib.lib()
    one = ctypes.c_void_p(256)
    # depth-to-space input gradient: dz channels and Cin in multiples of 8, output map 2 OH (+1) on a side
    assert L.ymx_conv2d_dgrad_s2(one, one, one, 1, 4, 4, 12, 8, 8, 8, None) == -1 and b"multiples of 8" in L.ymx_last_error()
    assert L.ymx_conv2d_dgrad_s2(one, one, one, 1, 4, 4, 16, 12, 8, 8, None) == -1
    assert L.ymx_conv2d_dgrad_s2(one, one, one, 1, 4, 4, 16, 8, 12, 8, None) == -1 and b"does not match" in L.ymx_last_error()
    # forward with two outputs: bf16 with an activation
    assert L.ymx_conv2d_fused_z(one, one, None, None, None, one, one, 1, 8, 8, 8, 8, 3, 1, 0, _lib.BF16, None) == -1
    assert L.ymx_conv2d_fused_z(one, one, None, None, None, one, None, 1, 8, 8, 8, 8, 3, 1, 2, _lib.BF16, None) == -1
    # pair pack: stride 2 only for 3x3 bf16; MFMA channel counts
    assert L.ymx_pack_weights_pair(one, one, one, 1, 8, 8, 8, 2, _lib.BF16, None) == -1
    assert L.ymx_pack_weights_pair(one, one, one, 3, 3, 8, 8, 1, _lib.BF16, None) == -2
    assert L.ymx_pack_weights_dgrad_s2(one, one, 12, 8, 8, _lib.BF16, None) == -1
    # first-layer weight gradient: at most 3 input channels, 64 output channels in multiples of 8
    assert L.ymx_conv_wgrad_first_workspace_bytes(2, 64, 64, 3, 32) == min(2 * 64, 768) * 4 * 27 * 32 * 4
    assert L.ymx_conv_wgrad_first_workspace_bytes(2, 64, 64, 8, 32) == 0
    assert L.ymx_conv_wgrad_first_workspace_bytes(2, 64, 64, 3, 36) == 0
    assert L.ymx_conv_wgrad_first(one, _lib.F32, one, 0, one, 2, 64, 64, 4, 32, one, 1 << 30, None) == -2
    assert L.ymx_conv_wgrad_first(one, _lib.F32, one, 0, one, 2, 64, 64, 3, 32, one, 16, None) == -3
    # batch-norm pieces and the SPP pair: null pointers, sizes, workspace
    assert L.ymx_bn_workspace_bytes(1000, 64) == 2 * 2 * 64 * 4
    assert L.ymx_bn_stats(one, 1000, 64, _lib.BF16, one, one, 8, None) == -1 and b"workspace" in L.ymx_last_error()
    assert L.ymx_bn_act_forward(one, None, one, None, one, 10, 8, 2, _lib.BF16, None) == -1
    assert L.ymx_bn_backward_apply(one, one, one, one, one, one, 0, 8, _lib.BF16, None) == -1
    assert L.ymx_spp_concat_train(one, one, None, 1, 13, 13, 8, _lib.BF16, None) == -1
    assert L.ymx_spp_concat_backward(one, one, one, 1, 13, 13, 8, 7, None) == -1
    # the weight-gradient label covers the stride-2 all-taps kernel's shapes
    assert L.ymx_conv_wgrad_kernel_label(64, 416, 416, 32, 64, 3, 2) == b"wgrad_s2"
    assert L.ymx_conv_wgrad_kernel_label(64, 208, 208, 64, 128, 3, 2) == b"wgrad_dma"
    assert L.ymx_conv_wgrad_kernel_label(64, 52, 52, 512, 1024, 3, 1) == b"wgrad_slab"
