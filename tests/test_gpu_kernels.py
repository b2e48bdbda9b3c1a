"""GPU parity tests: every HIP kernel, called through the C ABI (via the Python
modules), against the CPU oracle on identical inputs.

Tolerances (the ones the task states): fp32 results within 1e-3 relative, bf16 within
1e-2 relative; NMS outputs and compaction indices bit-exact on identical decoded
inputs.  "Relative" for conv outputs is against the tensor's scale (max |reference|),
since individual outputs of a convolution can cancel to ~0.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import cref
from oracle import net_torch as N
from oracle import postproc_np as P

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
F = np.float32


def rel_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def tol_for(dtype):
    return 1e-3 if dtype == torch.float32 else 1e-2


def make_conv(p, k, down, act, bn, dtype, out_dtype=None):
    from yolo_amd.layers import YOLOConv
    cin, cout = p["kernel"].shape[2], p["kernel"].shape[3]
    layer = YOLOConv((k, k, cout), downsample=down, batchNormalization=bn, activation=act, dtype=dtype)
    layer.load_reference_params(p)
    return layer.to(DEV)


# ----------------------------------------------------------------------------- conv
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_goldens(golden_dir, dtype):
    files = sorted(glob.glob(os.path.join(golden_dir, "conv_*.npz")))
    assert len(files) == 5
    for f in files:
        g = np.load(f)
        p = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p_")}
        act = str(g["act"])
        layer = make_conv(p, int(g["k"]), bool(g["down"]), None if act == "none" else act, "gamma" in p, dtype)
        x = torch.from_numpy(g["x"]).to(DEV)
        first = x.shape[-1] == 3
        xin = x if first else x.to(dtype)            # the first layer takes the fp32 image as is
        res = torch.from_numpy(g["residual"]).to(DEV).to(dtype) if "residual" in g.files else None
        y = layer(xin, residual=res)
        assert y.dtype == dtype and tuple(y.shape) == g["y"].shape
        e = rel_err(y.float().cpu().numpy(), g["y"])
        assert e < tol_for(dtype), (os.path.basename(f), e)


CONV_SHAPES = [
    # B, H, W, Cin, Cout, k, down, act, residual, bn
    (2, 13, 13, 64, 255, 1, False, None, False, False),      # head output conv, Cout % 4 != 0
    (1, 26, 26, 32, 64, 3, False, "mish", True, False),      # Cin 32: a K step spans two taps
    (2, 8, 8, 768, 512, 1, False, "mish", False, False),     # concat width, not a power of two
    (1, 16, 16, 64, 32, 1, False, "leaky", False, True),     # narrow output tile
    (1, 20, 20, 128, 256, 3, True, "mish", False, False),    # downsample, even size
    (1, 13, 13, 512, 1024, 3, False, "mish", True, False),   # K = 4608, the heavy shape
    (3, 7, 9, 8, 24, 3, False, "mish", False, False),        # smallest MFMA-eligible channels (bf16)
    (1, 11, 11, 16, 48, 5, False, None, False, False),       # 5x5
    (1, 9, 9, 3, 20, 3, True, "mish", False, False),         # direct path, odd sizes, downsample
    (3, 9, 7, 3, 32, 3, False, "mish", False, False),        # first-layer shape (MFMA gather kernel in bf16)
    (2, 37, 41, 3, 32, 3, False, "leaky", False, True),      # same, several 64-pixel chunks, folded BN
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", range(len(CONV_SHAPES)))
def test_conv_random_shapes(case, dtype):
    B, H, W, cin, cout, k, down, act, use_res, bn = CONV_SHAPES[case]
    g = torch.Generator().manual_seed(100 + case)
    x = N.bf16_round(torch.randn(B, H, W, cin, generator=g))
    p = {"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5)}
    if bn:
        p.update(gamma=1 + 0.2 * torch.randn(cout, generator=g), beta=0.1 * torch.randn(cout, generator=g),
                 mean=0.1 * torch.randn(cout, generator=g), var=0.5 + torch.rand(cout, generator=g))
    else:
        p["bias"] = 0.1 * torch.randn(cout, generator=g)
    want = N.yolo_conv(x, p, down, act)
    res = None
    if use_res:
        res = N.bf16_round(torch.randn(*want.shape, generator=g))
        want = want + res
    layer = make_conv(p, k, down, act, bn, dtype)
    xin = x.to(DEV) if cin == 3 else x.to(DEV).to(dtype)
    y = layer(xin, residual=None if res is None else res.to(DEV).to(dtype))
    e = rel_err(y.float().cpu().numpy(), want.numpy())
    assert e < tol_for(dtype), e
    if dtype == torch.bfloat16 and cin != 3:
        # fp32 output from bf16 operands (what the head's output convs use)
        y32 = layer(xin, residual=None if res is None else res.to(DEV).to(dtype), out_dtype=torch.float32)
        assert y32.dtype == torch.float32 and rel_err(y32.cpu().numpy(), want.numpy()) < 1e-3


def test_conv_plain_entry_point_and_strides():
    """ymx_conv2d_fused (dense) and ymx_conv2d_fused_ex with pixel strides on both sides."""
    from yolo_amd import _lib
    g = torch.Generator().manual_seed(5)
    B, H, W, cin, cout = 2, 6, 5, 16, 32
    x = N.bf16_round(torch.randn(B, H, W, cin, generator=g))
    p = {"kernel": N.bf16_round(torch.randn(3, 3, cin, cout, generator=g) * 0.1), "bias": torch.randn(cout, generator=g)}
    want = N.yolo_conv(x, p, False, "mish").numpy()
    layer = make_conv(p, 3, False, "mish", False, torch.bfloat16)
    wp, scale, shift, _ = layer.packed(torch.bfloat16)
    L = _lib.lib()
    xd = x.to(DEV).bfloat16()
    y = torch.empty((B, H, W, cout), device=DEV, dtype=torch.bfloat16)
    _lib.check(L.ymx_conv2d_fused(_lib.ptr(xd), _lib.ptr(wp), None, _lib.ptr(shift), None, _lib.ptr(y), B, H, W, cin,
                                  cout, 3, 1, _lib.ACT_MISH, _lib.BF16, _lib.stream()))
    assert rel_err(y.float().cpu().numpy(), want) < 1e-2
    # input embedded in a wider buffer (x_ld = 24), output into channels 8.. of a 48-wide buffer
    xw = torch.zeros((B, H, W, 24), device=DEV, dtype=torch.bfloat16)
    xw[..., :cin] = xd
    yw = torch.full((B, H, W, 48), 7.0, device=DEV, dtype=torch.bfloat16)
    yv = yw[..., 8:]
    _lib.check(L.ymx_conv2d_fused_ex(_lib.ptr(xw), _lib.ptr(wp), None, _lib.ptr(shift), None,
                                     yv.data_ptr(), B, H, W, cin, cout, 3, 1, _lib.ACT_MISH, _lib.BF16, _lib.BF16,
                                     _lib.BF16, 24, 0, 48, _lib.stream()))
    assert torch.equal(yw[..., 8:40], y) and bool((yw[..., :8] == 7).all()) and bool((yw[..., 40:] == 7).all())


def test_conv_wide_kernels_with_pixel_strides():
    """The wide-layer kernels (resident-slab 3x3 and LDS-DMA 1x1) reading an input embedded in a
    wider buffer, adding a strided shortcut and writing into a slice of a wider output."""
    from yolo_amd import _lib
    g = torch.Generator().manual_seed(77)
    B, H, W, cin, cout = 2, 10, 12, 64, 256
    L = _lib.lib()
    try:
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 0))
        for k in (3, 1):
            x = N.bf16_round(torch.randn(B, H, W, cin, generator=g))
            p = {"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * 0.05), "bias": torch.randn(cout, generator=g)}
            res = N.bf16_round(torch.randn(B, H, W, cout, generator=g))
            want = (N.yolo_conv(x, p, False, "mish") + res).numpy()
            layer = make_conv(p, k, False, "mish", False, torch.bfloat16)
            wp, scale, shift, _ = layer.packed(torch.bfloat16)
            xd, rd = x.to(DEV).bfloat16(), res.to(DEV).bfloat16()
            y = torch.empty((B, H, W, cout), device=DEV, dtype=torch.bfloat16)
            _lib.check(L.ymx_conv2d_fused(_lib.ptr(xd), _lib.ptr(wp), None, _lib.ptr(shift), _lib.ptr(rd), _lib.ptr(y), B, H, W,
                                          cin, cout, k, 1, _lib.ACT_MISH, _lib.BF16, _lib.stream()))
            assert rel_err(y.float().cpu().numpy(), want) < 1e-2
            xw = torch.full((B, H, W, 96), 3.0, device=DEV, dtype=torch.bfloat16)
            xw[..., 8:72] = xd
            rw = torch.full((B, H, W, 264), -2.0, device=DEV, dtype=torch.bfloat16)
            rw[..., :cout] = rd
            yw = torch.full((B, H, W, 320), 7.0, device=DEV, dtype=torch.bfloat16)
            _lib.check(L.ymx_conv2d_fused_ex(xw[..., 8:].data_ptr(), _lib.ptr(wp), None, _lib.ptr(shift), _lib.ptr(rw),
                                             yw[..., 32:].data_ptr(), B, H, W, cin, cout, k, 1, _lib.ACT_MISH, _lib.BF16,
                                             _lib.BF16, _lib.BF16, 96, 264, 320, _lib.stream()))
            assert torch.equal(yw[..., 32:288], y), k
            assert bool((yw[..., :32] == 7).all()) and bool((yw[..., 288:] == 7).all()), k
    finally:
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 160))


DMA_SHAPES = [
    # B, H, W, Cin, Cout, k, down, act, residual     (all routed to the large-tile LDS-DMA kernel)
    (2, 13, 13, 64, 255, 1, False, None, False),        # one partial channel tile, ragged pixels
    (1, 13, 13, 512, 1024, 3, False, "mish", True),      # K = 4608, 4 channel tiles
    (3, 20, 20, 128, 256, 3, True, "mish", False),       # stride 2, padding on top/left only
    (2, 26, 26, 768, 512, 1, False, "leaky", False),     # Cin not a power of two
    (1, 9, 11, 32, 384, 3, False, "mish", True),         # Cin 32 (two taps per K step), 1.5 tiles
    (1, 33, 31, 8, 200, 5, False, None, False),          # 5x5, Cin 8, several pixel tiles
    (3, 13, 13, 128, 256, 3, False, "mish", True),       # 3x3 stride 1, whole 64-channel blocks: the resident-slab kernel
    (2, 26, 52, 64, 512, 3, False, "leaky", False),      # one channel block, W = 52, two channel tiles
    (1, 7, 59, 192, 136, 3, False, None, True),          # widest supported row, partial channel tile
    (40, 3, 2, 64, 256, 3, False, "mish", True),         # tiny images: most taps fall outside, one partial tile
    (70, 5, 1, 128, 256, 3, False, "leaky", False),      # one-pixel rows, tiles span dozens of images
    # narrow layers: the several-workgroups-per-CU kernel with 128/64/32-channel tiles
    (2, 24, 20, 64, 128, 3, False, "mish", True),
    (1, 31, 33, 32, 64, 3, True, "mish", False),         # stride 2, one tap per K step
    (2, 19, 17, 32, 64, 3, False, "mish", True),
    (2, 21, 22, 64, 32, 1, False, "mish", False),        # 32-channel tile
    (1, 15, 15, 128, 100, 1, False, "leaky", True),      # partial 128 tile
    (1, 17, 9, 16, 24, 3, False, None, False),           # partial 32 tile, Cin 16
    (1, 12, 12, 72, 48, 3, False, "mish", True),         # Cin not a power of two, partial 64 tile
    (3, 9, 208, 32, 64, 3, False, "mish", True),         # input-slab kernel, rows longer than a tile
    (5, 11, 104, 64, 128, 3, False, "leaky", False),     # input-slab kernel, 64 -> 128
    (2, 30, 7, 32, 40, 3, False, None, True),            # input-slab kernel, short rows, partial channel tile
    # short K: the two-stage variants (four / three workgroups per CU, 32-channel transpose images)
    (2, 23, 19, 128, 64, 1, False, "mish", True),        # 64-channel tile, shortcut fetched in two halves
    (3, 17, 13, 96, 56, 1, False, "leaky", True),        # partial 64 tile, K = 96
    (2, 21, 22, 64, 24, 1, False, "mish", True),         # 32-channel tile, two K steps, shortcut
    (1, 40, 9, 8, 64, 3, False, None, False),            # 3x3 with K = 72 on the 64-channel tile
    # rows at and beyond the input-slab kernel's LDS limit (image sides 686 ... 800 reach these widths)
    (1, 3, 343, 32, 64, 3, False, "mish", True),         # longest 32-channel row the slab kernel takes (80 KB)
    (1, 2, 352, 32, 64, 3, False, "mish", True),         # one past it: the streaming kernel
    (1, 2, 400, 32, 64, 3, False, "leaky", False),
    (1, 3, 123, 64, 128, 3, False, "mish", True),        # longest 64-channel row of the slab kernel
    (1, 3, 124, 64, 128, 3, False, "mish", False),
    # rows beyond the 256-pixel form of the resident-slab kernel (W > 59): its 192-pixel form up to W = 91
    (1, 5, 60, 64, 256, 3, False, "mish", True),
    (1, 4, 76, 128, 256, 3, False, "mish", True),        # the 76-wide maps of a 608 x 608 input
    (1, 3, 91, 64, 160, 3, False, "leaky", False),       # the widest row it takes
    (1, 3, 92, 64, 256, 3, False, "mish", True),         # one past it: the streaming kernels
]


@pytest.mark.parametrize("case", range(len(DMA_SHAPES)))
def test_conv_dma_kernel_forced(case):
    """The LDS-DMA kernels (both wide-layer variants and the narrow-layer tiles) on shapes
    they would not normally be picked for (dma_min_tiles = 0), bf16 and fp32 outputs,
    against the oracle and against the register-staged kernel (same summation order:
    identical bits)."""
    from yolo_amd import _lib
    B, H, W, cin, cout, k, down, act, use_res = DMA_SHAPES[case]
    g = torch.Generator().manual_seed(300 + case)
    x = N.bf16_round(torch.randn(B, H, W, cin, generator=g))
    p = {"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5),
         "bias": 0.1 * torch.randn(cout, generator=g)}
    want = N.yolo_conv(x, p, down, act)
    res = None
    if use_res:
        res = N.bf16_round(torch.randn(*want.shape, generator=g))
        want = want + res
    layer = make_conv(p, k, down, act, False, torch.bfloat16)
    xin = x.to(DEV).bfloat16()
    rin = None if res is None else res.to(DEV).bfloat16()
    L = _lib.lib()
    try:
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 1 << 40))
        y_small = layer(xin, residual=rin)
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 0))
        for narrow in (1, 2):        # narrow layers: streamed input tiles / input slab resident in LDS
            _lib.check(L.ymx_set_option(b"dma_narrow", narrow))
            assert torch.equal(layer(xin, residual=rin), y_small), narrow
        for variant in (1, 2, 4, 5):  # two-stage / two workgroups per CU / staggered groups / resident input slab (3x3)
            _lib.check(L.ymx_set_option(b"dma_variant", variant))
            y = layer(xin, residual=rin)
            y32 = layer(xin, residual=rin, out_dtype=torch.float32)
            assert rel_err(y.float().cpu().numpy(), want.numpy()) < 1e-2, variant
            assert rel_err(y32.cpu().numpy(), want.numpy()) < 1e-3, variant
            assert torch.equal(y, y_small), variant
        for pixels in (192, 256):     # both tile sizes of the resident-slab kernel (variant 5 is set)
            _lib.check(L.ymx_set_option(b"halo_pixels", pixels))
            assert torch.equal(layer(xin, residual=rin), y_small), pixels
            y32 = layer(xin, residual=rin, out_dtype=torch.float32)
            assert rel_err(y32.cpu().numpy(), want.numpy()) < 1e-3, pixels
        _lib.check(L.ymx_set_option(b"dma_variant", 4))
        for pixels in (192, 256):     # both tile sizes of the phase-staggered kernel
            _lib.check(L.ymx_set_option(b"dma4_pixels", pixels))
            assert torch.equal(layer(xin, residual=rin), y_small), pixels
            y32 = layer(xin, residual=rin, out_dtype=torch.float32)
            assert rel_err(y32.cpu().numpy(), want.numpy()) < 1e-3, pixels
    finally:
        _lib.check(L.ymx_set_option(b"dma_min_tiles", 160))
        _lib.check(L.ymx_set_option(b"dma_variant", 5))
        _lib.check(L.ymx_set_option(b"dma_narrow", 2))
        _lib.check(L.ymx_set_option(b"halo_pixels", 0))
        _lib.check(L.ymx_set_option(b"dma4_pixels", 0))


MIXED_SHAPES = [   # B, H, W, Cin, Cout, activation, shortcut, k, stride: launches of two to four rounds of 256-pixel tiles
    (64, 52, 52, 128, 256, "mish", True, 3, 1),        # 676 tiles: 512 of 256 pixels + 219 of 192
    (61, 26, 26, 128, 1000, "leaky", False, 3, 1),     # partial last pixel tile of the 192-pixel part, partial channel tile
    (33, 40, 40, 64, 512, "mish", True, 3, 1),         # 207 pixel tiles x 2: one whole round of 256-pixel tiles, ragged rest
    (64, 52, 52, 256, 512, "mish", True, 1, 1),        # the phase-staggered kernel: 1x1, 1 352 tiles
    (37, 52, 52, 192, 264, "leaky", False, 1, 1),      # ragged pixels and channels, K = 192
    (64, 104, 104, 64, 256, "mish", False, 3, 2),      # stride 2, 676 tiles
]


@pytest.mark.parametrize("case", range(len(MIXED_SHAPES)))
def test_conv_halo_mixed_tiles(case):
    """The wide kernels (resident-slab 3x3; phase-staggered 1x1 and stride 2) with both tile forms in
    one launch (whole rounds of 256-pixel tiles, the remaining pixels as 192-pixel tiles): the same
    bits as either pure form, bf16 and fp32 outputs, and the layer's own choice among the three is
    one of them."""
    from yolo_amd import _lib
    B, H, W, cin, cout, act, use_res, k, stride = MIXED_SHAPES[case]
    g = torch.Generator().manual_seed(700 + case)
    xin = torch.randn(B, H, W, cin, generator=g).to(DEV).bfloat16()
    p = {"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5),
         "bias": 0.1 * torch.randn(cout, generator=g)}
    oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
    rin = torch.randn(B, oh, ow, cout, generator=g).to(DEV).bfloat16() if use_res else None
    layer = make_conv(p, k, stride == 2, act, False, torch.bfloat16)
    halo = k == 3 and stride == 1
    opt = b"halo_pixels" if halo else b"dma4_pixels"
    names = {256: "halo", 192: "halo192", 448: "halo_mixed"} if halo else {256: "dma4", 192: "dma4_192", 448: "dma4_mixed"}
    L = _lib.lib()
    op = _lib.ymx_op()
    op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, B, H, W, cin, cout, k, stride
    op.dtype = op.x_dtype = op.out_dtype = _lib.BF16
    op.x = op.w = op.y = 256
    import ctypes
    try:
        outs = {}
        for pixels in (256, 192, 448):
            _lib.check(L.ymx_set_option(opt, pixels))
            assert L.ymx_conv_kernel_label(ctypes.byref(op)).decode() == names[pixels]
            outs[pixels] = (layer(xin, residual=rin).clone(), layer(xin, residual=rin, out_dtype=torch.float32).clone())
        for pixels in (192, 448):
            assert torch.equal(outs[pixels][0], outs[256][0]), pixels
            assert torch.equal(outs[pixels][1], outs[256][1]), pixels
        _lib.check(L.ymx_set_option(opt, 0))
        assert torch.equal(layer(xin, residual=rin), outs[256][0])
        # against the oracle on a slice of the batch (first and last image)
        for b in (0, B - 1):
            want = N.yolo_conv(xin[b:b + 1].float().cpu(), p, stride == 2, act)
            if use_res:
                want = want + rin[b:b + 1].float().cpu()
            assert rel_err(outs[448][1][b:b + 1].cpu().numpy(), want.numpy()) < 1e-3
    finally:
        _lib.check(L.ymx_set_option(opt, 0))


STEM_CASES = [   # B, H, W, activation, batch norm
    (2, 64, 64, "mish", False), (1, 416, 416, "mish", False), (3, 50, 38, "leaky", True), (2, 33, 47, "mish", True),
    (1, 17, 130, "leaky", False), (5, 16, 32, "mish", False), (1, 2, 2, "mish", False), (1, 19, 5, "leaky", False),
]


@pytest.mark.parametrize("case", range(len(STEM_CASES)))
def test_conv_stem_fused_matches_two_launches(case):
    """The backbone's first two layers (backbones.py:11-12) as one launch: the same bits as the
    first-layer kernel followed by the stride-2 kernel, for image sizes on and off the 8 x 16
    output tiles (odd sides, maps smaller than a tile), both activations, bias and folded batch
    norm; and within the bf16 tolerance of the oracle's two layers."""
    from yolo_amd import _lib
    B, H, W, act, bn = STEM_CASES[case]
    g = torch.Generator().manual_seed(900 + case)
    x = torch.rand(B, H, W, 3, generator=g)
    params = []
    for cin, cout in ((3, 32), (32, 64)):
        p = {"kernel": N.bf16_round(torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5)}
        if bn:
            p.update(gamma=1 + 0.2 * torch.randn(cout, generator=g), beta=0.1 * torch.randn(cout, generator=g),
                     mean=0.1 * torch.randn(cout, generator=g), var=0.5 + torch.rand(cout, generator=g))
        else:
            p["bias"] = 0.1 * torch.randn(cout, generator=g)
        params.append(p)
    l0 = make_conv(params[0], 3, False, act, bn, torch.bfloat16)
    l1 = make_conv(params[1], 3, True, act, bn, torch.bfloat16)
    xin = x.to(DEV)
    mid = l0(xin)
    want = l1(mid)
    w0, sc0, sh0, _ = l0.packed(torch.bfloat16)
    w1, sc1, sh1, _ = l1.packed(torch.bfloat16)
    y = torch.full_like(want, float("nan"))
    _lib.launch("ymx_conv_stem_fused", xin, _lib.ptr(xin), _lib.ptr(w0), _lib.ptr(sc0), _lib.ptr(sh0), _lib.ptr(w1), _lib.ptr(sc1),
                _lib.ptr(sh1), _lib.ptr(y), B, H, W, _lib.ACT_MISH if act == "mish" else _lib.ACT_LEAKY, 0)
    assert torch.equal(y, want), float((y.float() - want.float()).abs().max())
    # through the program executor: op 0 marked as read by op 1 alone
    ops = (_lib.ymx_op * 2)()
    y2 = torch.full_like(want, float("nan"))
    mid2 = torch.full_like(mid, float("nan"))
    for o, (layer, src, dst, cin, cout, stride, w, sc, sh) in zip(ops, ((l0, xin, mid2, 3, 32, 1, w0, sc0, sh0),
                                                                          (l1, mid2, y2, 32, 64, 2, w1, sc1, sh1))):
        o.kind, o.B, o.H, o.W, o.Cin, o.Cout, o.k, o.stride, o.act = 1, B, H, W, cin, cout, 3, stride, layer.act_code
        o.dtype = o.out_dtype = _lib.BF16
        o.x_dtype = _lib.F32 if cin == 3 else _lib.BF16
        o.x, o.w, o.y = src.data_ptr(), w.data_ptr(), dst.data_ptr()
        o.scale = None if sc is None else sc.data_ptr()
        o.shift = None if sh is None else sh.data_ptr()
    ops[0].C1 = 1
    _lib.check(_lib.lib().ymx_run_program(ops, 2, _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(y2, want) and bool(torch.isnan(mid2.float()).all())      # fused: the 32-channel map was never written
    ops[0].C1 = 0
    _lib.check(_lib.lib().ymx_run_program(ops, 2, _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(mid2, mid) and torch.equal(y2, want)
    ref = N.yolo_conv(N.bf16_round(N.yolo_conv(x, params[0], False, act)), params[1], True, act)
    assert rel_err(y.float().cpu().numpy(), ref.numpy()) < 2e-2


RACE_SHAPES = [   # B, H, W, Cin, Cout, k, stride: every LDS-DMA kernel family at benchmark-sized layers
    (64, 26, 26, 256, 512, 3, 1), (48, 13, 13, 512, 1024, 3, 1), (64, 52, 52, 1024, 512, 1, 1),
    (64, 26, 26, 512, 1024, 3, 2), (64, 208, 208, 32, 64, 3, 1), (48, 104, 104, 64, 128, 3, 1),
    (64, 104, 104, 128, 64, 1, 1), (64, 52, 52, 128, 256, 3, 1),     # the last one: 256-pixel tiles of the resident-slab kernel
]


def test_conv_kernels_repeat_bit_exact_under_concurrent_load():
    """Race screen: each LDS-DMA kernel, at benchmark-sized layers, run repeatedly while a
    second stream keeps the GPU busy with other convolutions (the engine runs two programs
    at once, so workgroups start at irregular times next to foreign traffic); every output
    must equal the register-staged kernel's bit for bit.  tools/race_screen.py is the long
    version (all shapes, thousands of repetitions)."""
    from yolo_amd import _lib
    from yolo_amd.layers import YOLOConv
    L = _lib.lib()

    def set_kernel(tiles, variant, narrow):
        _lib.check(L.ymx_set_option(b"dma_min_tiles", tiles))
        _lib.check(L.ymx_set_option(b"dma_variant", variant))
        _lib.check(L.ymx_set_option(b"dma_narrow", narrow))

    g = torch.Generator().manual_seed(77)
    side = torch.cuda.Stream()
    px = torch.randn(24, 26, 26, 256, generator=g).to(DEV).bfloat16()
    pl = YOLOConv((3, 3, 512), activation="mish", dtype=torch.bfloat16)
    pl.build(256, generator=g)
    pl.to(DEV)
    try:
        for (B, H, W, cin, cout, k, stride) in RACE_SHAPES:
            x = torch.randn(B, H, W, cin, generator=g).to(DEV).bfloat16()
            oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
            res = torch.randn(B, oh, ow, cout, generator=g).to(DEV).bfloat16()
            layer = YOLOConv((k, k, cout), downsample=(stride == 2), activation="mish", dtype=torch.bfloat16)
            layer.build(cin, generator=g)
            with torch.no_grad():
                layer.kernel.mul_(10.0)
            layer.to(DEV)
            set_kernel(1 << 40, 1, 0)
            ref = layer(x, residual=res).view(torch.int16).clone()
            torch.cuda.synchronize()
            variants = ((1, 1), (4, 1), (5, 2)) if cout > 128 else ((5, 1), (5, 2))
            for variant, narrow in variants:
                bad = torch.zeros((), device=DEV, dtype=torch.int64)
                for it in range(25):
                    set_kernel(0, variant, narrow)
                    bad += (layer(x, residual=res).view(torch.int16) != ref).sum()
                    with torch.cuda.stream(side):
                        set_kernel(0, 5, 2)
                        pl(px)
                side.synchronize()
                assert int(bad.item()) == 0, (B, H, W, cin, cout, k, stride, variant, narrow)
    finally:
        set_kernel(160, 5, 2)


def test_layers_standalone():
    from yolo_amd import layers
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 5, 7, 16, generator=g) * 4
    got = layers.Mish()(x.to(DEV)).cpu()
    assert rel_err(got.numpy(), N.mish(x).numpy()) < 1e-5
    got = layers.Mish()(x.to(DEV).bfloat16()).float().cpu()
    assert rel_err(got.numpy(), N.mish(N.bf16_round(x)).numpy()) < 1e-2
    bn = layers.BatchNormalization(16)
    with torch.no_grad():
        bn.gamma.copy_(torch.rand(16, generator=g) + 0.5); bn.beta.copy_(torch.randn(16, generator=g))
        bn.moving_mean.copy_(torch.randn(16, generator=g)); bn.moving_variance.copy_(torch.rand(16, generator=g) + 0.5)
    want = (x - bn.moving_mean) / torch.sqrt(bn.moving_variance + 1e-3) * bn.gamma.detach() + bn.beta.detach()
    got = bn.to(DEV)(x.to(DEV)).cpu()
    assert rel_err(got.numpy(), want.numpy()) < 1e-5
    # training=True normalises with the batch's own statistics (tests/test_gpu_train_bn_csp.py covers the numbers)
    tr = bn(x.to(DEV), training=True).detach().cpu()
    assert float(((tr - bn.beta.detach().cpu()) / bn.gamma.detach().cpu()).mean(dim=(0, 1, 2)).abs().max()) < 1e-4
    r = layers.RouteGroup(2, 1)(x.to(DEV))
    assert r.shape == (2, 5, 7, 8) and torch.equal(r.cpu(), x[..., 8:])
    # callable activation runs after the kernel
    p = {"kernel": torch.randn(1, 1, 16, 8, generator=g), "bias": torch.zeros(8)}
    layer = layers.YOLOConv((1, 1, 8), activation=torch.relu)
    layer.load_reference_params(p)
    got = layer.to(DEV)(x.to(DEV)).cpu()
    assert rel_err(got.numpy(), torch.relu(N.yolo_conv(x, p, False, None)).numpy()) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_concat(golden_dir, dtype):
    from yolo_amd import layers
    g = np.load(os.path.join(golden_dir, "upsample.npz"))
    x = torch.from_numpy(g["x"]).to(DEV).to(dtype)
    y = layers.Upsample()(x)
    assert rel_err(y.float().cpu().numpy(), g["y"]) < tol_for(dtype)
    gen = torch.Generator().manual_seed(3)
    a = N.bf16_round(torch.randn(2, 7, 4, 24, generator=gen))
    skip = N.bf16_round(torch.randn(2, 14, 8, 8, generator=gen))
    want = torch.cat([skip, N.upsample2x(a)], -1)
    got = layers.upsample2x_concat(skip.to(DEV).to(dtype), a.to(DEV).to(dtype))
    assert got.shape == want.shape
    assert torch.equal(got[..., :8].float().cpu(), skip)
    assert rel_err(got.float().cpu().numpy(), want.numpy()) < tol_for(dtype)


# ----------------------------------------------------------------------------- decode / filter
def _golden_heads(g):
    return [g["head%d" % i] for i in range(3)]


@pytest.mark.parametrize("name", ["postproc_c3", "postproc_c80", "postproc_c1"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decode_goldens(golden_dir, name, dtype):
    import yolo_amd as Y
    from yolo_amd import utils as U
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    C = int(g["C"])
    heads_np = _golden_heads(g)
    heads = [torch.from_numpy(h).to(DEV).to(dtype) for h in heads_np]
    if dtype == torch.bfloat16:
        want_b, want_s = P.decode_all([h.float().cpu().numpy() for h in heads], C)
    else:
        want_b, want_s = g["xywh"], g["scores"]
    cfg = Y.config
    # one call for all scales
    r = U.decode_heads(heads, C, cfg["strides"], cfg["xyscale"], U.default_anchors())
    n = r["N"]
    assert n == want_b.shape[1] and r["counts"].cpu().tolist() == [n] * heads[0].shape[0]
    got_b = r["boxes"][:, :n].cpu().numpy()
    got_s = r["scores_cm"][:, :, :n].transpose(1, 2).cpu().numpy()
    np.testing.assert_allclose(got_b, want_b, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(got_s, want_s, rtol=1e-3, atol=1e-6)
    # the per-scale API functions of the reference
    off = 0
    for i, h in enumerate(heads):
        S = h.shape[1]
        bx, sc = Y.decode_predict(h, i, S, C, cfg["strides"], cfg["xyscale"], U.default_anchors())
        m = 3 * S * S
        assert bx.shape == (h.shape[0], m, 4) and sc.shape == (h.shape[0], m, C)
        np.testing.assert_allclose(bx.cpu().numpy(), want_b[:, off:off + m], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(sc.cpu().numpy(), want_s[:, off:off + m], rtol=1e-3, atol=1e-6)
        xywh, conf, prob = Y.decode(h, i, S, C, cfg["strides"], cfg["xyscale"], U.default_anchors())
        w_xywh, w_conf, w_prob = P.decode(h.float().cpu().numpy(), i, S, C)
        np.testing.assert_allclose(xywh.cpu().numpy(), w_xywh, rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(conf.cpu().numpy(), w_conf, rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(prob.cpu().numpy(), w_prob, rtol=1e-3, atol=1e-6)
        tr = Y.decode_train(h, i, S, C, cfg["strides"], cfg["xyscale"], U.default_anchors())
        np.testing.assert_allclose(tr.cpu().numpy(), P.decode_train(h.float().cpu().numpy(), i, S, C), rtol=1e-3, atol=1e-4)
        off += m


@pytest.mark.parametrize("name", ["postproc_c3", "postproc_c80", "postproc_c1"])
def test_filter_goldens_exact(golden_dir, name):
    """filter_bboxes on the golden decoded tensors: survivors, order and values exact."""
    import yolo_amd as Y
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    xywh = torch.from_numpy(g["xywh"]).to(DEV)
    scores = torch.from_numpy(g["scores"]).to(DEV)
    for sc_in in (scores, scores.transpose(1, 2).contiguous().transpose(1, 2)):     # candidate- and class-major
        out = Y.filter_bboxes(xywh, sc_in, float(g["thr"]), float(g["image_size"]))
        boxes, sc = out
        np.testing.assert_array_equal(out.counts.cpu().numpy(), g["f_counts"])
        np.testing.assert_array_equal(out.src_idx.cpu().numpy(), g["f_src"])
        np.testing.assert_array_equal(sc.cpu().numpy(), g["f_scores"])
        np.testing.assert_allclose(boxes.cpu().numpy(), g["f_boxes"], rtol=1e-6, atol=1e-7)


def _check_fused_filter(heads_t, C, thr, image_size):
    """decode+filter in one launch vs the oracle applied to the GPU's own decode:
    identical decoded inputs -> identical survivors, in order."""
    import yolo_amd as Y
    from yolo_amd import utils as U
    cfg = Y.config
    dec = U.decode_heads(heads_t, C, cfg["strides"], cfg["xyscale"], U.default_anchors())
    n = dec["N"]
    d_boxes = dec["boxes"][:, :n].cpu().numpy()
    d_scores = dec["scores_cm"][:, :, :n].transpose(1, 2).cpu().numpy()
    fb, fs, counts, src = cref.filter_bboxes(d_boxes, d_scores, thr, image_size)
    r = U.decode_heads(heads_t, C, cfg["strides"], cfg["xyscale"], U.default_anchors(), filter=True,
                       score_threshold=thr, image_size=image_size)
    got_counts = r["counts"].cpu().numpy()
    np.testing.assert_array_equal(got_counts, counts)
    ws = r["workspace"].cpu().numpy().view(np.int32)
    assert ws[0] == 0, "look-back spin limit hit"
    for b in range(len(counts)):
        m = int(counts[b])
        np.testing.assert_array_equal(r["src_idx"][b, :m].cpu().numpy(), src[b, :m])
        np.testing.assert_array_equal(r["scores_cm"][b, :, :m].cpu().numpy().T, fs[b, :m])
        np.testing.assert_allclose(r["boxes"][b, :m].cpu().numpy(), fb[b, :m], rtol=1e-6, atol=1e-7)
    return r, (fb, fs, counts, src)


@pytest.mark.parametrize("name", ["postproc_c3", "postproc_c80", "postproc_c1"])
def test_fused_decode_filter(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    heads = [torch.from_numpy(h).to(DEV) for h in _golden_heads(g)]
    _check_fused_filter(heads, int(g["C"]), float(g["thr"]), float(g["image_size"]))


def test_fused_decode_filter_416_edges():
    """Real 416^2 geometry (52/26/13: ragged last segments), all-pass, none-pass, sparse."""
    gen = torch.Generator().manual_seed(9)
    C, B = 80, 3
    heads = [torch.randn(B, S, S, 3, 5 + C, generator=gen) * 2 for S in (52, 26, 13)]
    heads_t = [h.to(DEV) for h in heads]
    for thr in (0.25, -1.0, 2.0, 0.9):
        r, (_, _, counts, _) = _check_fused_filter(heads_t, C, thr, 416.0)
        if thr < 0:
            assert counts.tolist() == [10647] * B
        if thr > 1:
            assert counts.tolist() == [0] * B


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decode_filter_strided_cells(dtype):
    """ymx_decode_filter_strided on maps with padded grid cells (garbage in the padding) gives
    exactly what ymx_decode_filter gives on the dense maps, filtered and unfiltered."""
    import ctypes
    from yolo_amd import _lib, utils as U
    gen = torch.Generator().manual_seed(21)
    C, B = 80, 2
    sides, pads = (52, 26, 13), (1, 9, 0)                  # cell strides 256, 264, 255
    dense = [(torch.randn(B, S, S, 3 * (5 + C), generator=gen) * 2).to(DEV).to(dtype) for S in sides]
    padded = []
    for h, pad in zip(dense, pads):
        t = torch.full(h.shape[:3] + (h.shape[3] + pad,), float("nan"), device=DEV, dtype=dtype)
        t[..., :h.shape[3]] = h
        padded.append(t)
    anc = U.default_anchors()
    strides, xys = [8, 16, 32], [1.2, 1.1, 1.05]
    L = _lib.lib()
    for filt, thr in ((0, 0.0), (1, 0.25), (1, -1.0)):
        outs = []
        for maps, strided in ((dense, False), (padded, True)):
            sc = U._Scales(maps, strides, xys, anc, cell_ld=[m.shape[-1] for m in maps] if strided else None)
            cap = U._round4(sc.N)
            boxes = torch.zeros((B, cap, 4), device=DEV)
            scores = torch.zeros((B, C, cap), device=DEV)
            src = torch.zeros((B, cap), device=DEV, dtype=torch.int32)
            counts = torch.zeros((B,), device=DEV, dtype=torch.int32)
            wsb = L.ymx_decode_workspace_bytes(B, sc.n, sc.S)
            ws = torch.empty((wsb,), device=DEV, dtype=torch.uint8)
            tail = (sc.n, sc.S, B, C, _lib.dtype_code(dtype), sc.anchors, sc.strides, sc.xyscale, filt, thr, 416.0,
                    _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(src), _lib.ptr(counts), cap, _lib.ptr(ws), wsb, _lib.stream())
            if strided:
                _lib.check(L.ymx_decode_filter_strided(sc.ptrs, sc.cell_ld, *tail))
            else:
                _lib.check(L.ymx_decode_filter(sc.ptrs, *tail))
            n = counts.cpu().numpy()
            outs.append((n, [boxes[b, :n[b]].cpu().numpy() for b in range(B)],
                         [scores[b, :, :n[b]].cpu().numpy() for b in range(B)],
                         [src[b, :n[b]].cpu().numpy() for b in range(B)] if filt else None))
        (n0, b0, s0, i0), (n1, b1, s1, i1) = outs
        np.testing.assert_array_equal(n0, n1)
        for b in range(B):
            np.testing.assert_array_equal(b0[b], b1[b])
            np.testing.assert_array_equal(s0[b], s1[b])
            if filt:
                np.testing.assert_array_equal(i0[b], i1[b])
    # a cell stride below 3*(5+C) is refused
    bad = (ctypes.c_int * 3)(254, 255, 255)
    sc = U._Scales(dense, strides, xys, anc)
    rc = L.ymx_decode_filter_strided(sc.ptrs, bad, *tail)
    assert rc != 0 and b"cell stride" in L.ymx_last_error()


# ----------------------------------------------------------------------------- NMS
def _gpu_nms(boxes, scores, counts, per_class, total, iou, thr, src=None):
    import yolo_amd as Y
    out = Y.combined_non_max_suppression(
        torch.from_numpy(np.ascontiguousarray(boxes)).to(DEV)[:, :, None, :], torch.from_numpy(np.ascontiguousarray(scores)).to(DEV),
        per_class, total, iou, thr, counts=None if counts is None else torch.from_numpy(counts).to(DEV),
        src_idx=None if src is None else torch.from_numpy(src).to(DEV))
    return [t.cpu().numpy() for t in out] + [out.keep_idx.cpu().numpy()]


def _assert_nms_equal(got, want):
    for a, b, name in zip(got, want, ("boxes", "scores", "classes", "valid", "keep_idx")):
        np.testing.assert_array_equal(a, b, err_msg=name)


@pytest.mark.parametrize("name", ["nms_dense", "nms_ties"])
def test_nms_goldens_bit_exact(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    got = _gpu_nms(g["boxes"], g["scores"], g["counts"], int(g["per_class"]), int(g["total"]), float(g["iou"]), float(g["thr"]))
    _assert_nms_equal(got, [g[k] for k in ("n_boxes", "n_scores", "n_classes", "n_valid", "n_keep")])


@pytest.mark.parametrize("name", ["postproc_c3", "postproc_c80", "postproc_c1"])
def test_nms_on_golden_filtered_inputs(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    got = _gpu_nms(g["f_boxes"], g["f_scores"], g["f_counts"], int(g["per_class"]), int(g["total"]), float(g["iou"]),
                   float(g["thr"]), src=g["f_src"])
    _assert_nms_equal(got, [g[k] for k in ("n_boxes", "n_scores", "n_classes", "n_valid", "n_keep")])


def test_nms_edge_cases():
    cases = [
        ([[0, 0, .5, .5]] * 3, [[.9], [.8], [.7]], {}),                                    # identical boxes
        ([[0, 0, .5, 1.0], [0, 0, .5, .5]], [[.9], [.8]], dict(iou=0.5)),                  # IoU == threshold: kept
        ([[0, 0, .5, 1.0], [0, 0, .5, .5]], [[.9], [.8]], dict(iou=0.49)),
        ([[0, 0, .1, .1], [.5, .5, .6, .6]], [[.25], [.26]], {}),                          # strict score threshold
        ([[0, 0, .1, .1], [.5, .5, .6, .6], [.8, .8, .9, .9]], [[.5, .5]] * 3, {}),        # ties: class, then index
        ([[0, 0, .5, .5], [0, 0, .5, .45], [.6, .6, .9, .9]], [[.7]] * 3, {}),             # tie inside a class
        ([[.2, .2, .2, .6], [.2, .2, .2, .6], [.5, .5, 0, 0], [0, 0, .5, .5]], [[.9], [.8], [.7], [.6]], {}),
        ([[-.2, -.1, .3, 1.4]] + [[.1 * i, .1 * i, .1 * i + .05, .1 * i + .05] for i in range(1, 9)],
         [[.95]] + [[.9 - .01 * i] for i in range(1, 9)], dict(per_class=5, total=3)),     # clip + caps
        ([[0, 0, 1, 1]], [[0.1]], {}),                                                     # nothing passes
    ]
    for boxes, scores, kw in cases:
        b = np.asarray(boxes, F)[None]
        s = np.asarray(scores, F)[None]
        args = (kw.get("per_class", 32), kw.get("total", 32), kw.get("iou", 0.45), kw.get("thr", 0.25))
        want = P.combined_nms(b, s, None, *args)
        _assert_nms_equal(_gpu_nms(b, s, None, *args), want)


def _random_boxes(rng, B, M, sigma=0.5, mu=-3.0):
    cy, cx = rng.random((B, M)), rng.random((B, M))
    h, w = np.exp(rng.normal(mu, sigma, (B, M))), np.exp(rng.normal(mu, sigma, (B, M)))
    return np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(F)


def test_nms_large_random_vs_c_oracle():
    """20k candidates x 80 classes, threshold 0.001: many buffer cuts per class."""
    rng = np.random.default_rng(5)
    B, M, C = 3, 20000, 80
    boxes = _random_boxes(rng, B, M)
    scores = (rng.random((B, M, C)) ** 4).astype(F)
    counts = np.array([M, M - 1237, 17], np.int32)
    want = cref.combined_nms(boxes, scores, counts, 32, 32, 0.45, 0.001)
    _assert_nms_equal(_gpu_nms(boxes, scores, counts, 32, 32, 0.45, 0.001), want)
    want = cref.combined_nms(boxes, scores, counts, 64, 100, 0.3, 0.5)
    _assert_nms_equal(_gpu_nms(boxes, scores, counts, 64, 100, 0.3, 0.5), want)


def test_nms_deep_suppression_needs_further_tiers():
    """Thousands of near-identical high-score boxes ahead of the few distinct ones: the
    greedy pass must walk far past the first 512 candidates (several tiers)."""
    rng = np.random.default_rng(6)
    M, C = 6000, 2
    boxes = np.tile(np.array([[.2, .2, .6, .6]], F), (M, 1)) + rng.normal(0, 1e-3, (M, 4)).astype(F)
    scores = np.zeros((M, C), F)
    scores[:, 0] = np.linspace(0.99, 0.5, M, dtype=F)
    scores[:, 1] = (rng.random(M) * 0.4 + 0.5).astype(F)
    distinct = rng.choice(M, 40, replace=False)
    boxes[distinct] = _random_boxes(rng, 1, 40, mu=-3.5)[0]
    scores[distinct, 0] = np.linspace(0.3, 0.26, 40, dtype=F)          # visited last
    scores[:, 1] = np.round(scores[:, 1] * 32) / 32                    # heavy ties in class 1
    b, s = boxes[None], scores[None]
    want = cref.combined_nms(b, s, None, 32, 64, 0.45, 0.25)
    assert want[3][0] > 33                                             # the distinct boxes were reached
    _assert_nms_equal(_gpu_nms(b, s, None, 32, 64, 0.45, 0.25), want)


def test_bbox_nms_end_to_end(golden_dir):
    """heads -> bbox_nms in one go.  Compared with the oracle run on the GPU's own decoded
    values (NMS is exact only on identical decoded inputs)."""
    import yolo_amd as Y
    for name in ("postproc_c3", "postproc_c80", "postproc_c1"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        C, thr, size = int(g["C"]), float(g["thr"]), float(g["image_size"])
        heads = [torch.from_numpy(h).to(DEV) for h in _golden_heads(g)]
        _, (fb, fs, counts, src) = _check_fused_filter(heads, C, thr, size)
        want = cref.combined_nms(fb, fs, counts, int(g["per_class"]), int(g["total"]), float(g["iou"]), thr, src_idx=src)
        raw = [h.reshape(h.shape[0], h.shape[1], h.shape[2], -1) for h in heads]      # [B,S,S,3(5+C)] conv maps
        out = Y.bbox_nms(raw, size, int(g["per_class"]), int(g["total"]), float(g["iou"]), thr)
        got = [t.cpu().numpy() for t in out] + [out.keep_idx.cpu().numpy()]
        np.testing.assert_allclose(got[0], want[0], rtol=1e-6, atol=1e-7)
        for a, b in zip(got[1:], want[1:]):
            np.testing.assert_array_equal(a, b)
        # and against the committed golden result (decode differences are far below any score gap here)
        np.testing.assert_array_equal(got[4], g["n_keep"])
        np.testing.assert_array_equal(got[3], g["n_valid"])


def test_bbox_iou_giou():
    import yolo_amd as Y
    rng = np.random.default_rng(8)
    a = rng.random((5, 1, 4)).astype(F) * 10
    b = rng.random((1, 7, 4)).astype(F) * 10
    b[0, 0] = 0
    a[0, 0] = 0
    for fn, ref in ((Y.bbox_iou, P.bbox_iou), (Y.bbox_giou, P.bbox_giou)):
        got = fn(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
        assert got.shape == (5, 7)
        np.testing.assert_allclose(got, ref(a, b), rtol=1e-5, atol=1e-6)


RESBLOCK_CASES = [   # B, H, W, activation, batch norm, channels
    (2, 64, 64, "mish", False, 64), (1, 208, 208, "mish", False, 64), (3, 50, 38, "leaky", True, 64), (2, 33, 47, "mish", True, 64),
    (1, 17, 130, "leaky", False, 64), (5, 8, 16, "mish", False, 64), (1, 1, 1, "mish", False, 64), (1, 19, 5, "leaky", False, 64),
    (2, 9, 17, "mish", False, 64), (1, 7, 15, "mish", True, 64),
    (2, 64, 64, "mish", False, 128), (1, 104, 104, "mish", False, 128), (3, 50, 38, "leaky", True, 128), (2, 33, 47, "mish", True, 128),
    (1, 17, 130, "leaky", False, 128), (5, 8, 16, "mish", False, 128), (1, 1, 1, "mish", False, 128), (1, 19, 5, "leaky", False, 128),
    (2, 9, 17, "mish", False, 128), (300, 8, 16, "mish", True, 128),
]


@pytest.mark.parametrize("case", range(len(RESBLOCK_CASES)))
def test_conv_resblock_fused_matches_two_launches(case):
    """ResidialBlock(32, 64) on a 64-channel map and ResidialBlock(64, 128) on a 128-channel map
    (backbones.py:13-16) as one launch: the same bits as the
    1x1 kernel followed by the 3x3 kernel with the shortcut, for map sizes on and off the 8 x 16
    output tiles (odd sides, maps smaller than a tile, one pixel past a tile), both activations,
    bias and folded batch norm; through the C entry point and through the program executor (the
    32-channel map is then never written); and within the bf16 tolerance of the oracle's block."""
    from yolo_amd import _lib
    B, H, W, act, bn, C = RESBLOCK_CASES[case]
    g = torch.Generator().manual_seed(950 + case)
    x = N.bf16_round(torch.randn(B, H, W, C, generator=g))
    params = []
    for k, cin, cout in ((1, C, C // 2), (3, C // 2, C)):
        p = {"kernel": N.bf16_round(torch.randn(k, k, cin, cout, generator=g) * (2.0 / (k * k * cin)) ** 0.5)}
        if bn:
            p.update(gamma=1 + 0.2 * torch.randn(cout, generator=g), beta=0.1 * torch.randn(cout, generator=g),
                     mean=0.1 * torch.randn(cout, generator=g), var=0.5 + torch.rand(cout, generator=g))
        else:
            p["bias"] = 0.1 * torch.randn(cout, generator=g)
        params.append(p)
    l0 = make_conv(params[0], 1, False, act, bn, torch.bfloat16)
    l1 = make_conv(params[1], 3, False, act, bn, torch.bfloat16)
    xin = x.to(DEV).bfloat16()
    mid = l0(xin)
    want = l1(mid, residual=xin)
    w0, sc0, sh0, _ = l0.packed(torch.bfloat16)
    w1, sc1, sh1, _ = l1.packed(torch.bfloat16)
    code = _lib.ACT_MISH if act == "mish" else _lib.ACT_LEAKY
    y = torch.full_like(want, float("nan"))
    _lib.launch("ymx_conv_resblock_fused", xin, _lib.ptr(xin), _lib.ptr(w0), _lib.ptr(sc0), _lib.ptr(sh0), _lib.ptr(w1), _lib.ptr(sc1),
                _lib.ptr(sh1), _lib.ptr(y), B, H, W, C, code, 0, 0)
    assert torch.equal(y, want), float((y.float() - want.float()).abs().max())
    # through the program executor: op 0 marked as read by op 1 alone
    ops = (_lib.ymx_op * 2)()
    y2 = torch.full_like(want, float("nan"))
    mid2 = torch.full_like(mid, float("nan"))
    for o, (layer, src, dst, k, cin, cout, w, sc, sh) in zip(ops, ((l0, xin, mid2, 1, C, C // 2, w0, sc0, sh0),
                                                                     (l1, mid2, y2, 3, C // 2, C, w1, sc1, sh1))):
        o.kind, o.B, o.H, o.W, o.Cin, o.Cout, o.k, o.stride, o.act = 1, B, H, W, cin, cout, k, 1, layer.act_code
        o.dtype = o.out_dtype = o.x_dtype = _lib.BF16
        o.x, o.w, o.y = src.data_ptr(), w.data_ptr(), dst.data_ptr()
        o.scale = None if sc is None else sc.data_ptr()
        o.shift = None if sh is None else sh.data_ptr()
    ops[1].residual = xin.data_ptr()
    ops[0].C1 = 1
    _lib.check(_lib.lib().ymx_run_program(ops, 2, _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(y2, want) and bool(torch.isnan(mid2.float()).all())      # fused: the 32-channel map was never written
    ops[0].C1 = 0
    _lib.check(_lib.lib().ymx_run_program(ops, 2, _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(mid2, mid) and torch.equal(y2, want)
    # a block that is not the fused shape (the shortcut is another tensor) runs as two launches even when marked
    other = torch.zeros_like(xin)
    ops[1].residual = other.data_ptr()
    ops[0].C1 = 1
    mid2.fill_(float("nan"))
    _lib.check(_lib.lib().ymx_run_program(ops, 2, _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(mid2, mid) and torch.equal(y2, l1(mid))
    ref = N.yolo_conv(N.bf16_round(N.yolo_conv(x, params[0], False, act)), params[1], False, act) + x
    assert rel_err(y.float().cpu().numpy(), ref.numpy()) < 2e-2


def test_fused_blocks_beside_other_work():
    """The fused residual-block kernels, repeated while a second stream keeps the GPU busy with other
    convolutions (workgroups then start at irregular times and share L2 and HBM with foreign
    traffic): every repetition gives the bits of the first, quiet run."""
    from yolo_amd import _lib
    g = torch.Generator().manual_seed(77)
    side = torch.cuda.Stream()
    px = torch.randn(24, 26, 26, 256, generator=g).to(DEV).bfloat16()
    pl = make_conv({"kernel": N.bf16_round(torch.randn(3, 3, 256, 512, generator=g) * 0.02), "bias": torch.zeros(512)},
                   3, False, "mish", False, torch.bfloat16)
    for C, S, B in ((64, 208, 16), (128, 104, 32)):
        x = torch.randn(B, S, S, C, generator=g).to(DEV).bfloat16()
        p0 = {"kernel": N.bf16_round(torch.randn(1, 1, C, C // 2, generator=g) * (2.0 / C) ** 0.5), "bias": 0.1 * torch.randn(C // 2, generator=g)}
        p1 = {"kernel": N.bf16_round(torch.randn(3, 3, C // 2, C, generator=g) * (4.0 / (9 * C)) ** 0.5), "bias": 0.1 * torch.randn(C, generator=g)}
        l0 = make_conv(p0, 1, False, "mish", False, torch.bfloat16)
        l1 = make_conv(p1, 3, False, "mish", False, torch.bfloat16)
        want = l1(l0(x), residual=x)
        w0, sc0, sh0, _ = l0.packed(torch.bfloat16)
        w1, sc1, sh1, _ = l1.packed(torch.bfloat16)
        y = torch.empty_like(want)
        bad = torch.zeros((), device=DEV, dtype=torch.int64)
        torch.cuda.synchronize()
        for it in range(40):
            y.fill_(float("nan"))
            _lib.launch("ymx_conv_resblock_fused", x, _lib.ptr(x), _lib.ptr(w0), _lib.ptr(sc0), _lib.ptr(sh0), _lib.ptr(w1), _lib.ptr(sc1),
                        _lib.ptr(sh1), _lib.ptr(y), B, S, S, C, _lib.ACT_MISH, 0, 0)
            bad += (y.view(torch.int16) != want.view(torch.int16)).sum()
            with torch.cuda.stream(side):
                pl(px)
        side.synchronize()
        torch.cuda.synchronize()
        assert int(bad.item()) == 0, (C, int(bad.item()))
