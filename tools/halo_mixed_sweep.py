"""Mixed tiling of the resident-slab kernel: time against the number of whole rounds of 256-pixel
tiles in front of the 192-pixel tiles (tuning aid).    python tools/halo_mixed_sweep.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402
from yolo_amd.layers import YOLOConv            # noqa: E402
from halo_tile_sweep import time_layer          # noqa: E402

SHAPES = [(52, 512, 1024, 3, 1), (52, 128, 256, 3, 1), (26, 512, 1024, 3, 1), (26, 256, 512, 3, 1), (52, 256, 512, 3, 1),
          (52, 1024, 512, 1, 1), (52, 768, 512, 1, 1), (52, 1024, 256, 1, 1), (26, 1024, 512, 1, 1), (104, 128, 256, 3, 2),
          (52, 256, 512, 3, 2), (26, 512, 1024, 3, 2)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    for hw, cin, cout, k, stride in SHAPES:
        opt = b"halo_pixels" if (k == 3 and stride == 1) else b"dma4_pixels"
        x = torch.randn(B, hw, hw, cin, generator=g).cuda().bfloat16()
        layer = YOLOConv((k, k, cout), downsample=stride == 2, activation="mish", dtype=torch.bfloat16)
        layer.build(cin, generator=g)
        layer.cuda()
        ohw = hw // stride
        tiles = -(-B * ohw * ohw // 256) * -(-cout // 256)
        r256 = -(-tiles // 256)
        flops = 2.0 * B * ohw * ohw * k * k * cin * cout
        out = []
        for px in (256, 192):
            _lib.check(L.ymx_set_option(opt, px))
            ms = time_layer(layer, x)
            out.append("%d: %.4f (%4.0f)" % (px, ms, flops / ms / 1e9))
        _lib.check(L.ymx_set_option(opt, 448))
        for r in range(1, r256):
            _lib.check(L.ymx_set_option(b"halo_big_rounds", r))
            ms = time_layer(layer, x)
            out.append("r%d: %.4f (%4.0f)" % (r, ms, flops / ms / 1e9))
        _lib.check(L.ymx_set_option(b"halo_big_rounds", 0))
        _lib.check(L.ymx_set_option(opt, 0))
        _lib.check(L.ymx_set_option(opt, 0))
        op = _lib.ymx_op()
        op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, B, hw, hw, cin, cout, k, stride
        op.dtype = op.x_dtype = op.out_dtype = _lib.BF16
        op.x = op.w = op.y = 256
        out.append("auto: %s %.4f" % (L.ymx_conv_kernel_label(ctypes.byref(op)).decode(), time_layer(layer, x)))
        print("B%-3d %3dx%-3d k%d s%d %4d->%-4d %5d tiles = %5.2f rounds | " % (B, hw, hw, k, stride, cin, cout, tiles, tiles / 256.0) + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
