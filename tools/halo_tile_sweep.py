"""256- against 192-pixel tiles (and both forms in one launch) of the resident-slab kernel on the detector's wide 3x3 layers, at the
batch sizes the engine runs (whole batch and the two concurrent sub-batches), with the choice the
library's cost model makes (tuning aid).    python tools/halo_tile_sweep.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402
from yolo_amd.layers import YOLOConv            # noqa: E402

SHAPES = [(52, 128, 256), (52, 256, 512), (52, 512, 1024), (26, 256, 512), (26, 512, 1024), (13, 512, 1024)]


def time_layer(layer, x, iters=20):
    for _ in range(3):
        layer(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        layer(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    for B in (64, 48, 16):
        for hw, cin, cout in SHAPES:
            x = torch.randn(B, hw, hw, cin, generator=g).cuda().bfloat16()
            layer = YOLOConv((3, 3, cout), activation="mish", dtype=torch.bfloat16)
            layer.build(cin, generator=g)
            layer.cuda()
            op = _lib.ymx_op()
            op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, B, hw, hw, cin, cout, 3, 1
            op.dtype = op.x_dtype = op.out_dtype = _lib.BF16
            op.x = op.w = op.y = 256
            op.min_tiles = 80
            _lib.check(L.ymx_set_option(b"dma_min_tiles", 80))
            _lib.check(L.ymx_set_option(b"halo_pixels", 0))
            label = L.ymx_conv_kernel_label(ctypes.byref(op)).decode()
            res = {}
            for px in (256, 192, 448):               # 448 = both forms in one launch
                _lib.check(L.ymx_set_option(b"halo_pixels", px))
                got = L.ymx_conv_kernel_label(ctypes.byref(op)).decode()
                if not got.startswith("halo") or (px == 448 and got != "halo_mixed"):
                    continue
                res[px] = time_layer(layer, x)
            _lib.check(L.ymx_set_option(b"halo_pixels", 0))
            flops = 2.0 * B * hw * hw * 9 * cin * cout
            best = min(res, key=res.get) if res else None
            print("B%-3d %2dx%-2d %4d->%-4d  model: %-8s  " % (B, hw, hw, cin, cout, label) +
                  "  ".join("%d: %.4f ms %5.0f TF" % (px, ms, flops / ms / 1e9) for px, ms in res.items()) +
                  ("   best %s%s" % (best, "" if {"halo": 256, "halo192": 192, "halo_mixed": 448}.get(label) == best else "   <-- model differs") if res else ""), flush=True)


if __name__ == "__main__":
    main()
