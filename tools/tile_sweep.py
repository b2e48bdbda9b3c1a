"""256- against 192-pixel tiles of the phase-staggered kernel on the detector's wide 1x1 and stride-2
layers, at the batch sizes the engine runs, with the choice the library's cost model makes
(tuning aid).    python tools/tile_sweep.py [batch ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402
from yolo_amd.layers import YOLOConv            # noqa: E402

# input side, Cin, Cout, k, stride
SHAPES = [(52, 1024, 512, 1, 1), (52, 768, 512, 1, 1), (26, 512, 256, 1, 1), (26, 1024, 512, 1, 1), (26, 768, 256, 1, 1),
          (104, 128, 256, 3, 2), (52, 256, 512, 3, 2), (26, 512, 1024, 3, 2), (13, 1024, 512, 1, 1)]


def time_layer(layer, x, res, iters=20):
    for _ in range(3):
        layer(x, residual=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        layer(x, residual=res)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    batches = [int(v) for v in sys.argv[1:]] or [64, 48, 16]
    for B in batches:
        for hw, cin, cout, k, stride in SHAPES:
            opt, fam = b"dma4_pixels", "dma4"
            x = torch.randn(B, hw, hw, cin, generator=g).cuda().bfloat16()
            layer = YOLOConv((k, k, cout), downsample=(stride == 2), activation="mish", dtype=torch.bfloat16)
            layer.build(cin, generator=g)
            layer.cuda()
            oh = hw // stride
            op = _lib.ymx_op()
            op.kind, op.B, op.H, op.W, op.Cin, op.Cout, op.k, op.stride = 1, B, hw, hw, cin, cout, k, stride
            op.dtype = op.x_dtype = op.out_dtype = _lib.BF16
            op.x = op.w = op.y = 256
            op.min_tiles = 80
            _lib.check(L.ymx_set_option(b"dma_min_tiles", 80))
            _lib.check(L.ymx_set_option(opt, 0))
            label = L.ymx_conv_kernel_label(ctypes.byref(op)).decode()
            res = {}
            for px in (256, 192):
                _lib.check(L.ymx_set_option(opt, px))
                if not L.ymx_conv_kernel_label(ctypes.byref(op)).decode().startswith(fam):
                    continue
                res[px] = time_layer(layer, x, None)
            _lib.check(L.ymx_set_option(opt, 0))
            flops = 2.0 * B * oh * oh * k * k * cin * cout
            best = min(res, key=res.get) if res else None
            print("B%-3d %3dx%-3d %4d->%-4d k%d s%d  model: %-8s  " % (B, hw, hw, cin, cout, k, stride, label) +
                  "  ".join("%d: %.4f ms %5.0f TF" % (px, ms, flops / ms / 1e9) for px, ms in res.items()) +
                  ("   best %s  ratio %.3f%s" % (best, res[192] / res[256], "" if (label == "dma4_192") == (best == 192) else "   <-- model differs")
                   if len(res) == 2 else ""), flush=True)
    _lib.check(L.ymx_set_option(b"dma_min_tiles", 160))


if __name__ == "__main__":
    main()
