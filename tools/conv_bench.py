"""Times single convolution layers on the GPU (tuning aid, not part of the product).

    python tools/conv_bench.py                       # the detector's distinct heavy shapes, batch 64
    python tools/conv_bench.py 64,52,52,512,1024,3,1 # B,H,W,Cin,Cout,k,stride

For each shape: 128x128 kernel vs large-tile LDS-DMA kernel, with and without the Mish
epilogue and the shortcut add, on random bf16 data.
"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402

if os.environ.get("YMX_LIB"):                    # an experimental build of the library
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["YMX_LIB"])
from yolo_amd.layers import YOLOConv            # noqa: E402

DEFAULT = [
    (64, 52, 52, 512, 1024, 3, 1), (64, 52, 52, 1024, 512, 1, 1), (64, 52, 52, 128, 256, 3, 1),
    (64, 52, 52, 256, 128, 1, 1), (64, 26, 26, 256, 512, 3, 1), (64, 26, 26, 512, 256, 1, 1),
    (64, 13, 13, 512, 1024, 3, 1), (64, 13, 13, 1024, 512, 1, 1), (64, 104, 104, 64, 128, 3, 1),
    (64, 208, 208, 32, 64, 3, 1), (64, 416, 416, 32, 64, 3, 2), (64, 208, 208, 64, 32, 1, 1),
    (64, 208, 208, 64, 128, 3, 2), (64, 104, 104, 128, 64, 1, 1),
]


def time_layer(layer, x, res, iters=10):
    for _ in range(2):
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
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or DEFAULT
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    for (B, H, W, cin, cout, k, stride) in shapes:
        x = torch.randn(B, H, W, cin, generator=g).cuda().bfloat16()
        oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
        res = torch.randn(B, oh, ow, cout, generator=g).cuda().bfloat16()
        flops = 2.0 * B * oh * ow * k * k * cin * cout
        line = "B%d %dx%d %4d->%4d k%d s%d :" % (B, H, W, cin, cout, k, stride)
        for kern, tiles, variant in (("128", 1 << 40, 1), ("dma1", 0, 1), ("dma2", 0, 2), ("dma2s4", 0, 2), ("dma4", 0, 4), ("dma5", 0, 5), ("slab2", 0, 5)):
            if kern == "slab2" and not (k == 3 and stride == 1 and cin in (32, 64) and cout <= 2 * cin):
                continue
            if kern in ("dma4", "dma5") and cout <= 128:
                continue
            if kern == "dma2s4" and cout > 64:
                continue
            _lib.check(L.ymx_set_option(b"dma_stages", 4 if kern == "dma2s4" else 3))
            if kern == "dma1" and cout <= 128:
                continue
            if kern == "128" and os.environ.get("SKIP128"):
                continue
            _lib.check(L.ymx_set_option(b"dma_min_tiles", tiles))
            _lib.check(L.ymx_set_option(b"dma_variant", variant))
            _lib.check(L.ymx_set_option(b"dma_narrow", 0 if kern == "128" else (2 if kern == "slab2" else 1)))
            for act, use_res in ((None, False), ("mish", False), ("mish", True)):
                layer = YOLOConv((k, k, cout), downsample=(stride == 2), activation=act, dtype=torch.bfloat16)
                layer.build(cin, generator=g)
                with torch.no_grad():
                    layer.kernel.mul_(10.0)
                layer.cuda()
                ms = time_layer(layer, x, res if use_res else None)
                line += "  %s/%s%s %.3fms %.0fTF" % (kern, act or "lin", "+res" if use_res else "", ms, flops / ms / 1e9)
        print(line, flush=True)
    _lib.check(L.ymx_set_option(b"dma_min_tiles", 160))


if __name__ == "__main__":
    main()
