"""Race screen of the LDS-DMA convolution kernels (tuning/verification aid, not product).

Every kernel variant is run many times on one stream while a second stream keeps the GPU
busy with other convolutions (so workgroups start at irregular times and share L2 and HBM
with foreign traffic), and every output is compared bit for bit with the register-staged
128x128 kernel's.  A synchronisation slip between an LDS-DMA write and a fragment read shows
up here as an occasional mismatch.

    python tools/race_screen.py [iterations]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402
from yolo_amd.layers import YOLOConv            # noqa: E402

SHAPES = [
    (64, 52, 52, 512, 1024, 3, 1), (64, 26, 26, 256, 512, 3, 1), (64, 13, 13, 512, 1024, 3, 1),
    (16, 26, 26, 256, 512, 3, 1), (48, 13, 13, 512, 1024, 3, 1), (64, 52, 52, 128, 256, 3, 1),
    (64, 52, 52, 1024, 512, 1, 1), (64, 26, 26, 512, 256, 1, 1), (16, 52, 52, 768, 256, 1, 1),
    (64, 104, 104, 128, 256, 3, 2), (64, 26, 26, 512, 1024, 3, 2),
    (64, 208, 208, 32, 64, 3, 1), (64, 104, 104, 64, 128, 3, 1), (64, 208, 208, 64, 32, 1, 1),
    (64, 104, 104, 128, 64, 1, 1), (64, 52, 52, 256, 128, 1, 1), (64, 416, 416, 32, 64, 3, 2),
    (16, 208, 208, 32, 64, 3, 1), (48, 104, 104, 64, 128, 3, 1),
]


def set_kernel(L, tiles, variant, narrow, stages=4):
    _lib.check(L.ymx_set_option(b"dma_min_tiles", tiles))
    _lib.check(L.ymx_set_option(b"dma_variant", variant))
    _lib.check(L.ymx_set_option(b"dma_narrow", narrow))
    _lib.check(L.ymx_set_option(b"dma_stages", stages))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    side = torch.cuda.Stream()
    # the perturber: a mid-sized 3x3 layer and a narrow layer, launched on the side stream
    px = torch.randn(24, 26, 26, 256, generator=g).cuda().bfloat16()
    pl = YOLOConv((3, 3, 512), activation="mish", dtype=torch.bfloat16)
    pl.build(256, generator=g)
    pl.cuda()
    qx = torch.randn(8, 104, 104, 128, generator=g).cuda().bfloat16()
    ql = YOLOConv((1, 1, 64), activation="leaky", dtype=torch.bfloat16)
    ql.build(128, generator=g)
    ql.cuda()
    total_bad = 0
    for (B, H, W, cin, cout, k, stride) in SHAPES:
        x = torch.randn(B, H, W, cin, generator=g).cuda().bfloat16()
        oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
        res = torch.randn(B, oh, ow, cout, generator=g).cuda().bfloat16()
        layer = YOLOConv((k, k, cout), downsample=(stride == 2), activation="mish", dtype=torch.bfloat16)
        layer.build(cin, generator=g)
        with torch.no_grad():
            layer.kernel.mul_(10.0)
        layer.cuda()
        set_kernel(L, 1 << 40, 1, 0)
        ref = layer(x, residual=res).view(torch.int16).clone()
        torch.cuda.synchronize()
        if cout > 128:
            variants = [("dma1", 1, 1, 4), ("dma2", 2, 1, 4), ("dma4", 4, 1, 4), ("auto", 5, 2, 4)]
        else:
            variants = [("dma2s3", 5, 1, 3), ("dma2s4", 5, 1, 4), ("auto", 5, 2, 4)]
        line = "B%d %dx%d %4d->%4d k%d s%d :" % (B, H, W, cin, cout, k, stride)
        for name, variant, narrow, stages in variants:
            bad = torch.zeros((), device="cuda", dtype=torch.int64)
            main_stream = torch.cuda.current_stream()
            for it in range(iters):
                set_kernel(L, 0, variant, narrow, stages)
                y = layer(x, residual=res)
                bad += (y.view(torch.int16) != ref).sum()
                with torch.cuda.stream(side):
                    set_kernel(L, 0, 5, 2, 4)
                    (pl if it & 1 else ql)((px if it & 1 else qx))
            side.synchronize()
            torch.cuda.synchronize()
            nbad = int(bad.item())
            total_bad += nbad
            line += "  %s %s" % (name, "ok" if nbad == 0 else "MISMATCH(%d)" % nbad)
        print(line, flush=True)
    set_kernel(L, 160, 5, 2, 4)
    print("race screen: %s (%d iterations per kernel)" % ("clean" if total_bad == 0 else "%d mismatching elements" % total_bad, iters), flush=True)
    sys.exit(0 if total_bad == 0 else 1)


if __name__ == "__main__":
    main()
