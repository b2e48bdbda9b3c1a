"""The first ResidialBlock (64 -> 32 -> 64 + shortcut) as two launches against the fused kernel (tuning aid).
    python tools/resblock_time.py [batch] [side of the map] [channels: 64 or 128]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402
from yolo_amd.layers import YOLOConv            # noqa: E402
from stem_time import timed                     # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else (208 if C == 64 else 104)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, S, S, C, generator=g).cuda().bfloat16()
    for act in ("mish", "leaky"):
        l0 = YOLOConv((1, 1, C // 2), activation=act, dtype=torch.bfloat16)
        l0.build(C, generator=g)
        l1 = YOLOConv((3, 3, C), activation=act, dtype=torch.bfloat16)
        l1.build(C // 2, generator=g)
        l0.cuda(); l1.cuda()
        want = l1(l0(x), residual=x)
        w0, sc0, sh0, _ = l0.packed(torch.bfloat16)
        w1, sc1, sh1, _ = l1.packed(torch.bfloat16)
        y = torch.empty_like(want)
        code = _lib.ACT_MISH if act == "mish" else _lib.ACT_LEAKY

        def fused():
            _lib.launch("ymx_conv_resblock_fused", x, _lib.ptr(x), _lib.ptr(w0), _lib.ptr(sc0), _lib.ptr(sh0), _lib.ptr(w1),
                        _lib.ptr(sc1), _lib.ptr(sh1), _lib.ptr(y), B, S, S, C, code, 0, 0)
        fused()
        torch.cuda.synchronize()
        t0 = timed(lambda: l0(x))
        mid = l0(x)
        t1 = timed(lambda: l1(mid, residual=x))
        t2 = timed(lambda: l1(l0(x), residual=x))
        tf = timed(fused)
        gb = 2.0 * B * S * S * C * 2 / 1e9
        print("C%d B%d %dx%d %-5s: 1x1 %.4f ms, 3x3 %.4f ms, both %.4f ms | fused %.4f ms (%.2fx, %.2f TB/s of the block's own bytes) | %s" % (
            C, B, S, S, act, t0, t1, t2, tf, t2 / tf, gb / tf, "identical" if torch.equal(y, want) else "DIFFERENT"), flush=True)


if __name__ == "__main__":
    main()
