"""The backbone's first two layers as two launches against the fused kernel (tuning aid).
    python tools/stem_time.py [batch] [side]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402
from yolo_amd.layers import YOLOConv            # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, S, S, 3, generator=g).cuda()
    for act in ("mish", "leaky"):
        l0 = YOLOConv((3, 3, 32), activation=act, dtype=torch.bfloat16)
        l0.build(3, generator=g)
        l1 = YOLOConv((3, 3, 64), downsample=True, activation=act, dtype=torch.bfloat16)
        l1.build(32, generator=g)
        l0.cuda(); l1.cuda()
        want = l1(l0(x))
        w0, sc0, sh0, _ = l0.packed(torch.bfloat16)
        w1, sc1, sh1, _ = l1.packed(torch.bfloat16)
        y = torch.empty_like(want)
        code = _lib.ACT_MISH if act == "mish" else _lib.ACT_LEAKY

        def fused():
            _lib.launch("ymx_conv_stem_fused", x, _lib.ptr(x), _lib.ptr(w0), _lib.ptr(sc0), _lib.ptr(sh0), _lib.ptr(w1),
                        _lib.ptr(sc1), _lib.ptr(sh1), _lib.ptr(y), B, S, S, code, 0)
        fused()
        torch.cuda.synchronize()
        t0 = timed(lambda: l0(x))
        mid = l0(x)
        t1 = timed(lambda: l1(mid))
        t2 = timed(lambda: l1(l0(x)))
        tf = timed(fused)
        print("B%d %dx%d %-5s: layer 0 %.4f ms, layer 1 %.4f ms, both %.4f ms | fused %.4f ms (%.2fx) | %s" % (
            B, S, S, act, t0, t1, t2, tf, t2 / tf, "identical" if torch.equal(y, want) else "DIFFERENT"), flush=True)


if __name__ == "__main__":
    main()
