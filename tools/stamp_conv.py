"""In-kernel timeline of one convolution layer (tuning aid, not part of the product).

    make -C yolo_amd/csrc stamps
    python tools/stamp_conv.py dma2 64,208,208,32,64,3,1 [mish] [res]

Loads the diagnostic library (libymx_stamps.so), runs the layer a few times and prints,
per stamped interval, the median / 10th / 90th percentile over workgroups in shader
cycles, the shader clock, and the spread of workgroup start times.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("YMX_STAMP_LIB", "libymx_stamps.so"))
from yolo_amd.layers import YOLOConv            # noqa: E402

NBLK = 16384


def main():
    kern = sys.argv[1]
    B, H, W, cin, cout, k, stride = (int(v) for v in sys.argv[2].split(","))
    act = "mish" if "mish" in sys.argv[3:] else None
    use_res = "res" in sys.argv[3:]
    L = _lib.lib()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    reader = getattr(handle, "ymx_debug_stamps_" + ("dma6" if kern == "slab2" else kern[:4]))
    reader.restype = ctypes.c_int
    reader.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _lib.check(L.ymx_set_option(b"dma_min_tiles", 0))
    _lib.check(L.ymx_set_option(b"dma_variant", 5 if kern == "slab2" else int(kern[3])))
    _lib.check(L.ymx_set_option(b"dma_narrow", 2 if kern == "slab2" else 1))
    _lib.check(L.ymx_set_option(b"dma_stages", 4 if kern.endswith("s4") else 3))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, cin, generator=g).cuda().bfloat16()
    oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
    res = torch.randn(B, oh, ow, cout, generator=g).cuda().bfloat16() if use_res else None
    layer = YOLOConv((k, k, cout), downsample=(stride == 2), activation=act, dtype=torch.bfloat16)
    layer.build(cin, generator=g)
    layer.cuda()
    for _ in range(20):
        layer(x, residual=res)
    torch.cuda.synchronize()
    buf = np.zeros((NBLK, 8), dtype=np.uint64)
    rc = reader(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    used = buf[:, 0] != 0
    s = buf[used].astype(np.int64)
    nst = 6
    clk = (s[:, nst - 1] - s[:, 0]) / np.maximum(s[:, 6] - s[:, 7], 1) * 100e6
    print("%s  B%d %dx%d %d->%d k%d s%d act=%s res=%s: %d workgroups stamped, clock %.2f GHz (median)" % (
        kern, B, H, W, cin, cout, k, stride, act, use_res, s.shape[0], np.median(clk) / 1e9))
    names = ["setup", "prologue", "main loop", "epi: to LDS", "epi: out", "total"]
    for i in range(nst):
        d = (s[:, i + 1] - s[:, i]) if i < nst - 1 else (s[:, nst - 1] - s[:, 0])
        print("  %-12s median %7d  p10 %7d  p90 %7d cycles" % (names[i], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    t0 = s[:, 7] - s[:, 7].min()
    t1 = s[:, 6] - s[:, 7].min()
    print("  kernel span %.1f us (100 MHz clock); workgroup lifetime median %.1f us" % (t1.max() / 100.0, np.median(s[:, 6] - s[:, 7]) / 100.0))


if __name__ == "__main__":
    main()
