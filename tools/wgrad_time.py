"""Times ymx_conv_wgrad alone on the detector's weight-gradient shapes (tuning aid):
    python tools/wgrad_time.py [batch]          env WGRAD = option wgrad_variant"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                                                # noqa: E402

DEV = "cuda:0"
SHAPES = [  # H, Cin, Cout, k, stride
    (52, 512, 1024, 3, 1), (26, 512, 1024, 3, 1), (13, 512, 1024, 3, 1), (26, 256, 512, 3, 1), (52, 128, 256, 3, 1),
    (104, 64, 128, 3, 1), (208, 32, 64, 3, 1), (416, 8, 32, 3, 1),
    (416, 32, 64, 3, 2), (208, 64, 128, 3, 2), (104, 128, 256, 3, 2), (52, 256, 512, 3, 2), (26, 512, 1024, 3, 2),
    (13, 1024, 512, 1, 1), (52, 256, 128, 1, 1), (208, 64, 32, 1, 1),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L = _lib.lib()
    if os.environ.get("WGRAD"):
        _lib.check(L.ymx_set_option(b"wgrad_variant", int(os.environ["WGRAD"])))
    for H, cin, cout, k, stride in SHAPES:
        oh = H if stride == 1 else (H + 1 - k) // 2 + 1
        x = torch.randn(B, H, H, cin, device=DEV).bfloat16()
        dz = torch.randn(B, oh, oh, cout, device=DEV).bfloat16()
        dw = torch.empty(k, k, cin, cout, device=DEV)
        nb = L.ymx_conv_wgrad_workspace_bytes(B, H, H, cin, cout, k, stride)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)

        def run():
            _lib.launch("ymx_conv_wgrad", x, _lib.ptr(x), 0, _lib.ptr(dz), 0, _lib.ptr(dw), B, H, H, cin, cout, k, stride, _lib.ptr(ws), nb)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        a.record()
        for _ in range(reps):
            run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        fl = 2.0 * B * oh * oh * cin * cout * k * k
        print("wgrad %4d^2 %4d->%4d k%d s%d: %8.3f ms  %7.1f TFLOP/s" % (H, cin, cout, k, stride, ms, fl / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
