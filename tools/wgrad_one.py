"""One weight-gradient shape, a few launches (for rocprofv3 --pmc passes):
    python tools/wgrad_one.py H Cin Cout k stride [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                                                # noqa: E402

H, cin, cout, k, stride = [int(v) for v in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 64
DEV = "cuda:0"
L = _lib.lib()
oh = H if stride == 1 else (H + 1 - k) // 2 + 1
x = torch.randn(B, H, H, cin, device=DEV).bfloat16()
dz = torch.randn(B, oh, oh, cout, device=DEV).bfloat16()
dw = torch.empty(k, k, cin, cout, device=DEV)
nb = L.ymx_conv_wgrad_workspace_bytes(B, H, H, cin, cout, k, stride)
ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
for _ in range(3):
    _lib.launch("ymx_conv_wgrad", x, _lib.ptr(x), 0, _lib.ptr(dz), 0, _lib.ptr(dw), B, H, H, cin, cout, k, stride, _lib.ptr(ws), nb)
torch.cuda.synchronize()
