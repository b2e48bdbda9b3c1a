"""Which co-running kernel makes the decode kernel's packed-FMA box corners come out wrong?

Stream A repeats the detector's decode+filter launch on fixed head maps and counts, on the
device, the launches whose boxes differ from a launch that ran alone.  Stream B meanwhile loops
one co-runner: nothing, convolutions routed to the different kernel families of libymx, the
upsample kernel, plain torch elementwise work, a torch (hipBLASLt) matmul.
    YMX_LIBRARY=yolo_amd/libymx_packed_decode.so python tools/corner_bisect.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import Detector
from yolo_amd.layers import YOLOConv, upsample2x_concat

DEV = "cuda:0"
N = int(os.environ.get("N", "300"))
SIZE, B = 160, 3
bf = torch.bfloat16
x = torch.rand(B, SIZE, SIZE, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
det = Detector(num_classes=80, dtype=bf, seed=5, split="off").to(DEV)
det.calibrate_(x[:2].contiguous())
det(x)
torch.cuda.synchronize()
part = det._plan(B, SIZE, x.dtype, x.device).parts[0]
part.boxes.zero_()
det._decode_part(part, SIZE)
torch.cuda.synchronize()
ref = part.boxes.clone().view(torch.int32)
refc = part.counts.clone()
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()


def conv(k, cin, cout, hw, b, down=False, act="mish"):
    m = YOLOConv((k, k, cout), downsample=down, activation=act, dtype=bf, in_channels=cin).to(DEV)
    with torch.no_grad():
        m.kernel.mul_(30.0)
    t = torch.randn(b, hw, hw, cin, device=DEV).to(bf)
    m(t)
    return lambda: m(t)


g = torch.Generator(device=DEV).manual_seed(3)
big = torch.randn(1 << 24, device=DEV)
ma = torch.randn(2048, 2048, device=DEV).to(bf)
up_x = torch.randn(4, 10, 10, 512, device=DEV).to(bf)
up_s = torch.randn(4, 20, 20, 256, device=DEV).to(bf)
co = [
    ("nothing", None),
    ("torch elementwise (16M floats)", lambda: big.mul_(1.0000001)),
    ("torch bf16 matmul 2048^3", lambda: torch.matmul(ma, ma)),
    ("first layer 3->32 @160 B4", conv(3, 3, 32, 160, 4)),
    ("3x3 s2 32->64 @160 B4", conv(3, 32, 64, 160, 4, down=True)),
    ("1x1 64->32 @80 B4", conv(1, 64, 32, 80, 4)),
    ("3x3 32->64 @80 B4", conv(3, 32, 64, 80, 4)),
    ("3x3 128->256 @20 B4", conv(3, 128, 256, 20, 4)),
    ("1x1 256->128 @20 B4", conv(1, 256, 128, 20, 4)),
    ("3x3 512->1024 @5 B4", conv(3, 512, 1024, 5, 4)),
    ("1x1 1024->512 @5 B4", conv(1, 1024, 512, 5, 4)),
    ("3x3 512->1024 @20 B4", conv(3, 512, 1024, 20, 4)),
    ("1x1 1024->512 @20 B4", conv(1, 1024, 512, 20, 4)),
    ("3x3 512->1024 @52 B8 (resident slab)", conv(3, 512, 1024, 52, 8)),
    ("1x1 1024->512 @52 B8 (dma4)", conv(1, 1024, 512, 52, 8)),
    ("upsample+concat", lambda: upsample2x_concat(up_s, up_x)),
]
torch.cuda.synchronize()
for name, fn in co:
    bad = torch.zeros((), device=DEV, dtype=torch.int32)
    for i in range(N):
        if fn is not None:
            with torch.cuda.stream(Bs):
                for _ in range(3):
                    fn()
        with torch.cuda.stream(A):
            det._decode_part(part, SIZE)
            bad += ((part.boxes.view(torch.int32) != ref).any() | (part.counts != refc).any()).int()
    torch.cuda.synchronize()
    print("%-40s bad decode launches %4d of %d" % (name, int(bad), N), flush=True)
print("done")
