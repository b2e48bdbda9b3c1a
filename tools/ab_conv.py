"""A/B of two builds of libymx on single convolution layers, alternating inside one process
(same GPU, same clock state), median and best over rounds.

    python tools/ab_conv.py yolo_amd/libymx.so yolo_amd/libymx_exp.so [B,H,W,Cin,Cout,k,stride ...]

Each layer runs as the detector runs it (Mish + shortcut, bf16), weights packed by each
library itself; the two outputs are compared bit for bit."""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib

SHAPES = [(64, 52, 52, 512, 1024, 3, 1), (64, 52, 52, 128, 256, 3, 1), (64, 26, 26, 512, 1024, 3, 1), (64, 26, 26, 256, 512, 3, 1),
          (64, 13, 13, 512, 1024, 3, 1), (64, 52, 52, 1024, 512, 1, 1), (64, 52, 52, 768, 512, 1, 1), (64, 26, 26, 1024, 512, 1, 1),
          (64, 26, 26, 512, 256, 1, 1), (64, 13, 13, 1024, 512, 1, 1), (64, 52, 52, 256, 128, 1, 1), (64, 104, 104, 128, 256, 3, 2)]


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
    return h


def main():
    libs = [load(sys.argv[1]), load(sys.argv[2])]
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or SHAPES
    rounds, iters = int(os.environ.get("ROUNDS", "5")), int(os.environ.get("ITERS", "10"))
    g = torch.Generator().manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (B, H, W, cin, cout, k, stride) in shapes:
        x = torch.randn(B, H, W, cin, generator=g).cuda().bfloat16()
        oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
        res = torch.randn(B, oh, ow, cout, generator=g).cuda().bfloat16()
        w = (torch.randn(k, k, cin, cout, generator=g) * (1.0 / (k * k * cin) ** 0.5)).cuda()
        bias = (0.05 * torch.randn(cout, generator=g)).cuda()
        ys, wps = [], []
        for L in libs:
            wp = torch.empty(L.ymx_packed_weight_bytes(k, cin, cout, 1), dtype=torch.uint8, device="cuda")
            assert L.ymx_pack_weights(w.data_ptr(), wp.data_ptr(), k, cin, cout, 1, st) == 0
            wps.append(wp)
            ys.append(torch.empty(B, oh, ow, cout, device="cuda", dtype=torch.bfloat16))

        def run(i):
            rc = libs[i].ymx_conv2d_fused_ex(x.data_ptr(), wps[i].data_ptr(), None, bias.data_ptr(), res.data_ptr(), ys[i].data_ptr(),
                                             B, H, W, cin, cout, k, stride, 2, 1, 1, 1, 0, 0, 0, st)
            assert rc == 0, libs[i].ymx_last_error()
        times = [[], []]
        for r in range(rounds):
            for i in (0, 1):
                run(i); run(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / iters)
        same = torch.equal(ys[0], ys[1])
        flops = 2.0 * B * oh * ow * k * k * cin * cout
        ma, mb = statistics.median(times[0]), statistics.median(times[1])
        print("B%d %3dx%-3d %4d->%4d k%d s%d : A %.4f ms %5.0f TF (best %.4f) | B %.4f ms %5.0f TF (best %.4f) | B/A %.3f | outputs %s" %
              (B, H, W, cin, cout, k, stride, ma, flops / ma / 1e9, min(times[0]), mb, flops / mb / 1e9, min(times[1]), mb / ma,
               "identical" if same else "DIFFER"), flush=True)


if __name__ == "__main__":
    main()
