"""Cycles per K tile of each region of a convolution kernel's main loop (tuning aid).

    make -C yolo_amd/csrc -j8 regions RK=conv_igemm_halo RTAG=HALO          # or conv_igemm_dma4 / DMA4
    python tools/region_time.py 64,52,52,512,1024,3,1 [names.txt]

Every libymx_region_<n>.so times region n of the kernel's loop (YMX_REGION_BEGIN / _END in the
source) with two reads of the shader clock; this runs the layer on each and prints the median,
over the workgroups of the second round on, of the per-wave sums divided by the K tiles, for the
first wave group (waves 0-2) and the second (wave 4)."""
import ctypes, glob, os, re, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_conv import load

HALO = {1: "load 1: 16 fragment reads", 2: "load 1: vmcnt + lgkmcnt waits", 3: "barrier after load 1", 4: "mfma 1: 32 MFMAs",
        5: "barrier after mfma 1", 6: "load 2: 8 fragment reads", 7: "load 2: DMA issue", 8: "load 2: waits", 9: "barrier after load 2",
        10: "mfma 2: 32 MFMAs", 11: "barrier after mfma 2", 12: "load 1: DMA issue"}


def main():
    B, H, W, cin, cout, k, stride = (int(v) for v in sys.argv[1].split(","))
    g = torch.Generator().manual_seed(0)
    oh, ow = (H, W) if stride == 1 else ((H + 1 - k) // 2 + 1, (W + 1 - k) // 2 + 1)
    x = torch.randn(B, H, W, cin, generator=g).cuda().bfloat16()
    res = torch.randn(B, oh, ow, cout, generator=g).cuda().bfloat16()
    w = (torch.randn(k, k, cin, cout, generator=g) * (1.0 / (k * k * cin) ** 0.5)).cuda()
    bias = torch.zeros(cout).cuda()
    y = torch.empty(B, oh, ow, cout, device="cuda", dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nk = (k * k * cin + 63) // 64
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_amd")
    libs = sorted(glob.glob(os.path.join(here, "libymx_region_*.so")), key=lambda p: int(re.findall(r"_(\d+)\.so", p)[0]))
    tot = [0.0, 0.0]
    for path in libs:
        n = int(re.findall(r"_(\d+)\.so", path)[0])
        L = load(path)
        rd = L.ymx_debug_regions
        rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
        wp = torch.empty(L.ymx_packed_weight_bytes(k, cin, cout, 1), dtype=torch.uint8, device="cuda")
        assert L.ymx_pack_weights(w.data_ptr(), wp.data_ptr(), k, cin, cout, 1, st) == 0
        for _ in range(6):
            assert L.ymx_conv2d_fused_ex(x.data_ptr(), wp.data_ptr(), None, bias.data_ptr(), res.data_ptr(), y.data_ptr(), B, H, W, cin,
                                         cout, k, stride, 2, 1, 1, 1, 0, 0, 0, st) == 0, L.ymx_last_error()
        torch.cuda.synchronize()
        buf = np.zeros((2048, 8), dtype=np.uint64)
        assert rd(buf.ctypes.data, buf.nbytes) == 0
        s = buf[256:].astype(np.float64) / nk
        s = s[s[:, 0] > 0] if (s[:, 0] > 0).any() else s
        g0, g1 = float(np.median(s[:, 0:3])), float(np.median(s[:, 4]))
        tot[0] += g0; tot[1] += g1
        print("region %2d  %-32s  first group %6.0f   second group %6.0f" % (n, HALO.get(n, ""), g0, g1), flush=True)
    print("sum %53.0f %21.0f   (cycles per K tile; MFMA bound 2048)" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()
