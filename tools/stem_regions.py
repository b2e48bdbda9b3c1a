"""Where a workgroup of the fused first-two-layers kernel spends its cycles (tuning aid).

    make -C yolo_amd/csrc -j8 regions RK=conv_stem RTAG=STEM RN="1 2 3 4 5"
    python tools/stem_regions.py [batch] [side]

Every libymx_region_<n>.so times one phase (YMX_REGION_BEGIN / _END in conv_stem.hip) with two
reads of the shader clock, summed per wave over the workgroup's tiles; prints the median per tile."""
import ctypes, glob, os, re, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_conv import load

NAMES = {1: "phase A (first layer into LDS)", 2: "barrier after A", 3: "phase B (36 MFMAs)", 4: "barrier after B", 5: "epilogue"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, S, S, 3, generator=g).cuda()
    w0 = (torch.randn(27, 32, generator=g) * 0.2).cuda()
    w1 = (torch.randn(3, 3, 32, 64, generator=g) * 0.06).cuda()
    b0, b1 = torch.zeros(32).cuda(), torch.zeros(64).cuda()
    oh = (S - 2) // 2 + 1
    y = torch.empty(B, oh, oh, 64, device="cuda", dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ntiles = B * ((oh + 7) // 8) * ((oh + 15) // 16)
    per_wg = ntiles / min(ntiles, 512)
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_amd")
    libs = sorted(glob.glob(os.path.join(here, "libymx_region_*.so")), key=lambda p: int(re.findall(r"_(\d+)\.so", p)[0]))
    for path in libs:
        n = int(re.findall(r"_(\d+)\.so", path)[0])
        L = load(path)
        rd = L.ymx_debug_regions
        rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
        wp = torch.empty(L.ymx_packed_weight_bytes(3, 32, 64, 1), dtype=torch.uint8, device="cuda")
        assert L.ymx_pack_weights(w1.data_ptr(), wp.data_ptr(), 3, 32, 64, 1, st) == 0
        for _ in range(5):
            assert L.ymx_conv_stem_fused(x.data_ptr(), w0.data_ptr(), None, b0.data_ptr(), wp.data_ptr(), None, b1.data_ptr(),
                                         y.data_ptr(), B, S, S, 2, 0, st) == 0, L.ymx_last_error()
        torch.cuda.synchronize()
        buf = np.zeros((2048, 8), dtype=np.uint64)
        assert rd(buf.ctypes.data, buf.nbytes) == 0
        s = buf[:512].astype(np.float64) / per_wg
        print("region %d  %-32s  per tile, by wave: %s" % (n, NAMES.get(n, ""), " ".join("%6.0f" % np.median(s[:, w]) for w in range(8))), flush=True)


if __name__ == "__main__":
    main()
