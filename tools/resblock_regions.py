"""Where a workgroup of the fused 128-channel residual-block kernel spends its cycles (tuning aid).

    make -C yolo_amd/csrc -j8 regions RK=conv_resblock128 RTAG=RB128 RN="1 2 3 4 5 6 7"
    python tools/resblock_regions.py [batch] [side]

Every libymx_region_<n>.so times one phase (YMX_REGION_BEGIN / _END in conv_resblock128.hip) with two
reads of the shader clock, summed per wave over the workgroup's tiles; prints the median per tile."""
import ctypes, glob, os, re, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_conv import load

NAMES = {1: "phase A (1x1 into LDS)", 2: "next tile's loads, shortcut loads", 3: "barrier after A", 4: "phase B (144 MFMAs)",
         5: "epilogue math + staging", 6: "barrier after staging", 7: "read-back + stores"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 104
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, S, S, 128, generator=g).cuda().bfloat16()
    w1 = (torch.randn(1, 1, 128, 64, generator=g) * 0.1).cuda()
    w2 = (torch.randn(3, 3, 64, 128, generator=g) * 0.05).cuda()
    b1, b2 = torch.zeros(64).cuda(), torch.zeros(128).cuda()
    y = torch.empty(B, S, S, 128, device="cuda", dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ntiles = B * ((S + 7) // 8) * ((S + 15) // 16)
    per_wg = ntiles / min(ntiles, 256)
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_amd")
    libs = sorted(glob.glob(os.path.join(here, "libymx_region_*.so")), key=lambda p: int(re.findall(r"_(\d+)\.so", p)[0]))
    for path in libs:
        n = int(re.findall(r"_(\d+)\.so", path)[0])
        L = load(path)
        rd = L.ymx_debug_regions
        rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
        wp1 = torch.empty(L.ymx_packed_weight_bytes(1, 128, 64, 1), dtype=torch.uint8, device="cuda")
        wp2 = torch.empty(L.ymx_packed_weight_bytes(3, 64, 128, 1), dtype=torch.uint8, device="cuda")
        assert L.ymx_pack_weights(w1.data_ptr(), wp1.data_ptr(), 1, 128, 64, 1, st) == 0
        assert L.ymx_pack_weights(w2.data_ptr(), wp2.data_ptr(), 3, 64, 128, 1, st) == 0
        for _ in range(5):
            assert L.ymx_conv_resblock_fused(x.data_ptr(), wp1.data_ptr(), None, b1.data_ptr(), wp2.data_ptr(), None, b2.data_ptr(),
                                             y.data_ptr(), B, S, S, 128, 2, 0, 0, st) == 0, L.ymx_last_error()
        torch.cuda.synchronize()
        buf = np.zeros((2048, 8), dtype=np.uint64)
        assert rd(buf.ctypes.data, buf.nbytes) == 0
        s = buf[:256].astype(np.float64) / per_wg
        print("region %d  %-36s  per tile, by wave: %s" % (n, NAMES.get(n, ""), " ".join("%6.0f" % np.median(s[:, w]) for w in range(8))), flush=True)


if __name__ == "__main__":
    main()
