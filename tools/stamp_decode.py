"""In-kernel timelines of the decode+filter kernel and of the per-class NMS kernel on the
NMS-stress heads (tuning aid).

    make -C yolo_amd/csrc stamps && python tools/stamp_decode.py [batch] [detector]

Prints, per interval of a workgroup's life, the median / p10 / p90 over the stamped
workgroups in microseconds (100 MHz wall clock).
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd import _lib                      # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libymx_stamps.so")
import yolo_amd as Y                            # noqa: E402
from yolo_amd import utils as U                 # noqa: E402

NBLK = 16384


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    stress = len(sys.argv) <= 2 or sys.argv[2] != "detector"
    C = 80
    g = torch.Generator().manual_seed(10)
    heads = []
    for S in ((160, 80, 40) if stress else (52, 26, 13)):
        h = torch.randn(B, S, S, 3, 5 + C, generator=g) * 2
        if stress:
            h[..., 4] += 6.0
        heads.append(h.cuda())
    thr, size = (0.001, 1280.0) if stress else (0.25, 416.0)
    cfg = Y.config
    handle = ctypes.CDLL(_lib.LIB_PATH)
    reader = handle.ymx_debug_stamps_decode
    reader.restype = ctypes.c_int
    reader.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    for _ in range(3):
        r = U.decode_heads(heads, C, cfg["strides"], cfg["xyscale"], U.default_anchors(), filter=True,
                           score_threshold=thr, image_size=size)
    torch.cuda.synchronize()
    buf = np.zeros((NBLK, 8), dtype=np.uint64)
    assert reader(buf.ctypes.data, buf.nbytes) == 0
    s = buf[buf[:, 0] != 0].astype(np.int64)
    names = ["wait for rows", "compute + ballot", "publish, next rows, look-back", "stores", "iteration"]
    print("mean survivors per image %.0f" % float(r["counts"].float().mean()))
    print("%d workgroups stamped at their 4th segment" % s.shape[0])
    for i in range(5 if s.shape[0] else 0):
        d = (s[:, i + 1] - s[:, i]) if i < 4 else (s[:, 4] - s[:, 0])
        d = d / 100.0
        print("  %-18s median %7.2f  p10 %7.2f  p90 %7.2f us" % (names[i], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    if s.shape[0]:
        print("  span of the stamped iterations %.1f us" % ((s[:, 4].max() - s[:, 0].min()) / 100.0))
    # per-class NMS kernel on the decoded result
    from yolo_amd import utils as UU
    for _ in range(2):
        UU.nms_class_major(r["boxes"], r["scores_cm"], r["counts"], r["src_idx"], 32, 32, 0.45, thr)
    torch.cuda.synchronize()
    reader = handle.ymx_debug_stamps_nms
    reader.restype = ctypes.c_int
    reader.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    buf = np.zeros((NBLK, 8), dtype=np.uint64)
    assert reader(buf.ctypes.data, buf.nbytes) == 0
    s = buf[buf[:, 0] != 0].astype(np.int64)
    names = ["sampled histogram pass", "floor + key pass", "sort", "greedy", "total"]
    print("nms_class_kernel: %d workgroups" % s.shape[0])
    for i in range(5):
        d = (s[:, i + 1] - s[:, i]) if i < 4 else (s[:, 4] - s[:, 0])
        d = d / 100.0
        print("  %-24s median %7.2f  p10 %7.2f  p90 %7.2f us" % (names[i], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    print("  kernel span %.1f us" % ((s[:, 4].max() - s[:, 0].min()) / 100.0))
    # inside "sort" (last selection round of the workgroup): selection histogram + compaction, the sort, the greedy pass
    ok = s[(s[:, 5] > s[:, 2]) & (s[:, 7] >= s[:, 6])]
    if ok.shape[0]:
        for name, i0, i1 in (("  select (min/max, histogram, cut, compact)", 2, 5), ("  sort of the selection", 5, 6), ("  greedy pass", 6, 7), ("  after it", 7, 3)):
            d = (ok[:, i1] - ok[:, i0]) / 100.0
            print("  %-44s median %7.2f  p10 %7.2f  p90 %7.2f us" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))


if __name__ == "__main__":
    main()
