"""ymx_nms_combined_static (the entry point that allocates nothing) against
ymx_nms_combined, both called through the C ABI on identical device inputs.

Bar: every output bit-exact (boxes, scores, classes, valid counts, keep indices).  Two
calls in a row on the same stream, the second with a smaller input, so that whatever the
first call left in the library's built-in scratch buffer is still there when the second
one runs; then a size beyond the buffer must be rejected without touching the device.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B, M, C, seed):
    rng = np.random.default_rng(seed)
    cap = (M + 3) // 4 * 4
    ctr = rng.random((B, cap, 2), dtype=np.float32)
    wh = np.exp(rng.normal(-2.5, 0.5, (B, cap, 2))).astype(np.float32)
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], -1).astype(np.float32)
    scores = (rng.random((B, C, cap), dtype=np.float32) ** 4).astype(np.float32)
    counts = rng.integers(M // 2, M + 1, size=B).astype(np.int32)
    src = rng.permutation(4 * cap)[:cap].astype(np.int32)[None].repeat(B, 0)
    return [torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in (boxes, scores, counts, src)], cap


def _outputs(B, T):
    f32, i32 = torch.float32, torch.int32
    return [torch.full((B, T, 4), -7.0, device=DEV, dtype=f32), torch.full((B, T), -7.0, device=DEV, dtype=f32),
            torch.full((B, T), -7.0, device=DEV, dtype=f32), torch.full((B,), -7, device=DEV, dtype=i32),
            torch.full((B, T), -7, device=DEV, dtype=i32)]


def _both(B, M, C, per_class, total, thr, seed):
    from yolo_amd import _lib
    L = _lib.lib()
    (boxes, scores, counts, src), cap = _inputs(B, M, C, seed)
    stream = _lib.stream(boxes)
    want = _outputs(B, total)
    ws_bytes = L.ymx_nms_workspace_bytes(B, C, per_class)
    ws = torch.empty((ws_bytes,), device=DEV, dtype=torch.uint8)
    _lib.check(L.ymx_nms_combined(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(counts), _lib.ptr(src), B, cap, C,
                                  per_class, total, 0.45, thr, *[_lib.ptr(t) for t in want], _lib.ptr(ws), ws_bytes, stream))
    got = _outputs(B, total)
    _lib.check(L.ymx_nms_combined_static(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(counts), _lib.ptr(src), B, cap, C,
                                         per_class, total, 0.45, thr, *[_lib.ptr(t) for t in got], stream))
    torch.cuda.synchronize()
    return got, want


def test_static_scratch_nms_equals_workspace_nms_twice():
    # first call: large (80 classes, 6000 candidates, batch 8); second: smaller in every size
    for (B, M, C, pc, T, thr, seed) in ((8, 6000, 80, 32, 32, 0.05, 1), (3, 700, 5, 8, 16, 0.3, 2)):
        got, want = _both(B, M, C, pc, T, thr, seed)
        assert int(want[3].min()) > 0
        for g, w in zip(got, want):
            assert torch.equal(g, w)
    # and once more at the first size: stale survivors of the small call must not leak
    got, want = _both(8, 6000, 80, 32, 32, 0.05, 3)
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_static_scratch_size_limit_and_arguments():
    from yolo_amd import _lib
    L = _lib.lib()
    one = ctypes.c_void_p(256)
    limit = 4 << 20
    assert L.ymx_nms_workspace_bytes(256, 80, 32) > limit
    rc = L.ymx_nms_combined_static(one, one, None, None, 256, 128, 80, 32, 32, 0.5, 0.1, one, one, one, one, one, None)
    assert rc == -2 and b"scratch" in L.ymx_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    rc = L.ymx_nms_combined_static(one, one, None, None, 0, 128, 80, 32, 32, 0.5, 0.1, one, one, one, one, one, None)
    assert rc == -1
    # within the limit the ordinary checks apply (here: cap not a multiple of 4)
    rc = L.ymx_nms_combined_static(one, one, None, None, 2, 126, 80, 32, 32, 0.5, 0.1, one, one, one, one, one, None)
    assert rc == -1 and b"multiple of 4" in L.ymx_last_error()
