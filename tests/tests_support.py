"""Input builders shared by the CPU and GPU test files."""
import numpy as np

from oracle import postproc_np as P


def loss_case(B, S, C, scale, seed, n_gt=3):
    """Seeded raw map, label tensor and ground-truth boxes in the target layout of
    /root/reference/losses.py:58-62: a few responsible (cell, anchor) pairs per image with
    smoothed one-hot classes, their boxes also listed among the 128 ground-truth slots."""
    rng = np.random.default_rng(seed)
    D = 5 + C
    conv = (rng.standard_normal((B, S, S, 3, D)) * 1.5).astype(np.float32)
    label = np.zeros((B, S, S, 3, D), np.float32)
    bboxes = np.zeros((B, 128, 4), np.float32)
    stride = P.STRIDES[scale]
    for b in range(B):
        for j in range(n_gt):
            r, c, a = rng.integers(0, S), rng.integers(0, S), rng.integers(0, 3)
            w, h = P.ANCHORS[scale][a] * rng.uniform(0.6, 1.6, 2)
            box = np.array([(c + rng.uniform(0.1, 0.9)) * stride, (r + rng.uniform(0.1, 0.9)) * stride, w, h], np.float32)
            onehot = np.full(C, 0.01 / C, np.float32)
            onehot[rng.integers(0, C)] += 0.99
            label[b, r, c, a, :4] = box
            label[b, r, c, a, 4] = 1.0
            label[b, r, c, a, 5:] = onehot
            bboxes[b, j] = box
    return conv, label, bboxes
