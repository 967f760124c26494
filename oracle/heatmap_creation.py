"""Oracle for target-heatmap rendering (TEST INFRASTRUCTURE, not shipped code).

CPU/numpy restatement of `get_heatmaps`, `get_kernel`, `create_heatmap`
(reference detector/input_pipeline/heatmap_creation.py:6-118).  Pinned: bit-identical to the outputs of the imported
reference on tests/golden/render_goldens.npz (generated here with numpy 2.2 / scipy 1.15, i.e. NEP-50 promotion for
the float32-scalar arithmetic: `sigma` and `2*sigma*sigma` stay float32, the window itself is float64).

Restated as a closed form per output pixel instead of the reference's pad / paste / crop / stack / max sequence:
    out[y, x, j] = max(0, max over visible persons p of float32(g_p[|y - cy|] * g_p[|x - cx|]))   for |dy|,|dx| <= k_p
"""
import math

import numpy as np

NUM_KEYPOINTS = 17
MIN_SIGMA, MAX_SIGMA = np.float32(1.0), np.float32(4.0)     # heatmap_creation.py:22


def person_sigmas(boxes):
    """heatmap_creation.py:30-37: sigma = clip(0.007 * sqrt(box area), 1, 4), float32 arithmetic."""
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    return np.clip(np.sqrt(area) * np.float32(0.007), MIN_SIGMA, MAX_SIGMA)


def half_window(sigma):
    """heatmap_creation.py:78-81 (get_kernel): k = ceil(sqrt(-2 sigma^2 ln 0.01)); sigma^2 in float32, the rest float64.
    Returns (k, g) with g[d] = exp(-d^2 / float64(float32(2 sigma sigma))), d = 0..k (scipy windows.gaussian, sym)."""
    s = np.float32(sigma)
    k = int(math.ceil(math.sqrt(float(np.float32(-2.0) * (s * s)) * math.log(0.01))))
    sig2 = float(np.float32(2.0) * s * s)
    d = np.arange(0, k + 1, dtype=np.float64)
    return k, np.exp(-d ** 2 / sig2)


def centres(keypoints, width, height, w, h):
    """heatmap_creation.py:23-24,57,104-107: normalise by (size-1) in float32, scale by (out-1) in float32,
    round half to even."""
    kp = np.asarray(keypoints)
    yx = kp[:, :, :2].astype(np.float32) / np.array([height - 1.0, width - 1.0], np.float32)
    cy = np.rint(yx[:, :, 0] * np.float32(h - 1)).astype(np.int64)
    cx = np.rint(yx[:, :, 1] * np.float32(w - 1)).astype(np.int64)
    return cy, cx


def get_heatmaps(keypoints, boxes, width, height, downsample):
    """heatmap_creation.py:6-72.  keypoints int [P,17,3] (y,x,vis), boxes f32 [P,4] -> float32 [h,w,17]."""
    h = math.ceil(height / downsample)                                 # :27-28
    w = math.ceil(width / downsample)
    out = np.zeros((h, w, NUM_KEYPOINTS), np.float32)
    kp = np.asarray(keypoints)
    if kp.shape[0] == 0:
        return out
    sig = person_sigmas(boxes)
    cy, cx = centres(kp, width, height, w, h)
    ys, xs = np.arange(h)[:, None], np.arange(w)[None, :]
    for p in range(kp.shape[0]):
        k, g = half_window(sig[p])
        for j in range(NUM_KEYPOINTS):
            if kp[p, j, 2] <= 0:                                       # :44
                continue
            y0, y1 = max(cy[p, j] - k, 0), min(cy[p, j] + k, h - 1)
            x0, x1 = max(cx[p, j] - k, 0), min(cx[p, j] + k, w - 1)
            if y0 > y1 or x0 > x1:
                continue
            gy = g[np.abs(ys[y0:y1 + 1] - cy[p, j])]
            gx = g[np.abs(xs[:, x0:x1 + 1] - cx[p, j])]
            blob = (gy * gx).astype(np.float32)                        # np.outer(...).astype(float32), :85
            np.maximum(out[y0:y1 + 1, x0:x1 + 1, j], blob, out=out[y0:y1 + 1, x0:x1 + 1, j])   # :69
    return out
