"""Independent plain-numpy (explicit loops) restatement of the TensorFlow-1.15 op semantics the
oracle depends on. TEST INFRASTRUCTURE ONLY; used to cross-check oracle/network.py on small cases.

Each function follows TF's documented behaviour for the op the reference calls:
  conv_same       - tf.nn.conv2d / slim.conv2d padding='SAME'   (mobilenet_v1.py:56,73)
  depthwise_same  - tf.nn.depthwise_conv2d 'SAME'                 (mobilenet_v1.py:101)
  nearest_up2     - tf.image.resize_nearest_neighbor              (fpn.py:71)
  bilinear_legacy - tf.image.resize_bilinear (align_corners=False, no half-pixel) (keypoint_subnet.py:86)
All arrays NHWC, weights HWIO.
"""
import math

import numpy as np


def _same_pad(size, k, s):
    out = int(math.ceil(size / s))
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def conv_same(x, w, stride):
    n, h, ww, ci = x.shape
    k, _, _, co = w.shape
    oh, pt = _same_pad(h, k, stride)
    ow, pl = _same_pad(ww, k, stride)
    y = np.zeros((n, oh, ow, co), np.float64)
    for oy in range(oh):
        for ox in range(ow):
            for ky in range(k):
                for kx in range(k):
                    iy, ix = oy * stride + ky - pt, ox * stride + kx - pl
                    if 0 <= iy < h and 0 <= ix < ww:
                        y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[ky, kx].astype(np.float64)
    return y


def depthwise_same(x, w, stride):
    n, h, ww, c = x.shape
    k = w.shape[0]
    oh, pt = _same_pad(h, k, stride)
    ow, pl = _same_pad(ww, k, stride)
    y = np.zeros((n, oh, ow, c), np.float64)
    for oy in range(oh):
        for ox in range(ow):
            for ky in range(k):
                for kx in range(k):
                    iy, ix = oy * stride + ky - pt, ox * stride + kx - pl
                    if 0 <= iy < h and 0 <= ix < ww:
                        y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) * w[ky, kx, :, 0].astype(np.float64)
    return y


def nearest_up2(x):
    n, h, w, c = x.shape
    y = np.zeros((n, 2 * h, 2 * w, c), x.dtype)
    for i in range(2 * h):
        for j in range(2 * w):
            y[:, i, j] = x[:, i // 2, j // 2]
    return y


def bilinear_legacy(x, oh, ow):
    n, h, w, c = x.shape
    y = np.zeros((n, oh, ow, c), np.float64)
    sy, sx = h / oh, w / ow
    for i in range(oh):
        fy = i * sy
        y0 = int(math.floor(fy)); y1 = min(y0 + 1, h - 1); ly = fy - y0
        for j in range(ow):
            fx = j * sx
            x0 = int(math.floor(fx)); x1 = min(x0 + 1, w - 1); lx = fx - x0
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
            y[:, i, j] = top + (bot - top) * ly
    return y
