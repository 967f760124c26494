"""CPU restatement (numpy, float32) of the PRN inference glue of the reference's create_pb.py:86-142.

TEST INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu legs, never by the product path.
PARITY UNPINNED: the arithmetic lives in tensorflow==1.15 (tf.image.crop_and_resize, tf.nn.softmax, tf.argmax), which is
absent here; this file restates the published kernels (tensorflow/core/kernels/crop_and_resize_op.cc, bilinear,
extrapolation_value 0) and the reference's own lines, each cited below.
"""
import numpy as np

F = np.float32


def normalize_heatmaps(heatmaps, threshold=0.2):
    """create_pb.py:90-94: M, m = max / min over (h, w) per image and channel; (h - m)/(M - m) * (M > 0.2)."""
    h = np.asarray(heatmaps, F)
    M = h.max(axis=(1, 2), keepdims=True)
    m = h.min(axis=(1, 2), keepdims=True)
    mask = (M > F(threshold)).astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        return ((h - m) / (M - m) * mask).astype(F), m.reshape(h.shape[0], -1), M.reshape(h.shape[0], -1)


def crop_and_resize(image, boxes, box_ind, crop_size):
    """tf.image.crop_and_resize (create_pb.py:106-109), method bilinear, extrapolation_value 0, float32 arithmetic in
    the order of crop_and_resize_op.cc (CPU functor)."""
    image = np.asarray(image, F)
    B, H, W, C = image.shape
    ch, cw = crop_size
    nb = len(boxes)
    out = np.zeros((nb, ch, cw, C), F)
    for n in range(nb):
        b = int(box_ind[n])
        if b < 0 or b >= B:
            continue
        y1, x1, y2, x2 = [F(v) for v in boxes[n]]
        hs = (y2 - y1) * F(H - 1) / F(ch - 1) if ch > 1 else F(0)
        ws = (x2 - x1) * F(W - 1) / F(cw - 1) if cw > 1 else F(0)
        for y in range(ch):
            in_y = y1 * F(H - 1) + F(y) * hs if ch > 1 else F(0.5) * (y1 + y2) * F(H - 1)
            if in_y < 0 or in_y > H - 1:
                continue
            ty, by = int(np.floor(in_y)), int(np.ceil(in_y))
            yl = F(in_y - F(ty))
            xs = np.arange(cw, dtype=F)
            in_x = (x1 * F(W - 1) + xs * ws) if cw > 1 else np.full(cw, F(0.5) * (x1 + x2) * F(W - 1), F)
            ok = ~((in_x < 0) | (in_x > W - 1))
            lx = np.floor(in_x).astype(np.int64)
            rx = np.ceil(in_x).astype(np.int64)
            xl = (in_x - lx.astype(F)).astype(F)[:, None]
            lxc, rxc = np.clip(lx, 0, W - 1), np.clip(rx, 0, W - 1)
            tl, tr = image[b, ty, lxc], image[b, ty, rxc]
            bl, br = image[b, by, lxc], image[b, by, rxc]
            top = (tl + (tr - tl) * xl).astype(F)
            bot = (bl + (br - bl) * xl).astype(F)
            row = (top + (bot - top) * yl).astype(F)
            row[~ok] = 0
            out[n, y] = row
    return out


def decode(logits):
    """create_pb.py:114-138: softmax over the h*w positions per channel; scores = max probability, positions =
    argmax_2d / (h, w). Returns (scores [nb,C] f32, positions [nb,C,2] f32)."""
    z = np.asarray(logits, F)
    nb, h, w, c = z.shape
    flat = z.reshape(nb, h * w, c)
    m = flat.max(axis=1, keepdims=True)
    e = np.exp(flat - m).astype(F)
    prob = e / e.sum(axis=1, keepdims=True, dtype=F)
    scores = prob.max(axis=1)
    arg = prob.argmax(axis=1)
    pos = np.stack([(arg // w).astype(F) / F(h), (arg % w).astype(F) / F(w)], axis=2)
    return scores.astype(F), pos.astype(F)
