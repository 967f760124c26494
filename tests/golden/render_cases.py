"""Deterministic INPUTS of the target-heatmap rendering golden cases (seeded numpy RandomState).

Shared by `make_render_goldens.py` (which runs the reference's `get_heatmaps` on them and stores only its OUTPUTS in
`render_goldens.npz`) and by the tests.  Input contract of the reference
(`detector/input_pipeline/heatmap_creation.py:7-19`): keypoints int [P,17,3] as (y, x, visibility) with
y in [0,height-1], x in [0,width-1]; boxes float32 [P,4] (ymin,xmin,ymax,xmax) absolute; width, height ints.
"""
import numpy as np


def _people(rs, P, width, height, p_visible=0.7, box_scale=(0.05, 1.0)):
    kp = np.zeros((P, 17, 3), np.int32)
    kp[:, :, 0] = rs.randint(0, height, size=(P, 17))
    kp[:, :, 1] = rs.randint(0, width, size=(P, 17))
    kp[:, :, 2] = (rs.rand(P, 17) < p_visible).astype(np.int32) * rs.randint(1, 3, size=(P, 17))
    bh = rs.uniform(*box_scale, size=P) * height
    bw = rs.uniform(*box_scale, size=P) * width
    y0 = rs.uniform(0, 1, size=P) * (height - bh)
    x0 = rs.uniform(0, 1, size=P) * (width - bw)
    boxes = np.stack([y0, x0, y0 + bh, x0 + bw], axis=1).astype(np.float32)
    return kp, boxes


def cases():
    """Yield (name, keypoints[P,17,3] int32, boxes[P,4] f32, width, height, downsample)."""
    rs = np.random.RandomState(7)
    kp, bx = _people(rs, 6, 512, 512)
    yield "train512_6p", kp, bx, 512, 512, 4
    kp, bx = _people(rs, 1, 512, 512)
    yield "train512_1p", kp, bx, 512, 512, 4
    kp, bx = _people(rs, 23, 512, 512, box_scale=(0.02, 0.5))
    yield "crowd512_23p", kp, bx, 512, 512, 4
    kp, bx = _people(rs, 5, 640, 384)
    yield "wide640x384", kp, bx, 640, 384, 4
    kp, bx = _people(rs, 4, 384, 640)
    yield "tall384x640", kp, bx, 384, 640, 4
    # sizes that are not multiples of the downsample factor -> ceil
    kp, bx = _people(rs, 3, 333, 251)
    yield "ragged333x251", kp, bx, 333, 251, 4
    kp, bx = _people(rs, 3, 130, 67)
    yield "ragged130x67_ds8", kp, bx, 130, 67, 8
    kp, bx = _people(rs, 2, 96, 96)
    yield "ds1_96", kp, bx, 96, 96, 1

    # sigma clipping: tiny boxes -> sigma 1 (9x9 blob); whole-image 1024 boxes -> sigma 4 (27x27 blob)
    kp, _ = _people(rs, 3, 256, 256, p_visible=1.0)
    tiny = np.array([[10, 10, 20, 20], [0, 0, 1, 1], [50, 60, 50, 60]], np.float32)
    yield "sigma_min", kp, tiny, 256, 256, 4
    kp, _ = _people(rs, 2, 1024, 1024, p_visible=1.0)
    big = np.array([[0, 0, 1024, 1024], [0, 0, 600, 1000]], np.float32)
    yield "sigma_max", kp, big, 1024, 1024, 4
    # sigmas sweeping through the window-size steps (k = ceil(sqrt(-2 s^2 ln .01)))
    P = 40
    kp, _ = _people(rs, P, 512, 512, p_visible=0.3)
    side = np.linspace(100, 620, P).astype(np.float32)
    sweep = np.stack([np.zeros(P), np.zeros(P), side, side], axis=1).astype(np.float32)
    yield "sigma_sweep", kp, sweep, 512, 512, 4

    # blobs hanging over every border and corner
    kp = np.zeros((2, 17, 3), np.int32)
    pts = [(0, 0), (0, 511), (511, 0), (511, 511), (0, 256), (256, 0), (511, 256), (256, 511),
           (3, 3), (508, 508), (2, 509), (509, 2), (255, 255), (256, 256), (1, 1), (510, 510), (7, 500)]
    for j, (y, x) in enumerate(pts):
        kp[0, j] = (y, x, 2)
        kp[1, j] = (511 - y, x, 1)
    bx = np.array([[0, 0, 512, 512], [100, 100, 300, 250]], np.float32)
    yield "borders", kp, bx, 512, 512, 4

    # overlapping blobs of different sigma on the same part -> max
    kp = np.zeros((3, 17, 3), np.int32)
    for j in range(17):
        kp[0, j] = (200 + j, 200, 1)
        kp[1, j] = (204 + j, 206 + j, 1)
        kp[2, j] = (200 + j, 200, 1)
    bx = np.array([[0, 0, 150, 150], [0, 0, 400, 400], [0, 0, 512, 512]], np.float32)
    yield "overlap_max", kp, bx, 512, 512, 4

    # nothing visible at all; one part visible only
    kp, bx = _people(rs, 4, 256, 256, p_visible=0.0)
    yield "none_visible", kp, bx, 256, 256, 4
    kp2 = kp.copy()
    kp2[2, 11, 2] = 1
    yield "one_visible", kp2, bx, 256, 256, 4

    # rounding of the centre: half-way cases of round() (banker's rounding on float32)
    kp = np.zeros((1, 17, 3), np.int32)
    for j in range(17):
        kp[0, j] = (2 + 4 * j, 6 + 4 * j, 1)
    yield "round_half", kp, np.array([[0, 0, 129, 129]], np.float32), 129, 129, 4
    kp, bx = _people(rs, 8, 97, 61, p_visible=0.9)
    yield "round_odd_ds2", kp, bx, 97, 61, 2
