"""Deterministic INPUTS of the decode golden cases (seeded numpy RandomState).

Shared by `make_decode_goldens.py` (which runs the reference on them and stores
only the reference's OUTPUTS in `decode_goldens.npz`) and by the tests.
"""
import numpy as np


def cases():
    """Yield (name, heatmaps[h,w,17], box[4], threshold)."""
    rs = np.random.RandomState(0)
    hm = rs.rand(128, 128, 17).astype(np.float32)
    yield "rand128_fullbox", hm, np.array([0, 0, 512, 512]), 0.2
    yield "rand128_floatbox", hm, np.array([10.5, 20.25, 300.75, 410.5]), 0.2
    yield "rand128_f32box", hm, np.array([10.5, 20.25, 300.75, 410.5], dtype=np.float32), 0.2
    yield "rand128_highthr", hm, np.array([0, 0, 512, 512]), 0.99995

    # two equal peaks -> first in row-major order wins
    t = np.zeros((4, 5, 17), np.float32)
    t[1, 2, :] = 0.9
    t[3, 4, :] = 0.9
    yield "tie_4x5", t, np.array([0, 0, 4, 5]), 0.5

    # max == threshold -> strict '>' means not emitted
    e = np.zeros((8, 8, 17), np.float32)
    e[3, 3, :] = np.float32(0.2)
    yield "eq_threshold", e, np.array([0, 0, 64, 64]), 0.2
    e2 = e.copy()
    e2[3, 3, 5] = np.nextafter(np.float32(0.2), np.float32(1.0))
    yield "just_above_threshold", e2, np.array([0, 0, 64, 64]), 0.2

    # NaN in one channel -> that channel skipped
    n = rs.rand(16, 16, 17).astype(np.float32)
    n[5, 7, 3] = np.nan
    n[0, 0, 9] = -np.nan
    yield "nan_channel", n, np.array([0, 0, 160, 160]), 0.2

    # truncation: 8x6 map, peak (3,2), box 75x61
    p = np.zeros((8, 6, 17), np.float32)
    p[3, 2, :] = 1.0
    yield "trunc_8x6", p, np.array([0, 0, 75, 61]), 0.5
    yield "offset_ignored", p, np.array([100, 200, 175, 261]), 0.5
    yield "inverted_box", p, np.array([10, 10, 5, 5]), 0.5
    yield "float_box_trunc", p, np.array([0.0, 0.0, 74.9, 60.9]), 0.5

    # constant channel -> index (0,0)
    c = np.full((8, 8, 17), 0.7, np.float32)
    yield "constant", c, np.array([0, 0, 80, 80]), 0.5

    # +inf peak
    i = rs.rand(8, 8, 17).astype(np.float32)
    i[5, 5, :] = np.inf
    yield "plus_inf", i, np.array([0, 0, 80, 60]), 0.5
    # -inf everywhere except one
    m = np.full((8, 8, 17), -np.inf, np.float32)
    m[2, 6, 4] = -1.0
    yield "minus_inf", m, np.array([0, 0, 80, 60]), -2.0

    # signed zeros: -0.0 then +0.0, numpy compares by value -> first index
    z = np.full((4, 4, 17), -0.0, np.float32)
    z[2, 1, :] = 0.0
    yield "signed_zero", z, np.array([0, 0, 40, 40]), -1.0

    # negative values and negative threshold
    g = -rs.rand(12, 20, 17).astype(np.float32)
    yield "negative_vals", g, np.array([3, 5, 99, 205]), -0.5

    # fp16 input
    h16 = rs.rand(32, 32, 17).astype(np.float16)
    yield "fp16_input", h16, np.array([0, 0, 128, 128]), 0.2

    # other sizes, incl. non-square and non-multiple-of-anything
    for (hh, ww) in [(64, 64), (160, 96), (200, 336), (56, 36), (7, 3), (1, 1)]:
        x = rs.rand(hh, ww, 17).astype(np.float32)
        yield f"rand_{hh}x{ww}", x, np.array([0, 0, 4 * hh, 4 * ww]), 0.2
    # sigmoid of N(-4.6, 1.5) logits: the benchmark's distribution
    lg = rs.randn(128, 128, 17).astype(np.float32) * 1.5 - 4.6
    s = (1.0 / (1.0 + np.exp(-lg))).astype(np.float32)
    yield "sigmoid_bench", s, np.array([0, 0, 512, 512]), 0.2
    # duplicated global max far apart (tests cross-block tie-break on the GPU)
    d = rs.rand(128, 128, 17).astype(np.float32) * 0.5
    for j in range(17):
        d[5 + j, 100, j] = 0.75
        d[120, 3 + j, j] = 0.75
        d[64, 64, j] = 0.75
    yield "far_ties", d, np.array([0, 0, 512, 512]), 0.2
