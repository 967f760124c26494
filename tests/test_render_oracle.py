"""CPU: the target-heatmap rendering oracle is pinned against goldens produced by the imported reference."""
import numpy as np
import pytest

from render_cases import cases
from oracle import heatmap_creation as oracle_render
from util import render_golden

CASES = list(cases())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_goldens(case):
    name, kp, boxes, width, height, ds = case
    want = render_golden(name)
    got = oracle_render.get_heatmaps(kp, boxes, width, height, ds)
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)


def test_all_golden_names_covered():
    from util import RENDER_GOLD
    assert sorted(RENDER_GOLD["names"].tolist()) == sorted(c[0] for c in CASES)


def test_peaks_are_exactly_one_and_windows_clip():
    # the focal loss (keypoints_model.py:160) keys on `== 1.0`
    name, kp, boxes, width, height, ds = CASES[0]
    hm = oracle_render.get_heatmaps(kp, boxes, width, height, ds)
    h, w, _ = hm.shape
    cy, cx = oracle_render.centres(kp, width, height, w, h)
    for p in range(kp.shape[0]):
        for j in range(17):
            if kp[p, j, 2] > 0:
                assert hm[cy[p, j], cx[p, j], j] == 1.0
    assert oracle_render.half_window(1.0)[0] == 4 and oracle_render.half_window(4.0)[0] == 13
    assert hm.max() == 1.0 and hm.min() == 0.0


def test_empty_person_list():
    hm = oracle_render.get_heatmaps(np.zeros((0, 17, 3), np.int32), np.zeros((0, 4), np.float32), 64, 48, 4)
    assert hm.shape == (12, 16, 17) and not hm.any()
