"""GPU: mpn_heatmap_render (through the reference-shaped shim) is bit-identical to the reference goldens / oracle."""
import numpy as np
import pytest
import torch

from render_cases import cases, _people
from oracle import heatmap_creation as oracle_render
from util import render_golden

pytestmark = pytest.mark.gpu
CASES = list(cases())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_drop_in_matches_reference_goldens(case):
    from multiposenet_amd.detector.input_pipeline import get_heatmaps
    name, kp, boxes, width, height, ds = case
    got = get_heatmaps(kp, boxes, width, height, ds)
    want = render_golden(name)
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)


def test_batch_of_goldens_with_empty_image():
    from multiposenet_amd.detector.input_pipeline import get_heatmaps_batch
    picked = [c for c in CASES if (c[3], c[4], c[5]) == (512, 512, 4)]
    assert len(picked) >= 5
    people = [(c[1], c[2]) for c in picked]
    people.insert(2, (np.zeros((0, 17, 3), np.int32), np.zeros((0, 4), np.float32)))
    out = get_heatmaps_batch(people, 512, 512, 4).cpu().numpy()
    want = [render_golden(c[0]) for c in picked]
    want.insert(2, np.zeros_like(want[0]))
    np.testing.assert_array_equal(out, np.stack(want))


@pytest.mark.parametrize("width,height,ds,persons", [(512, 512, 4, 150), (200, 120, 4, 61), (77, 53, 1, 9)])
def test_many_persons_vs_oracle(width, height, ds, persons):
    # more persons than one culling pass holds (60), ragged tile edges, w % 4 != 0
    from multiposenet_amd.detector.input_pipeline import get_heatmaps
    rs = np.random.RandomState(persons)
    kp, bx = _people(rs, persons, width, height, box_scale=(0.02, 1.0))
    got = get_heatmaps(kp, bx, width, height, ds)
    np.testing.assert_array_equal(got, oracle_render.get_heatmaps(kp, bx, width, height, ds))


def test_full_size_batch_properties():
    # BASELINE cfg2 label shape: 32 x 128 x 128 x 17
    from multiposenet_amd.detector.input_pipeline import HeatmapRenderer
    rs = np.random.RandomState(3)
    people = [_people(rs, rs.randint(1, 12), 512, 512) for _ in range(32)]
    first = np.cumsum([0] + [k.shape[0] for k, _ in people]).astype(np.int32)
    kp = torch.from_numpy(np.concatenate([k for k, _ in people])).cuda()
    bx = torch.from_numpy(np.concatenate([b for _, b in people])).cuda()
    r = HeatmapRenderer(32, 512, 512, 4)
    out = r(kp, bx, torch.from_numpy(first).cuda())
    assert out.shape == (32, 128, 128, 17)
    o = out.cpu().numpy()
    assert o.min() == 0.0 and o.max() == 1.0
    # every visible keypoint is an exact 1.0 peak (the focal loss keys on == 1.0, keypoints_model.py:160)
    for b, (k, _) in enumerate(people):
        cy, cx = oracle_render.centres(k, 512, 512, 128, 128)
        for p, j in zip(*np.nonzero(k[:, :, 2] > 0)):
            assert o[b, cy[p, j], cx[p, j], j] == 1.0
    # images are independent: image 5 alone gives the same map; a second call is idempotent
    np.testing.assert_array_equal(o[5], oracle_render.get_heatmaps(people[5][0], people[5][1], 512, 512, 4))
    other = torch.full_like(out, 7.0)
    r(kp, bx, torch.from_numpy(first).cuda(), out=other)
    assert torch.equal(other, out)


def test_error_behaviour():
    from multiposenet_amd.detector.input_pipeline import get_heatmaps, HeatmapRenderer
    kp = np.zeros((1, 17, 3), np.int32)
    bx = np.array([[0, 0, 10, 10]], np.float32)
    kp[0, 0] = (64, 3, 1)
    with pytest.raises(ValueError):
        get_heatmaps(kp, bx, 64, 64, 4)                  # y == height is out of range
    with pytest.raises(ValueError):
        get_heatmaps(np.zeros((1, 16, 3), np.int32), bx, 64, 64, 4)
    with pytest.raises(ValueError):
        get_heatmaps(np.zeros((2, 17, 3), np.int32), bx, 64, 64, 4)
    with pytest.raises(ValueError):
        HeatmapRenderer(1, 1, 64, 4)
    r = HeatmapRenderer(2, 64, 64, 4)
    with pytest.raises(ValueError):
        r(torch.zeros((1, 17, 3), dtype=torch.int32).cuda(), torch.zeros((1, 4)).cuda(),
          torch.zeros(2, dtype=torch.int32).cuda())      # first_person must have B+1 entries
