"""PRN inference glue (create_pb.py:86-142) on the GPU against the numpy restatement (oracle/prn_post.py, unpinned)."""
import numpy as np
import pytest
import torch

from oracle import prn_post as oracle

pytestmark = pytest.mark.gpu


def _assigner(net=None):
    from multiposenet_amd.prn_inference import KeypointAssigner
    return KeypointAssigner(net)


def _heatmaps(rs, b, h, w):
    hm = 1.0 / (1.0 + np.exp(-(rs.randn(b, h, w, 17) * 1.5 - 3.0)))
    hm[0, :, :, 3] = 0.1 * hm[0, :, :, 3]          # a channel whose maximum stays under the 0.2 threshold -> masked
    return hm.astype(np.float32)


def _boxes(rs, n):
    y1, x1 = rs.rand(n) * 0.6 - 0.05, rs.rand(n) * 0.6 - 0.05      # some boxes stick out of the image (extrapolation)
    hh, ww = 0.1 + rs.rand(n) * 0.5, 0.05 + rs.rand(n) * 0.5
    return np.stack([y1, x1, y1 + hh, x1 + ww], 1).astype(np.float32)


@pytest.mark.parametrize("b,h,w", [(2, 32, 24), (3, 128, 128), (1, 65, 31)])
def test_minmax_and_crops_match_the_restatement(cuda, b, h, w):
    rs = np.random.RandomState(b * 100 + h)
    hm = _heatmaps(rs, b, h, w)
    n = 9
    boxes = _boxes(rs, n)
    boxes[1] = [0.0, 0.0, 1.0, 1.0]                # the whole image
    boxes[2] = [0.25, 0.25, 0.25, 0.75]            # zero height: every row samples the same line
    ind = rs.randint(0, b, n).astype(np.int32)
    ind[4] = -1                                    # padding slot -> zero crop
    a = _assigner()
    got = a.crops(torch.tensor(hm).cuda(), torch.tensor(boxes).cuda(), torch.tensor(ind).cuda()).cpu().numpy()
    norm, m, M = oracle.normalize_heatmaps(hm)
    want = oracle.crop_and_resize(norm, boxes, ind, (56, 36))
    assert got.shape == (n, 56, 36, 17)
    assert np.all(got[4] == 0)
    assert np.all(got[ind == 0][..., 3] == 0)      # masked channel
    np.testing.assert_array_equal(got, want)       # same f32 operations in the same order: bit-identical


def test_decode_matches_the_restatement(cuda):
    rs = np.random.RandomState(7)
    z = (rs.randn(6, 56, 36, 17) * 3).astype(np.float32)
    z[0, 10, 20, 5] = 40.0                         # a sharp peak: probability ~1
    z[1, :, :, 2] = 0.25                           # constant channel: every position ties, first index wins
    z[2, 55, 35, 0] = z[2, 3, 4, 0] = 50.0         # two equal maxima: the first one
    a = _assigner()
    s, p = a.decode(torch.tensor(z).cuda())
    ws, wp = oracle.decode(z)
    np.testing.assert_array_equal(p.cpu().numpy(), wp)                  # integer argmax / (56, 36): exact
    np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=1e-5, atol=0)  # expf ulps + sum order of 2016 f32 terms
    assert tuple(p[1, 2].tolist()) == (0.0, 0.0)
    assert np.allclose(p[2, 0].cpu().numpy(), [3 / 56, 4 / 36])


def test_assigner_end_to_end(cuda):
    """heatmaps + boxes -> crops -> PRN (f32 build) -> scores / positions, against the oracle chain with the oracle PRN."""
    from multiposenet_amd.prn import PoseResidualNet, initial_values
    from oracle import prn as oprn
    rs = np.random.RandomState(11)
    b, max_boxes = 2, 4
    hm = _heatmaps(rs, b, 64, 48)
    boxes = _boxes(rs, b * max_boxes).reshape(b, max_boxes, 4)
    num = np.array([3, 1], np.int32)
    values = initial_values(seed=3)
    net = PoseResidualNet(values=values, batch=8, dtype=torch.float32)
    a = _assigner(net)
    s, p = a(torch.tensor(hm).cuda(), torch.tensor(boxes).cuda(), torch.tensor(num).cuda(), compact=True)
    assert s.shape == (4, 17) and p.shape == (4, 17, 2)
    norm, _, _ = oracle.normalize_heatmaps(hm)
    fb = np.concatenate([boxes[i, :num[i]] for i in range(b)])
    fi = np.concatenate([np.full(num[i], i, np.int32) for i in range(b)])
    crops = oracle.crop_and_resize(norm, fb, fi, (56, 36))
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in values.items()}
    logits = oprn.prn(torch.tensor(crops, dtype=torch.float64), pt).numpy()
    ws, wp = oracle.decode(logits.astype(np.float32))
    np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=5e-3)          # f32 GEMMs over K = 34272
    agree = np.mean(np.all(p.cpu().numpy() == wp, axis=-1))
    assert agree >= 0.95, agree                                         # argmax flips only on near-ties of the logits
