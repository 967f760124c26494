"""CPU checks of the PRN-glue restatement (oracle/prn_post.py): properties that hold for tf.image.crop_and_resize and
the reference's normalisation / argmax_2d (create_pb.py:86-142). The restatement itself is unpinned (no TensorFlow here)."""
import numpy as np

from oracle import prn_post as o


def test_normalisation_range_and_mask():
    rs = np.random.RandomState(0)
    hm = rs.rand(2, 9, 7, 17).astype(np.float32)
    hm[1, :, :, 4] *= 0.1                                   # maximum below the 0.2 threshold
    n, m, M = o.normalize_heatmaps(hm)
    assert n.dtype == np.float32
    assert np.all(n[1, :, :, 4] == 0)
    keep = np.ones((2, 17), bool); keep[1, 4] = False
    assert np.allclose(n.max(axis=(1, 2))[keep], 1.0) and np.allclose(n.min(axis=(1, 2))[keep], 0.0)
    np.testing.assert_array_equal(m, hm.min(axis=(1, 2)))
    np.testing.assert_array_equal(M, hm.max(axis=(1, 2)))


def test_identity_crop_and_extrapolation():
    rs = np.random.RandomState(1)
    img = rs.rand(2, 56, 36, 17).astype(np.float32)
    boxes = np.array([[0, 0, 1, 1], [0, 0, 1, 1], [-0.5, -0.5, 0.5, 0.5], [0.2, 0.2, 0.4, 0.4]], np.float32)
    ind = np.array([0, 1, 0, 5], np.int32)
    c = o.crop_and_resize(img, boxes, ind, (56, 36))
    np.testing.assert_array_equal(c[0], img[0])            # the full box at the image's own size is the image
    np.testing.assert_array_equal(c[1], img[1])
    assert np.all(c[2][:27] == 0) and np.all(c[2][:, :17] == 0) and np.any(c[2][30:, 20:] != 0)   # outside -> 0
    assert np.all(c[3] == 0)                                # box_ind outside the batch: zero crop (padding slots)
    lo, hi = img[0].min(), img[0].max()
    inner = o.crop_and_resize(img, np.array([[0.1, 0.2, 0.7, 0.9]], np.float32), np.array([0], np.int32), (56, 36))
    assert inner.min() >= lo and inner.max() <= hi          # bilinear: inside the range of the taps


def test_decode_one_hot_and_ties():
    z = np.zeros((2, 56, 36, 17), np.float32)
    z[0, 10, 20, 3] = 30.0
    s, p = o.decode(z)
    assert s.shape == (2, 17) and p.shape == (2, 17, 2)
    assert abs(s[0, 3] - 1.0) < 1e-6 and np.allclose(p[0, 3], [10 / 56, 20 / 36])
    assert np.allclose(s[1], 1.0 / 2016) and np.all(p[1] == 0)   # constant channel: uniform softmax, first position
