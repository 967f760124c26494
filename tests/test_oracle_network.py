"""CPU: internal consistency of the network oracle (torch restatement vs plain-numpy loops),
parameter inventory, loss/optimizer formulas."""
import math

import numpy as np
import torch

from oracle import network as onet
from oracle import tf_semantics_np as tfnp


def _t(x):
    return torch.tensor(x, dtype=torch.float64)


def test_param_inventory_matches_survey():
    shapes = onet.param_shapes(1.0)
    n_train = sum(int(np.prod(s)) for k, s in shapes.items() if onet.is_trainable(k))
    assert n_train == 5521490          # trainable params
    assert shapes["final_conv3x3/kernel"] == (3, 3, 512, 64)
    assert shapes["heatmaps/kernel"] == (1, 1, 64, 18)
    assert sum(1 for k in shapes if k.endswith("/gamma")) == 40   # 40 BN layers


def test_same_padding_stride2_is_asymmetric():
    assert onet.tf_same_padding(512, 3, 2) == (0, 1)
    assert onet.tf_same_padding(512, 3, 1) == (1, 1)
    assert onet.tf_same_padding(7, 3, 2) == (1, 1)


def test_conv_and_depthwise_same_vs_numpy_loops():
    rs = np.random.RandomState(0)
    for (h, w, s) in [(8, 6, 2), (8, 6, 1), (7, 5, 2)]:
        x = rs.randn(2, h, w, 3)
        k = rs.randn(3, 3, 3, 4)
        got = onet.conv2d_tf_same(_t(x).permute(0, 3, 1, 2), _t(k), s).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, tfnp.conv_same(x, k, s), atol=1e-12)
        kd = rs.randn(3, 3, 3, 1)
        got = onet.depthwise_conv2d_tf_same(_t(x).permute(0, 3, 1, 2), _t(kd), s).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, tfnp.depthwise_same(x, kd, s), atol=1e-12)


def test_resizes_vs_numpy_loops():
    rs = np.random.RandomState(1)
    x = rs.randn(2, 4, 6, 3)
    xt = _t(x).permute(0, 3, 1, 2)
    np.testing.assert_array_equal(onet.nearest_neighbor_upsample(xt).permute(0, 2, 3, 1).numpy(), tfnp.nearest_up2(x))
    for u in (1, 2, 4, 8):
        got = onet.resize_bilinear_legacy(xt, 4 * u, 6 * u).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, tfnp.bilinear_legacy(x, 4 * u, 6 * u), atol=1e-12)
    # legacy downsample by 2 == [::2, ::2]  (keypoints_model.py:73-74)
    m = rs.rand(2, 8, 8, 1)
    got = onet.resize_bilinear_legacy(_t(m).permute(0, 3, 1, 2), 4, 4).permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(got, m[:, ::2, ::2])


def test_batch_norm_training_and_moving_update():
    rs = np.random.RandomState(2)
    x = _t(rs.randn(3, 5, 4, 4) * 2 + 1)
    p = {"bn/gamma": _t(rs.rand(5) + 0.5), "bn/beta": _t(rs.randn(5)),
         "bn/moving_mean": _t(rs.randn(5)), "bn/moving_variance": _t(rs.rand(5) + 0.5)}
    upd = {}
    y = onet.batch_norm(x, p, "bn", True, upd)
    xn = x.numpy()
    mean = xn.mean((0, 2, 3)); var = xn.var((0, 2, 3))
    want = (xn - mean[None, :, None, None]) / np.sqrt(var + 1e-3)[None, :, None, None] \
        * p["bn/gamma"].numpy()[None, :, None, None] + p["bn/beta"].numpy()[None, :, None, None]
    np.testing.assert_allclose(y.numpy(), want, atol=1e-12)
    n = 3 * 4 * 4
    np.testing.assert_allclose(upd["bn/moving_variance"].numpy(),
                               p["bn/moving_variance"].numpy() * 0.95 + var * n / (n - 1) * 0.05, atol=1e-12)
    np.testing.assert_allclose(upd["bn/moving_mean"].numpy(), p["bn/moving_mean"].numpy() * 0.95 + mean * 0.05, atol=1e-12)


def test_focal_loss_against_direct_formula():
    rs = np.random.RandomState(3)
    y = rs.rand(2, 4, 4, 17); y[0, 1, 1, 3] = 1.0; y[1, 2, 0, 5] = 1.0
    x = rs.randn(2, 4, 4, 17) * 3
    nb = np.array([2, 0])
    got = onet.focal_loss(_t(y), torch.tensor(nb), _t(x)).numpy()
    p = 1 / (1 + np.exp(-x))
    pos = y == 1.0
    ce = np.where(pos, -np.log(p), -np.log(1 - p))
    w = np.where(pos, (1 - p) ** 2, (1 - y) ** 4 * p ** 2)
    want = (w * ce).sum(3) / (nb.reshape(-1, 1, 1) + 1.0)
    np.testing.assert_allclose(got, want, rtol=1e-10)


def test_cosine_decay_and_adam():
    assert abs(onet.cosine_decay(3e-4, 0, 200000) - 3e-4) < 1e-18
    assert abs(onet.cosine_decay(3e-4, 200000, 200000) - 3e-8) < 1e-15
    assert abs(onet.cosine_decay(3e-4, 300000, 200000) - 3e-8) < 1e-15
    p = np.array([1.0, -2.0]); g = np.array([300.0, -0.5]); m = np.zeros(2); v = np.zeros(2)
    onet.adam_step(p, g, m, v, 1e-3, 1)
    gc = np.array([200.0, -0.5])
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    want = np.array([1.0, -2.0]) - lr_t * (0.1 * gc) / (np.sqrt(0.001 * gc * gc) + 1e-8)
    np.testing.assert_allclose(p, want, rtol=1e-12)


def test_forward_shapes_config1():
    # BASELINE config 1 plumbing: one 256x256x3 image, forward
    params = {k: torch.tensor(v) for k, v in onet.init_params(0).items()}
    img = torch.rand(1, 256, 256, 3)
    with torch.no_grad():
        heat, enr = onet.forward(img, params, False)
    assert tuple(heat.shape) == (1, 64, 64, 18)
    assert tuple(enr["p2"].shape) == (1, 64, 64, 128) and tuple(enr["p5"].shape) == (1, 8, 8, 128)
    # bias init: sigmoid(-log 99) = 0.01 on the 17 keypoint channels
    assert abs(float(torch.sigmoid(heat[..., :17]).mean()) - 0.01) < 2e-3
