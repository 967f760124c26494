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


# ------------------------------------------------------------------------------------------------------------------
# Known-answer vectors (computed by hand from the formulas of the TF 1.15 sources cited in oracle/tf_semantics_np.py)
# for the loop-level restatement, then oracle/network.py (torch) against that restatement on random inputs.
# ------------------------------------------------------------------------------------------------------------------
def test_known_answers_fused_batch_norm():
    # one channel, rows 1,2,3,4: mean 2.5, biased var 1.25, Bessel var 5/3; gamma 2, beta -1, eps 1e-3
    x = np.array([1.0, 2.0, 3.0, 4.0]).reshape(1, 2, 2, 1)
    y, mean, var_u, var_b = tfnp.fused_batch_norm_op_train(x, [2.0], [-1.0], 1e-3)
    assert mean[0] == 2.5 and var_b[0] == 1.25 and abs(var_u[0] - 5.0 / 3.0) < 1e-15
    inv = 1.0 / math.sqrt(1.251)
    np.testing.assert_allclose(y.reshape(-1), [-1.5 * inv * 2 - 1, -0.5 * inv * 2 - 1, 0.5 * inv * 2 - 1, 1.5 * inv * 2 - 1], rtol=1e-15)
    # constant input: variance 0, y = beta; a single row: the Bessel factor is rows / max(rows - 1, 1) = 1
    y, mean, var_u, var_b = tfnp.fused_batch_norm_op_train(np.full((1, 1, 3, 1), 7.0), [3.0], [0.25], 1e-3)
    assert mean[0] == 7.0 and var_b[0] == 0.0 and var_u[0] == 0.0 and np.all(y == 0.25)
    y, mean, var_u, var_b = tfnp.fused_batch_norm_op_train(np.array([[[[5.0]]]]), [1.0], [0.0], 1e-3)
    assert var_u[0] == 0.0 and mean[0] == 5.0
    # two channels are independent: channel 1 = (0, 10) -> mean 5, var 25, Bessel 50
    x = np.array([[1.0, 0.0], [3.0, 10.0]]).reshape(1, 1, 2, 2)
    _, mean, var_u, var_b = tfnp.fused_batch_norm_op_train(x, [1.0, 1.0], [0.0, 0.0], 1e-3)
    np.testing.assert_array_equal(mean, [2.0, 5.0]); np.testing.assert_array_equal(var_b, [1.0, 25.0])
    np.testing.assert_array_equal(var_u, [2.0, 50.0])
    # layer, training: moving = moving - (moving - batch) * (1 - 0.95): mean 1.0 -> 1 - (1 - 2.5)*0.05 = 1.075,
    # variance 2.0 -> 2 - (2 - 5/3)*0.05 = 1.98333...
    x = np.array([1.0, 2.0, 3.0, 4.0]).reshape(1, 2, 2, 1)
    _, mm, mv = tfnp.batch_norm_layer(x, [1.0], [0.0], [1.0], [2.0], 0.95, 1e-3, True)
    assert abs(mm[0] - 1.075) < 1e-15 and abs(mv[0] - (2.0 - (2.0 - 5.0 / 3.0) * 0.05)) < 1e-15
    # layer, inference: (x - moving_mean) / sqrt(moving_var + eps) * gamma + beta, moving statistics untouched
    y, mm, mv = tfnp.batch_norm_layer(x, [2.0], [1.0], [2.0], [3.999], 0.95, 1e-3, False)
    np.testing.assert_allclose(y.reshape(-1), [0.0, 1.0, 2.0, 3.0], rtol=1e-15, atol=1e-15)
    assert mm[0] == 2.0 and mv[0] == 3.999


def test_known_answers_adam_cosine_clip():
    # first apply (beta powers = beta): alpha = lr*sqrt(1-0.999)/(1-0.9); m = 0.1 g; v = 0.001 g^2
    lr = 1e-3
    var, m, v, b1p, b2p = tfnp.adam_apply([1.0], [0.0], [0.0], 0.9, 0.999, lr, [2.0])
    alpha = lr * math.sqrt(0.001) / 0.1
    assert abs(m[0] - 0.2) < 1e-15 and abs(v[0] - 0.004) < 1e-15
    assert abs(var[0] - (1.0 - 0.2 * alpha / (math.sqrt(0.004) + 1e-8))) < 1e-15
    assert abs(b1p - 0.81) < 1e-15 and abs(b2p - 0.998001) < 1e-15
    # zero gradient: nothing moves; tiny gradient: epsilon (outside the bias correction) dominates the denominator
    var, m, v, _, _ = tfnp.adam_apply([1.0], [0.0], [0.0], 0.9, 0.999, lr, [0.0])
    assert var[0] == 1.0 and m[0] == 0.0 and v[0] == 0.0
    var, m, v, _, _ = tfnp.adam_apply([0.0], [0.0], [0.0], 0.9, 0.999, lr, [1e-12])
    assert abs(var[0] - (-(1e-13 * alpha) / (math.sqrt(1e-27) + 1e-8))) < 1e-20
    # second apply continues from the updated powers
    var2, m2, v2, b1p2, _ = tfnp.adam_apply(var, m, v, 0.81, 0.998001, lr, [1e-12])
    assert abs(b1p2 - 0.729) < 1e-15
    # cosine decay with alpha = 1e-4 (keypoints_model.py:109-112): start, middle, end, beyond the end
    assert tfnp.cosine_decay(3e-4, 0, 200000, 1e-4) == 3e-4
    assert abs(tfnp.cosine_decay(3e-4, 100000, 200000, 1e-4) - 3e-4 * (0.9999 * 0.5 + 1e-4)) < 1e-18
    assert abs(tfnp.cosine_decay(3e-4, 200000, 200000, 1e-4) - 3e-8) < 1e-18
    assert abs(tfnp.cosine_decay(3e-4, 10 ** 7, 200000, 1e-4) - 3e-8) < 1e-18
    assert abs(tfnp.cosine_decay(1.0, 50000, 200000, 0.0) - 0.5 * (1 + math.sqrt(0.5))) < 1e-15
    assert tfnp.clip_by_value([-300.0, -200.0, 0.5, 200.0, 1e9], -200, 200) == [-200.0, -200.0, 0.5, 200.0, 200.0]


def test_known_answers_cross_entropy_focal_l2():
    ln2 = math.log(2.0)
    ce = tfnp.sigmoid_cross_entropy_with_logits([0.0, 1.0, 1.0, 0.0, 1.0], [0.0, 0.0, 100.0, -100.0, -100.0])
    np.testing.assert_allclose(ce, [ln2, ln2, math.log1p(math.exp(-100.0)), math.log1p(math.exp(-100.0)),
                                    100.0 + math.log1p(math.exp(-100.0))], rtol=1e-15)
    assert tfnp.l2_loss([3.0, 4.0]) == 12.5 and tfnp.l2_loss([]) == 0.0 and tfnp.l2_loss([[1.0, -1.0], [2.0, 0.0]]) == 3.0
    # focal loss, logit 0 (p = 1/2): positive pixel: (1/2)^2 ln2; negative with y = 0: (1)^4 (1/2)^2 ln2;
    # negative with y = 0.5: (1/2)^4 (1/2)^2 ln2; one channel each, num_boxes = 1 -> divide by 2
    y = np.array([1.0, 0.0, 0.5]).reshape(1, 1, 3, 1)
    got = tfnp.focal_loss(y, [1], np.zeros((1, 1, 3, 1)))
    np.testing.assert_allclose(got.reshape(-1), [0.25 * ln2 / 2, 0.25 * ln2 / 2, 0.25 * ln2 / 16 / 2], rtol=1e-15)
    # y = 1 - 2^-24 is NOT an extreme point (exact comparison with 1.0); channels sum; num_boxes = 0 -> divide by 1
    y = np.array([[1.0 - 2.0 ** -24, 1.0]]).reshape(1, 1, 1, 2)
    got = tfnp.focal_loss(y, [0], np.zeros((1, 1, 1, 2)))
    assert abs(got[0, 0, 0] - ((2.0 ** -24) ** 4 * 0.25 * ln2 + 0.25 * ln2)) < 1e-18


def test_known_answers_resize_bilinear_and_mask_halving():
    # 1x2 -> 1x4: scale 0.5: src = 0, .5, 1, 1.5 -> values a, (a+b)/2, b, b (upper index clamped)
    x = np.array([10.0, 20.0]).reshape(1, 1, 2, 1)
    np.testing.assert_array_equal(tfnp.resize_bilinear_tf(x, 1, 4).reshape(-1), [10.0, 15.0, 20.0, 20.0])
    # 2x2 -> 4x4 corner / centre values: x first then y
    x = np.array([[0.0, 4.0], [8.0, 16.0]]).reshape(1, 2, 2, 1)
    y = tfnp.resize_bilinear_tf(x, 4, 4)[0, :, :, 0]
    assert y[0, 0] == 0.0 and y[0, 1] == 2.0 and y[1, 0] == 4.0 and y[1, 1] == 7.0 and y[3, 3] == 16.0 and y[2, 3] == 16.0
    # factor 1 is the identity; halving is exactly [::2, ::2] (src = 2 i, lerp = 0) - keypoints_model.py:73-74
    rs = np.random.RandomState(5)
    m = rs.rand(2, 8, 6, 1)
    np.testing.assert_array_equal(tfnp.resize_bilinear_tf(m, 8, 6), m)
    np.testing.assert_array_equal(tfnp.resize_bilinear_tf(m, 4, 3), m[:, ::2, ::2])
    np.testing.assert_array_equal(tfnp.resize_bilinear_tf(tfnp.resize_bilinear_tf(m, 4, 3), 2, 1), m[:, ::4, ::4][:, :, :1])
    # and the 128 -> 64 -> 32 -> 16 chain of the real label size, on a {0,1} mask
    mask = (rs.rand(1, 128, 128, 1) < 0.9).astype(np.float64)
    cur = mask
    for f in (2, 4, 8):
        cur = tfnp.resize_bilinear_tf(cur, 128 // f, 128 // f)
        np.testing.assert_array_equal(cur, mask[:, ::f, ::f])


def test_torch_oracle_agrees_with_loop_restatement():
    """oracle/network.py (what the HIP kernels are tested against) vs the loop-level restatement, random inputs."""
    rs = np.random.RandomState(11)
    # batch norm: training output + moving update, inference output
    x = rs.randn(2, 3, 4, 5) * 1.7 + 0.3                                   # NHWC
    gamma, beta = rs.rand(5) + 0.5, rs.randn(5)
    mm, mv = rs.randn(5), rs.rand(5) + 0.5
    p = {"bn/gamma": _t(gamma), "bn/beta": _t(beta), "bn/moving_mean": _t(mm), "bn/moving_variance": _t(mv)}
    for training in (True, False):
        upd = {}
        got = onet.batch_norm(_t(x).permute(0, 3, 1, 2), p, "bn", training, upd).permute(0, 2, 3, 1).numpy()
        want, wm, wv = tfnp.batch_norm_layer(x, gamma, beta, mm, mv, 0.95, 1e-3, training)
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12)
        if training:
            np.testing.assert_allclose(upd["bn/moving_mean"].numpy(), wm, rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(upd["bn/moving_variance"].numpy(), wv, rtol=1e-12, atol=1e-14)
    # Adam: three consecutive applies with clipping, against oracle.adam_step (t = 1, 2, 3)
    n = 6
    var0 = rs.randn(n); grads = [rs.randn(n) * s for s in (1.0, 500.0, 1e-6)]
    po, mo, vo = var0.copy(), np.zeros(n), np.zeros(n)
    pl, ml, vl, b1, b2 = list(var0), [0.0] * n, [0.0] * n, 0.9, 0.999
    for t, g in enumerate(grads, 1):
        lr = tfnp.cosine_decay(3e-4, t - 1, 200000, 1e-4)
        assert abs(lr - onet.cosine_decay(3e-4, t - 1, 200000)) < 1e-20
        onet.adam_step(po, g, mo, vo, lr, t)
        pl, ml, vl, b1, b2 = tfnp.adam_apply(pl, ml, vl, b1, b2, lr, tfnp.clip_by_value(g, -200.0, 200.0))
        np.testing.assert_allclose(po, pl, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(mo, ml, rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(vo, vl, rtol=1e-12, atol=1e-18)
    for step in (0, 1, 777, 100000, 199999, 200000, 250000):
        assert abs(onet.cosine_decay(3e-4, step, 200000) - tfnp.cosine_decay(3e-4, step, 200000, 1e-4)) < 1e-19
    # losses: focal + regression + four segmentation terms with the masks halved by resize_bilinear (loops) vs [::2, ::2]
    b, h, w = 2, 16, 16
    logits = rs.randn(b, h, w, 18) * 2
    hm = rs.rand(b, h, w, 17) * 0.9
    hm[0, 3, 4, 2] = 1.0; hm[1, 0, 0, 16] = 1.0; hm[1, 15, 15, 0] = 1.0
    labels = {"heatmaps": hm, "loss_masks": (rs.rand(b, h, w) < 0.8).astype(np.float64),
              "segmentation_masks": (rs.rand(b, h, w) < 0.3).astype(np.float64), "num_boxes": np.array([3, 0])}
    enr = {l: rs.randn(b, h >> (l - 2), w >> (l - 2), 4) for l in (2, 3, 4, 5)}
    want = tfnp.keypoint_losses(logits, {l: enr[l][..., 0] for l in enr}, labels)
    lab_t = {k: (torch.tensor(v) if k == "num_boxes" else _t(v)) for k, v in labels.items()}
    total, got = onet.losses_fn(_t(logits), {f"p{l}": _t(enr[l]) for l in enr}, lab_t)
    for k, v in got.items():
        assert abs(float(v) - want[k]) <= 1e-12 * max(1.0, abs(want[k])), k
    assert abs(float(total) - want["total_loss"]) <= 1e-12 * max(1.0, abs(want["total_loss"]))
    # bilinear up-sampling: the torch oracle lerps y then x, TF x then y - equal up to rounding (documented in network.py)
    xs = rs.randn(1, 5, 7, 3)
    for u in (2, 4, 8):
        got = onet.resize_bilinear_legacy(_t(xs).permute(0, 3, 1, 2), 5 * u, 7 * u).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, tfnp.resize_bilinear_tf(xs, 5 * u, 7 * u), rtol=0, atol=1e-14)


def test_storage_emulation_rounds_where_the_16_bit_build_stores():
    """oracle.network.storage_emulation(bf16): every raw conv output is exactly representable in bf16, the f32 heads are
    not rounded, gradients still reach every master variable, and outside the context nothing changes."""
    rs = np.random.RandomState(0)
    p = onet.randomize_bn(onet.init_params(3), 4)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=onet.is_trainable(k)) for k, v in p.items()}
    x = torch.tensor(rs.rand(1, 64, 64, 3), dtype=torch.float64)
    taps0, taps1 = {}, {}
    h0, _ = onet.forward(x, pt, True, taps=taps0)
    with onet.storage_emulation(torch.bfloat16):
        h1, _ = onet.forward(x, pt, True, taps=taps1)
    h2, _ = onet.forward(x, pt, True)
    assert torch.equal(h0, h2) and not torch.equal(h0, h1)
    for k, v in taps1.items():
        if k == "concat":
            continue          # (level 2's slice holds activated f32 values in the oracle: the build keeps the raw tensor)
        assert torch.equal(v, v.to(torch.bfloat16).to(v.dtype)), k
    assert not torch.equal(taps0["MobilenetV1/Conv2d_3_pointwise/raw"], taps1["MobilenetV1/Conv2d_3_pointwise/raw"])
    rel = float((h1 - h0).norm() / h0.norm())
    assert 1e-6 < rel < 0.2, rel          # (the logits are mostly their bias at initialisation)
    h1.sum().backward()
    assert all(v.grad is not None and torch.isfinite(v.grad).all() for k, v in pt.items() if onet.is_trainable(k))
