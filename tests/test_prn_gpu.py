"""GPU: pose residual network (detector/prn.py, prn_model.py) against the torch-CPU oracle (oracle/prn.py)."""
import numpy as np
import pytest
import torch

from oracle import prn as oprn

pytestmark = pytest.mark.gpu


def _data(rs, B, h, w, c):
    x = np.zeros((B, h, w, c), np.float32)
    y = np.zeros((B, h, w, c), np.float32)
    for b in range(B):
        for j in range(c):
            if rs.rand() < 0.8:     # a blurred blob in the input crop, a one-hot peak in the label (prn_pipeline.py)
                py, px = rs.randint(0, h), rs.randint(0, w)
                y[b, py, px, j] = 1.0
                yy, xx = np.mgrid[0:h, 0:w]
                x[b, :, :, j] = np.exp(-((yy - py - rs.randn()) ** 2 + (xx - px - rs.randn()) ** 2) / 8.0)
    x += 0.05 * rs.rand(B, h, w, c).astype(np.float32)
    return x, y


def _values(seed, h, w, c, hidden):
    p = oprn.init_params(seed, h, w, c, hidden)
    rs = np.random.RandomState(seed + 1)
    for k in p:                      # non-zero biases and larger weights so that both ReLUs are partly active
        if k.endswith("biases"):
            p[k] = (0.1 * rs.randn(*p[k].shape)).astype(np.float32)
        else:
            p[k] = (p[k] * 3.0).astype(np.float32)
    return p


@pytest.mark.parametrize("B,h,w,hidden", [(16, 8, 6, 64), (128, 56, 36, 1024)])
def test_prn_f32_forward_loss_grads_vs_oracle(cuda, B, h, w, hidden):
    from multiposenet_amd.prn import PoseResidualNet
    rs = np.random.RandomState(B)
    x, y = _data(rs, B, h, w, 17)
    vals = _values(3, h, w, 17, hidden)
    net = PoseResidualNet(values=vals, batch=B, h=h, w=w, hidden=hidden, dtype=torch.float32)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    net.forward(xd)
    loss = float(net.loss(yd))
    net.backward()
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in vals.items()}
    logits = oprn.prn(torch.tensor(x, dtype=torch.float64), p)
    want = oprn.log_loss(torch.tensor(y, dtype=torch.float64), logits)
    want.backward()
    np.testing.assert_allclose(net.logits.cpu().numpy().reshape(B, h, w, 17), logits.detach().numpy(), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(loss, float(want), rtol=2e-4)
    for k in vals:
        g, gw = net.grads[k].cpu().numpy(), p[k].grad.numpy()
        assert np.abs(g - gw).max() <= 1e-3 * np.abs(gw).max() + 1e-9, k


def test_prn_train_step_and_model_fn_f32(cuda):
    from multiposenet_amd.prn_model import model_fn
    from multiposenet_amd.keypoints_model import ModeKeys
    rs = np.random.RandomState(5)
    B, h, w, hidden = 16, 8, 6, 1024
    x, y = _data(rs, B, h, w, 17)
    vals = _values(7, h, w, 17, hidden)
    hp = {"initial_learning_rate": 1e-3, "num_steps": 200000, "dtype": "f32", "values": vals}
    ref = {k: v.copy() for k, v in vals.items()}
    m = {k: np.zeros_like(v) for k, v in ref.items()}
    v_ = {k: np.zeros_like(v) for k, v in ref.items()}
    losses = []
    for step in range(2):
        spec = model_fn(x, y, ModeKeys.TRAIN, hp)
        want, _, _ = oprn.train_step(ref, m, v_, x, y, step, hp)
        losses.append((float(spec.loss), want))
    for got, want in losses:
        np.testing.assert_allclose(got, want, rtol=5e-4)
    from multiposenet_amd import prn_model
    net = next(iter(prn_model._models.values()))
    sd = net.state_dict()
    for k in ref:   # Adam moves every weight by ~lr per step: allow sign flips on tiny gradients
        assert np.abs(sd[k] - ref[k]).max() <= 2 * 2 * 1e-3 + 1e-6, k
        assert np.mean(np.abs(sd[k] - ref[k]) > 1e-4) < 0.02, k
    ev = model_fn(x, y, ModeKeys.EVAL, hp)
    assert np.isfinite(float(ev.loss)) and "eval_loss" in ev.eval_metric_ops
    assert int(net.global_step.item()) == 2


def test_prn_bf16_tracks_f32_full_size(cuda):
    from multiposenet_amd.prn import PoseResidualNet
    rs = np.random.RandomState(11)
    B, h, w, hidden = 128, 56, 36, 1024
    x, y = _data(rs, B, h, w, 17)
    vals = _values(13, h, w, 17, hidden)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        net = PoseResidualNet(values=vals, batch=B, dtype=dt)
        net.forward(xd)
        loss = float(net.loss(yd))
        net.backward()
        out[dt] = (loss, net.grad.cpu().numpy().copy())
        del net
    lf, gf = out[torch.float32]
    lb, gb = out[torch.bfloat16]
    np.testing.assert_allclose(lb, lf, rtol=1e-2)
    cos = float(np.dot(gf, gb) / (np.linalg.norm(gf) * np.linalg.norm(gb)))
    assert cos > 0.995


def test_prn_shim_signature(cuda):
    from multiposenet_amd.detector.prn import prn
    rs = np.random.RandomState(2)
    x, _ = _data(rs, 8, 8, 6, 17)
    vals = _values(1, 8, 6, 17, 1024)
    got = prn(x, is_training=False, values=vals, dtype=torch.float32).cpu().numpy()
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
    want = oprn.prn(torch.tensor(x, dtype=torch.float64), p).numpy()
    np.testing.assert_allclose(got, want, atol=1e-3, rtol=1e-3)


def test_prn_model_fn_eval_at_another_batch_size_scores_the_trained_variables(cuda):
    """variables are keyed by (model_dir, dtype, seed), not by batch size - the reference's eval pipeline ends on
    a partial batch (`dataset.repeat(1).batch(b)`), which must see the weights TRAIN produced, and a partial train batch
    must step the same model."""
    from multiposenet_amd import prn_model
    from multiposenet_amd.keypoints_model import ModeKeys
    from multiposenet_amd.prn import PoseResidualNet
    from multiposenet_amd.prn_model import model_fn
    prn_model.reset_registry()
    rs = np.random.RandomState(21)
    h, w, hidden = 8, 6, 1024
    x4, y4 = _data(rs, 4, h, w, 17)
    x3, y3 = _data(rs, 3, h, w, 17)
    vals = _values(3, h, w, 17, hidden)
    hp = {"initial_learning_rate": 1e-2, "num_steps": 1000, "dtype": "f32", "values": vals, "model_dir": "prn-advice"}
    untrained = float(model_fn(x3, y3, ModeKeys.EVAL, hp).loss)
    for _ in range(3):
        model_fn(x4, y4, ModeKeys.TRAIN, hp)
    got = float(model_fn(x3, y3, ModeKeys.EVAL, hp).loss)
    assert len(prn_model._models) == 1
    base = next(iter(prn_model._models.values()))
    assert int(base.global_step.item()) == 3 and set(base._siblings) == {3, 4}
    fresh = PoseResidualNet(values=base.state_dict(), batch=3, h=h, w=w, dtype=torch.float32)   # the trained weights, alone
    fresh.forward(torch.from_numpy(x3).cuda())
    want = float(fresh.loss(torch.from_numpy(y3).cuda(), with_grad=False))
    assert got == want
    assert abs(got - untrained) > 1e-6          # (and NOT the loss of freshly initialised weights)
    model_fn(x3, y3, ModeKeys.TRAIN, hp)        # a partial train batch steps the SAME model
    assert int(base.global_step.item()) == 4
    prn_model.reset_registry()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 1024, 34272), (24, 256, 2056), (130, 132, 72), (8, 4, 8)])
def test_gemm_nt_split_k(cuda, dtype, M, N, K):
    """mpn_gemm_nt (C = A B^T, both operands K-contiguous, split over K, partial slab + fixed-order reduction) against a
    float64 product of the same 16-bit operands; ragged M / N tiles, a K tail that is not a multiple of the 32-deep step."""
    from multiposenet_amd import ops
    rs = np.random.RandomState(M + N + K)
    a = torch.tensor(rs.randn(M, K).astype(np.float32)).to(dtype).cuda()
    b = torch.tensor((rs.randn(N, K) / np.sqrt(K)).astype(np.float32)).to(dtype).cuda()
    slab = torch.full((ops.gemm_nt_num_parts(K) * M * N + 16,), float("nan"), device="cuda")
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt(a, b, out, slab)
    want = a.double().cpu() @ b.double().cpu().t()
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-5 * float(want.abs().max()) * np.sqrt(K / 32))
    assert bool(torch.isnan(slab[-16:]).all())                      # nothing written behind the slab
    out2 = torch.empty_like(out)
    ops.gemm_nt(a, b, out2, slab)
    assert torch.equal(out, out2)                                   # deterministic
