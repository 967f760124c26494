"""GPU: the fp16 storage type (MPN_F16) of the dense convolutions and of the pose residual network - BASELINE.json
config 5 quotes the PRN in fp16. Same oracle, same cases as the bf16 tests; the tolerance is fp16's 2^-11 rounding of the
stored outputs (operands are rounded to fp16 on both sides, products accumulate in f32)."""
import numpy as np
import pytest
import torch

from oracle import network as onet
from oracle import prn as oprn
from util import act_ref, assert_close, dev, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu
F16 = torch.float16


def _ops():
    from multiposenet_amd import ops
    return ops


CASES = [
    # N, H, W, Cin, Cout, k, affine_act, stats
    (2, 16, 16, 128, 128, 3, 1, True),
    (1, 12, 20, 32, 64, 3, 2, True),
    (1, 16, 16, 512, 64, 3, 0, False),
    (2, 16, 16, 64, 256, 1, 2, True),
    (1, 10, 6, 32, 64, 1, 2, True),          # ragged M
    (3, 16, 16, 1024, 1024, 1, 2, True),     # stays on the tiled kernel (the GEMM kernel of pointwise.hip is bf16 only)
    (1, 1, 128, 1024, 34272, 1, 0, False),   # the PRN's second layer as the library runs it
    (2, 6, 10, 136, 200, 1, 2, True),
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[3]}to{c[4]}k{c[5]}_{c[1]}x{c[2]}" for c in CASES])
def test_conv_fwd_fp16(cuda, case):
    ops = _ops()
    N, H, W, Cin, Cout, k, act, stats = case
    rs = np.random.RandomState(Cin + Cout + k)
    x = rnd(rs.randn(N, H, W, Cin), F16)
    w = rs.randn(k, k, Cin, Cout).astype(np.float32) / np.sqrt(k * k * Cin)
    aff, a = None, x
    if act:
        sc = torch.tensor(0.5 + rs.rand(Cin), dtype=torch.float32)
        sh = torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32)
        aff = ops.Affine(dev(sc), dev(sh), act)
        a = rnd(act_ref(x * sc + sh, act), F16)
    want = nhwc(onet.conv2d_same(nchw(a), rnd(w, F16)))
    pc = ops.PackedConv(dev(w), F16)
    part = None
    if stats:
        part = torch.full((ops.conv_num_parts(N, H, W, k), 2, Cout), float("nan"), device="cuda")
    y = ops.conv_fwd(dev(x, F16), pc.fwd, Cout, k, aff, stats_part=part)
    assert y.dtype == F16
    assert_close(y, want, F16, k * k * Cin)
    if stats:
        rows = ops.conv_stats_rows(N, H, W, Cin, Cout, k, F16)
        assert bool(torch.isnan(part[rows:]).all())
        s = part[:rows].double().sum(0).cpu()
        wd = want.double().reshape(-1, Cout)
        np.testing.assert_allclose(s[0].numpy() / wd.shape[0], wd.mean(0).numpy(), atol=5e-4)
        np.testing.assert_allclose(s[1].numpy() / wd.shape[0], (wd * wd).mean(0).numpy(), rtol=3e-3)


@pytest.mark.parametrize("k,Cin,Cout", [(3, 128, 128), (1, 256, 128), (1, 512, 1024)])
def test_conv_dgrad_fp16(cuda, k, Cin, Cout):
    ops = _ops()
    rs = np.random.RandomState(k + Cin)
    N, H, W = 2, 12, 16
    w = rs.randn(k, k, Cin, Cout).astype(np.float32) / np.sqrt(k * k * Cin)
    dy = rnd(rs.randn(N, H, W, Cout), F16)
    xin = torch.zeros(N, Cin, H, W, requires_grad=True)
    onet.conv2d_same(xin, rnd(w, F16)).backward(nchw(dy))
    pc = ops.PackedConv(dev(w), F16)
    dx = ops.conv_fwd(dev(dy, F16), pc.bwd, Cin, k)
    assert_close(dx, nhwc(xin.grad), F16, k * k * Cout)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,act", [(2, 16, 16, 128, 128, 3, 1), (2, 16, 16, 512, 64, 3, 0), (2, 16, 16, 256, 128, 1, 2),
                                                 (1, 10, 6, 64, 128, 1, 2), (1, 1, 34272, 128, 1024, 1, 0)])
def test_conv_wgrad_fp16(cuda, N, H, W, Cin, Cout, k, act):
    ops = _ops()
    rs = np.random.RandomState(Cin + k + N)
    x = rnd(rs.randn(N, H, W, Cin), F16)
    dy = rnd(rs.randn(N, H, W, Cout), F16)
    aff, a = None, x
    if act:
        sc = torch.tensor(0.5 + rs.rand(Cin), dtype=torch.float32)
        sh = torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32)
        aff = ops.Affine(dev(sc), dev(sh), act)
        a = rnd(act_ref(x * sc + sh, act), F16)
    w = torch.zeros(k, k, Cin, Cout, requires_grad=True)
    onet.conv2d_same(nchw(a), w).backward(nchw(dy))
    dw = torch.full((k, k, Cin, Cout), float("nan"), device="cuda")
    ops.conv_bwd_weight(dev(x, F16), dev(dy, F16), k, aff, dw)
    assert_close(dw, w.grad, torch.float32, N * H * W, scale=float(w.grad.abs().max()))   # f32 result of exact fp16 products


def test_prn_fp16_forward_loss_grads_vs_oracle(cuda):
    """Small PRN in fp16 against the f64 oracle evaluated on the fp16-rounded weights and input: what remains is the
    fp16 rounding of the stored activations (hidden, y2) - 2^-11 each."""
    from multiposenet_amd.prn import PoseResidualNet
    from test_prn_gpu import _data, _values
    B, h, w, hidden = 16, 8, 6, 1024
    rs = np.random.RandomState(5)
    x, y = _data(rs, B, h, w, 17)
    vals = _values(3, h, w, 17, hidden)
    net = PoseResidualNet(values=vals, batch=B, h=h, w=w, hidden=hidden, dtype=F16)
    assert net.loss_scale == 8192.0          # largest power of two <= 16*8*6*17 (exact scaling)
    net.forward(torch.from_numpy(x).cuda())
    loss = float(net.loss(torch.from_numpy(y).cuda()))
    net.backward()
    p = {k: (rnd(v, F16).double() if k.endswith("weights") else torch.tensor(v, dtype=torch.float64)).requires_grad_(True)
         for k, v in vals.items()}
    # the residual term of the logits is the f32 input; the fc1 operand is its fp16 rounding (prn.py forward)
    xt, xr = torch.tensor(x, dtype=torch.float64), rnd(x, F16).double()
    logits = oprn.prn(xr, p) - xr + xt
    want = oprn.log_loss(torch.tensor(y, dtype=torch.float64), logits)
    want.backward()
    np.testing.assert_allclose(net.logits.cpu().numpy().reshape(B, h, w, 17), logits.detach().numpy(), atol=4e-3, rtol=4e-3)
    np.testing.assert_allclose(loss, float(want.detach()), rtol=2e-3)
    for k in vals:
        g, gw = net.grads[k].cpu().numpy().ravel() / net.loss_scale, p[k].grad.numpy().ravel()   # (backward carries the loss scale)
        cos = float(np.dot(g, gw) / (np.linalg.norm(g) * np.linalg.norm(gw) + 1e-30))
        assert cos > 0.9995, (k, cos)
        assert np.abs(g - gw).max() <= 2e-2 * np.abs(gw).max() + 1e-9, k


def test_prn_fp16_tracks_f32_full_size(cuda):
    """BASELINE config 5 as named: 56x36x17 crops, hidden 1024, fp16 operands. fp16's 11-bit significand tracks the f32
    build closer than bf16 does (tests/test_prn_gpu.py holds the bf16 bound: loss 1e-2, gradient cosine 0.995)."""
    from multiposenet_amd.prn import PoseResidualNet
    from test_prn_gpu import _data, _values
    rs = np.random.RandomState(11)
    B, h, w, hidden = 128, 56, 36, 1024
    x, y = _data(rs, B, h, w, 17)
    vals = _values(13, h, w, 17, hidden)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = {}
    for dt in (torch.float32, F16):
        net = PoseResidualNet(values=vals, batch=B, dtype=dt)
        net.forward(xd)
        loss = float(net.loss(yd))
        net.backward()
        out[dt] = (loss, net.grad.cpu().numpy().copy() / net.loss_scale)
        assert np.all(np.isfinite(out[dt][1]))
        assert np.isfinite(float(net.train_step(xd, yd, 1e-3, 200000)))
        del net
    lf, gf = out[torch.float32]
    lh, gh = out[F16]
    np.testing.assert_allclose(lh, lf, rtol=2e-3)
    cos = float(np.dot(gf, gh) / (np.linalg.norm(gf) * np.linalg.norm(gh)))
    assert cos > 0.9995, cos
    np.testing.assert_allclose(np.linalg.norm(gh), np.linalg.norm(gf), rtol=5e-3)   # nothing lost to fp16 underflow


def test_assigner_fp16_agrees_with_f32(cuda):
    """BASELINE config 5's inference side (create_pb.py:86-142) with the fp16 PRN: same scores / positions as the f32
    build up to argmax flips on near-ties."""
    from multiposenet_amd.prn import PoseResidualNet, initial_values
    from multiposenet_amd.prn_inference import KeypointAssigner
    from test_prn_post_gpu import _boxes, _heatmaps
    rs = np.random.RandomState(4)
    b, max_boxes = 2, 4
    hm = torch.tensor(_heatmaps(rs, b, 64, 48)).cuda()
    boxes = torch.tensor(_boxes(rs, b * max_boxes).reshape(b, max_boxes, 4)).cuda()
    num = torch.tensor(np.array([4, 2], np.int32)).cuda()
    values = initial_values(seed=3)
    got = {}
    for dt in (torch.float32, F16):
        net = PoseResidualNet(values=values, batch=8, dtype=dt)
        s, p = KeypointAssigner(net)(hm, boxes, num, compact=True)
        got[dt] = (s.cpu().numpy(), p.cpu().numpy())
    np.testing.assert_allclose(got[F16][0], got[torch.float32][0], rtol=1e-2)
    assert np.mean(np.all(got[F16][1] == got[torch.float32][1], axis=-1)) >= 0.95


def test_prn_model_fn_accepts_fp16(cuda):
    from multiposenet_amd.prn_model import model_fn, reset_registry
    from multiposenet_amd.keypoints_model import ModeKeys
    from test_prn_gpu import _data, _values
    reset_registry()
    x, y = _data(np.random.RandomState(0), 8, 8, 6, 17)    # (the batch is a GEMM dimension: a multiple of 8)
    hp = {"dtype": "fp16", "initial_learning_rate": 1e-3, "num_steps": 100, "model_dir": "fp16-test",
          "values": _values(1, 8, 6, 17, 1024)}
    try:
        spec = model_fn(x, y, ModeKeys.TRAIN, hp)
        assert np.isfinite(float(spec.loss))
    finally:
        reset_registry()
