"""GPU: forward kernels (through the C ABI) vs the oracle's TF-semantics restatement."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import network as onet
from util import DTYPES, act_ref, assert_close, dev, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu
IDS = ["f32", "bf16"]


def _ops():
    from multiposenet_amd import ops
    return ops


CONV_CASES = [
    # N, H, W, Cin, Cout, k, affine_act, stats, up_res
    (2, 16, 16, 128, 128, 3, 1, True, False),
    (1, 12, 20, 32, 64, 3, 2, True, False),      # partial tiles, 128-byte rows
    (1, 16, 16, 512, 64, 3, 0, True, False),     # multi-chunk K (final_conv3x3 shape)
    (2, 8, 8, 64, 512, 3, 0, False, False),      # 4 n-tiles (final_conv3x3 dgrad shape)
    (1, 4, 4, 128, 128, 3, 1, True, False),      # map smaller than a tile
    (2, 16, 16, 64, 256, 1, 2, True, False),
    (1, 8, 8, 1024, 128, 1, 2, False, False),    # lateral5 shape
    (2, 16, 16, 256, 128, 1, 2, False, True),    # lateral + upsample-add (narrow map: the direct epilogue)
    (1, 32, 32, 128, 128, 1, 2, False, True),    # ... through the tile image: a tile = four map rows
    (2, 64, 48, 256, 128, 1, 2, False, True),    # ... tiles that start inside a row and wrap twice
    (3, 10, 40, 64, 128, 1, 1, False, True),     # ... tiles that cross an image boundary, ragged last tile
    (1, 128, 128, 128, 128, 1, 2, False, True),  # ... lateral2's map: a tile = one row
    (1, 10, 6, 32, 64, 1, 2, True, False),       # ragged M
    (3, 16, 16, 1024, 1024, 1, 2, True, False),  # pointwise 13 shape
    (1, 10, 14, 40, 72, 3, 1, True, False),      # channel counts off the tile grid: partial K chunk, partial last n-tile
    (2, 6, 10, 136, 200, 1, 2, True, False),     # same for 1x1 (XCD-remapped block map, 4 n-tiles of 64)
    # the GEMM kernel of pointwise.hip (bf16, K >= 256, N % 256 == 0; the f32 build stays on the tiled kernel)
    (2, 16, 16, 512, 512, 1, 2, True, False),    # 128-pixel tiles (few tiles), affine + ReLU6, statistics
    (1, 13, 11, 256, 256, 1, 1, True, False),    # ragged M (143 pixels): partial tile, partial statistics row
    (2, 16, 16, 512, 256, 1, 0, False, False),   # no affine: A rows by LDS-DMA (the data-gradient configuration)
    (1, 9, 15, 320, 512, 1, 0, True, False),     # K = 5 k-steps, ragged M, LDS-DMA path with statistics
    (8, 96, 96, 256, 512, 1, 2, True, False),    # 256-pixel tiles (73 728 pixels x 2 n-tiles = 576 blocks)
    (9, 85, 83, 256, 256, 1, 0, True, False),    # 256-pixel tiles, ragged M (63 495 pixels), LDS-DMA path
    # the persistent 3x3 kernel of conv3x3.hip (16-bit storage, Cin % 64 == 0, Cout % 128 == 0; f32 stays on the tiled kernel)
    (6, 112, 112, 128, 128, 3, 1, True, False),  # 294 tiles on 256 persistent blocks: some blocks walk two tiles
    (3, 100, 52, 64, 256, 3, 2, True, False),    # one 64-channel chunk per tile, two n-tiles, ragged tiles (100 = 6 x 16 + 4)
    (2, 40, 24, 256, 128, 3, 0, False, False),   # four chunks, no affine / statistics (the data-gradient configuration)
    (1, 20, 36, 64, 640, 3, 1, True, False),     # more than 512 OUTPUT channels (five n-tiles): only the fused reduction's table is sized by Cout
    (1, 16, 16, 64, 1024, 3, 0, False, False),   # ... plain
    # its 64-channel-tile variant (Cout % 128 == 64: the two waves of a row group split K and add their sums through LDS)
    (5, 128, 128, 512, 64, 3, 1, True, False),   # final_conv3x3 at its full map: 320 tiles on 256 blocks, 8 chunks, affine + statistics
    (2, 50, 38, 64, 64, 3, 1, True, False),      # the detector's tower shape, one chunk, ragged tiles
    (3, 24, 40, 128, 192, 3, 2, True, False),    # three n-tiles of 64
    (2, 33, 17, 256, 64, 3, 0, False, False),    # no affine / statistics (a tower's data gradient)
]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[f"{c[3]}to{c[4]}k{c[5]}_{c[1]}x{c[2]}" for c in CONV_CASES])
def test_conv_fwd(cuda, dtype, case):
    ops = _ops()
    N, H, W, Cin, Cout, k, act, stats, upres = case
    rs = np.random.RandomState(Cin + Cout + k)
    x = rnd(rs.randn(N, H, W, Cin), dtype)
    w = rs.randn(k, k, Cin, Cout).astype(np.float32) / np.sqrt(k * k * Cin)
    wq = rnd(w, dtype)
    aff = None
    a = x
    if act:
        sc = torch.tensor(0.5 + rs.rand(Cin), dtype=torch.float32)
        sh = torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32)
        aff = ops.Affine(dev(sc), dev(sh), act)
        a = rnd(act_ref(x * sc + sh, act), dtype)   # the A operand is rounded to the storage type
    res = None
    want = nhwc(onet.conv2d_same(nchw(a), wq))
    if upres:
        r = rnd(rs.randn(N, H // 2, W // 2, Cout), dtype)
        res = dev(r, dtype)
        want = want + nhwc(onet.nearest_neighbor_upsample(nchw(r)))
    pc = ops.PackedConv(dev(w), dtype)
    part = None
    if stats:
        nparts = ops.conv_num_parts(N, H, W, k)
        part = torch.full((nparts, 2, Cout), float("nan"), device="cuda")
    y = ops.conv_fwd(dev(x, dtype), pc.fwd, Cout, k, aff, stats_part=part, up_res=res)
    assert_close(y, want, dtype, k * k * Cin)
    if stats:
        # the rows the kernel says it writes are all written, nothing behind them is (one row per block on the persistent 3x3 kernel)
        rows = ops.conv_stats_rows(N, H, W, Cin, Cout, k, dtype)
        assert 0 < rows <= nparts and bool(torch.isnan(part[rows:]).all())
        s = part[:rows].double().sum(0).cpu()
        wd = want.double().reshape(-1, Cout)
        n = wd.shape[0]
        np.testing.assert_allclose(s[0].numpy() / n, wd.mean(0).numpy(), atol=3e-3 if dtype == torch.bfloat16 else 1e-5)
        np.testing.assert_allclose(s[1].numpy() / n, (wd * wd).mean(0).numpy(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("k,Cin,Cout", [(3, 128, 128), (3, 512, 64), (3, 64, 64), (3, 64, 192), (1, 256, 128), (1, 64, 128), (1, 512, 1024), (1, 256, 512)])
def test_conv_dgrad_is_transposed_conv(cuda, dtype, k, Cin, Cout):
    ops = _ops()
    rs = np.random.RandomState(k + Cin)
    N, H, W = 2, 12, 16
    w = rs.randn(k, k, Cin, Cout).astype(np.float32) / np.sqrt(k * k * Cin)
    dy = rnd(rs.randn(N, H, W, Cout), dtype)
    xin = torch.zeros(N, Cin, H, W, requires_grad=True)
    out = onet.conv2d_same(xin, rnd(w, dtype))
    out.backward(nchw(dy))
    pc = ops.PackedConv(dev(w), dtype)
    dx = ops.conv_fwd(dev(dy, dtype), pc.bwd, Cin, k)
    assert_close(dx, nhwc(xin.grad), dtype, k * k * Cout)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("N,H,W,C,stride", [(2, 16, 16, 32, 1), (2, 16, 16, 64, 2), (1, 12, 20, 128, 1), (1, 14, 10, 256, 2),
                                            (1, 9, 7, 64, 2), (2, 8, 8, 1024, 1), (1, 32, 32, 512, 2), (1, 6, 6, 8, 1),
                                            (1, 9, 7, 64, 1), (2, 5, 13, 32, 1), (1, 70, 33, 128, 1)])   # odd widths: the two-column kernel's tail
def test_dwconv_fwd(cuda, dtype, N, H, W, C, stride):
    ops = _ops()
    rs = np.random.RandomState(C + stride)
    x = rnd(rs.randn(N, H, W, C), dtype)
    w = rs.randn(3, 3, C, 1).astype(np.float32) / 3
    sc = torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)
    sh = torch.tensor(rs.randn(C) * 0.5, dtype=torch.float32)
    a = torch.clamp(x * sc + sh, 0, 6)
    want = nhwc(onet.depthwise_conv2d_tf_same(nchw(a), torch.tensor(w), stride))
    nparts = ops.dwconv_num_parts(N, H, W, C, stride, dtype)
    part = torch.full((nparts, 2, C), float("nan"), device="cuda")
    y = ops.dwconv_fwd(dev(x, dtype), dev(w), stride, ops.Affine(dev(sc), dev(sh), 2), stats_part=part)
    assert tuple(y.shape) == tuple(want.shape)
    assert_close(y, want, dtype, 9)
    s = part.double().sum(0).cpu()
    n = want.numel() // C
    np.testing.assert_allclose(s[0].numpy() / n, want.double().reshape(-1, C).mean(0).numpy(), atol=1e-5 * 100)
    np.testing.assert_allclose(s[1].numpy() / n, (want.double() ** 2).reshape(-1, C).mean(0).numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("C0", [32, 16, 64])
@pytest.mark.parametrize("N,H,W,u8", [(2, 32, 32, False), (1, 64, 48, False), (1, 30, 34, True), (1, 17, 9, False)])
def test_stem_fwd(cuda, dtype, N, H, W, u8, C0):
    ops = _ops()
    rs = np.random.RandomState(H)
    w = rs.randn(3, 3, 3, C0).astype(np.float32) / 5
    if u8:
        img8 = rs.randint(0, 256, (N, H, W, 3)).astype(np.uint8)
        img = torch.tensor(img8.astype(np.float32) * np.float32(1 / 255.0))
        d_img = dev(img8)
    else:
        img = torch.tensor(rs.rand(N, H, W, 3).astype(np.float32))
        d_img = dev(img)
    want = nhwc(onet.conv2d_tf_same(nchw(2.0 * img - 1.0), torch.tensor(w), 2))
    y = ops.stem_conv_fwd(d_img, dev(w), C0, dtype)
    assert_close(y, want, dtype, 27)
    if dtype == torch.bfloat16:
        # the matrix-core kernel splits image and weights into bf16 hi + lo parts: the sums keep f32 accuracy, so the stored
        # output is the bf16 rounding of the exact value except where that value sits on a rounding boundary
        assert float((y.cpu() == want.to(torch.bfloat16)).float().mean()) > 0.995
    # the same launch with its batch-norm partial sums: identical output; the slab finalizes to the mean / variance of the
    # stored (rounded) output, exactly what mpn_bn_stats on that tensor gives (up to the f32 summation order)
    rows = ops.stem_conv_fwd_num_parts(N, H, W, C0, dtype)
    assert rows > 0
    slab = torch.full((rows * 2 * C0,), float("nan"), device="cuda")
    y2 = ops.stem_conv_fwd(d_img, dev(w), C0, dtype, stats_part=slab)
    assert torch.equal(y, y2)
    M = y.numel() // C0
    one = lambda: torch.ones(C0, device="cuda")
    bn_a = ops.BNState(one(), torch.zeros(C0, device="cuda"), torch.zeros(C0, device="cuda"), one(), 2)
    bn_b = ops.BNState(one(), torch.zeros(C0, device="cuda"), torch.zeros(C0, device="cuda"), one(), 2)
    ops.bn_finalize(bn_a, slab, rows, M)
    part, nparts = ops.bn_stats(y)
    ops.bn_finalize(bn_b, part, nparts, M)
    yd = y.double().reshape(M, C0)
    np.testing.assert_allclose(bn_a.mean.cpu().numpy(), yd.mean(0).cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(bn_a.mean.cpu().numpy(), bn_b.mean.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(bn_a.invstd.cpu().numpy(), bn_b.invstd.cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(bn_a.moving_var.cpu().numpy(), bn_b.moving_var.cpu().numpy(), rtol=1e-5)
    assert ops.stem_conv_fwd_num_parts(N, H, W, 24, torch.bfloat16) == 0      # 3 pieces per pixel: not fused


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("M,C,act", [(4096, 128, 1), (777, 32, 2), (50000, 64, 2), (300, 1024, 0)])
def test_bn_stats_finalize_apply(cuda, dtype, M, C, act):
    ops = _ops()
    rs = np.random.RandomState(C)
    x = rnd(rs.randn(1, 1, M, C) * 2 + 0.7, dtype)
    g, b = rs.rand(C).astype(np.float32) + 0.5, rs.randn(C).astype(np.float32)
    mm, mv = rs.randn(C).astype(np.float32), rs.rand(C).astype(np.float32) + 0.5
    bn = ops.BNState(dev(g), dev(b), dev(mm), dev(mv), act)
    dx = dev(x, dtype)
    part, nparts = ops.bn_stats(dx)
    ops.bn_finalize(bn, part, nparts, M, training=True)
    xd = x.double().reshape(M, C)
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    np.testing.assert_allclose(bn.mean.cpu().numpy(), mean.numpy(), atol=2e-6 * 10)
    np.testing.assert_allclose(bn.invstd.cpu().numpy(), (1 / torch.sqrt(var + 1e-3)).numpy(), rtol=2e-5)
    np.testing.assert_allclose(bn.moving_mean.cpu().numpy(), mm * 0.95 + mean.numpy() * 0.05, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.moving_var.cpu().numpy(), mv * 0.95 + var.numpy() * M / (M - 1) * 0.05, rtol=1e-5)
    y = ops.bn_act_apply(dx, bn.affine)
    want = act_ref((xd - mean) / torch.sqrt(var + 1e-3) * torch.tensor(g).double() + torch.tensor(b).double(), act)
    assert_close(y.reshape(M, C), want, dtype)
    # inference affine
    bn2 = ops.BNState(dev(g), dev(b), dev(mm), dev(mv), act)
    ops.bn_inference_affine(bn2)
    np.testing.assert_allclose(bn2.scale.cpu().numpy(), g / np.sqrt(mv + 1e-3), rtol=1e-5)
    np.testing.assert_allclose(bn2.shift.cpu().numpy(), b - mm * g / np.sqrt(mv + 1e-3), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("u", [1, 2, 4, 8])
def test_bilinear_into_concat_slice(cuda, dtype, u):
    ops = _ops()
    rs = np.random.RandomState(u)
    N, h, w, C = 2, 6, 10, 128
    x = rnd(rs.randn(N, h, w, C), dtype)
    sc = torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)
    sh = torch.tensor(rs.randn(C) * 0.5, dtype=torch.float32)
    a = torch.relu(x * sc + sh)
    want = nhwc(onet.resize_bilinear_legacy(nchw(a), h * u, w * u))
    out = torch.zeros((N, h * u, w * u, 512), dtype=dtype, device="cuda")
    ops.bilinear_up_fwd(dev(x, dtype), u, out, 256, ops.Affine(dev(sc), dev(sh), 1))
    assert_close(out[..., 256:384], want, dtype)
    assert float(out[..., :256].abs().max()) == 0 and float(out[..., 384:].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_sumpool2x2(cuda, dtype):
    ops = _ops()
    rs = np.random.RandomState(0)
    src = rnd(rs.randn(2, 8, 12, 128), dtype)
    dst0 = rnd(rs.randn(2, 4, 6, 128), dtype)
    want = F.avg_pool2d(nchw(src), 2) * 4
    got = ops.sumpool2x2(dev(src, dtype))
    assert_close(got, nhwc(want), dtype)
    d = dev(dst0, dtype)
    ops.sumpool2x2(dev(src, dtype), d, accumulate=True)
    assert_close(d, nhwc(want) + dst0, dtype)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("M,C", [(256 * 3, 64), (1000, 64), (777, 32), (300, 16)])
def test_heatmap_head_fwd(cuda, dtype, M, C):
    """f32 tolerance in both builds: the bf16 build's matrix-core kernel (C = 16 / 32 / 64) splits activations and weights
    into bf16 hi + lo parts."""
    ops = _ops()
    rs = np.random.RandomState(M)
    x = rnd(rs.randn(1, 1, M, C), dtype)
    w = (rs.randn(1, 1, C, 18) * 0.1).astype(np.float32)
    b = rs.randn(18).astype(np.float32)
    sc = torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)
    sh = torch.tensor(rs.randn(C) * 0.5, dtype=torch.float32)
    a = torch.relu(x * sc + sh).reshape(M, C)
    want = a @ torch.tensor(w).reshape(C, 18) + torch.tensor(b)
    aff = ops.Affine(dev(sc), dev(sh), 1)
    got = ops.heatmap_head_fwd(dev(x, dtype), dev(w), dev(b), aff)
    assert_close(got.reshape(M, 18), want, torch.float32, 64)
    hm, seg = ops.heatmap_head_fwd(dev(x, dtype), dev(w), dev(b), aff, inference=True)
    assert_close(hm.reshape(M, 17), torch.sigmoid(want[:, :17]), torch.float32, 64)
    assert_close(seg.reshape(M), want[:, 17], torch.float32, 64)


def _labels(rs, B, h, w):
    hm = (rs.rand(B, h, w, 17) * 0.9).astype(np.float32)
    for b in range(B):
        for _ in range(6):
            hm[b, rs.randint(h), rs.randint(w), rs.randint(17)] = 1.0
    return {"heatmaps": hm, "loss_masks": (rs.rand(B, h, w) < 0.9).astype(np.float32),
            "segmentation_masks": (rs.rand(B, h, w) < 0.3).astype(np.float32),
            "num_boxes": rs.randint(0, 5, B).astype(np.int32)}


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_keypoint_loss_and_gradients(cuda, dtype):
    ops = _ops()
    rs = np.random.RandomState(5)
    B, h, w = 2, 16, 24
    lab = _labels(rs, B, h, w)
    logits = torch.tensor(rs.randn(B, h, w, 18).astype(np.float32) * 2, requires_grad=True)
    ps = [rnd(rs.randn(B, h >> k, w >> k, 128), dtype).requires_grad_(True) for k in range(4)]
    tl = {k: torch.tensor(v) for k, v in lab.items()}
    total, losses = onet.losses_fn(logits, {f"p{l}": ps[l - 2] for l in range(2, 6)}, tl)
    total.backward()
    dl = torch.empty((B, h, w, 18), device="cuda")
    daux = [torch.empty((B, h >> k, w >> k), device="cuda") for k in range(4)]
    dlab = {k: dev(v) for k, v in lab.items()}
    out = ops.keypoint_loss(dev(logits.detach()), dlab, [dev(p.detach(), dtype) for p in ps], dl, daux).cpu().numpy()
    want = [float(v) for v in losses.values()] + [float(total)]
    np.testing.assert_allclose(out[:7], want, rtol=2e-5)
    np.testing.assert_allclose(out[7], float(onet.per_pixel_reg_loss(logits.detach(), tl)), rtol=2e-5)
    assert_close(dl, logits.grad, torch.float32, scale=float(logits.grad.abs().max()))
    for k in range(4):
        assert_close(daux[k], ps[k].grad[..., 0], torch.float32, scale=float(ps[k].grad.abs().max()) or 1.0)


def test_adam_matches_tf_semantics(cuda):
    ops = _ops()
    rs = np.random.RandomState(9)
    n = 4096
    p = rs.randn(n).astype(np.float32); g = (rs.randn(n) * 100).astype(np.float32); g[:8] = 1e4
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    dp, dg, dm, dv = dev(p), dev(g), dev(m), dev(v)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    hyper = torch.zeros(4, device="cuda")
    p64, m64, v64 = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for it in range(3):
        ops.adam_prepare(step, hyper, 3e-4, 200000)
        ops.adam_step(dp, dg, dm, dv, hyper, grad_scale=0.5)
        lr = onet.cosine_decay(3e-4, it, 200000)
        onet.adam_step(p64, g.astype(np.float64) * 0.5, m64, v64, lr, it + 1)
        assert int(step.item()) == it + 1
        np.testing.assert_allclose(float(hyper[1]), lr, rtol=1e-6)
        np.testing.assert_allclose(dp.cpu().numpy(), p64, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(dv.cpu().numpy(), v64, rtol=1e-4)  # (1-beta2) is rounded in f32, as in TF


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_conv_fwd_grouped_equals_separate_launches(cuda, dtype):
    """mpn_conv_fwd_grouped: four independent 3x3 convolutions (pyramid levels) in one grid = the four separate launches:
    outputs bit for bit, statistics the same row count and the same sums to f32 rounding of a block's partial sums (f32 takes
    the documented fallback: the separate launches themselves)."""
    ops = _ops()
    rs = np.random.RandomState(17)
    N, C = 2, 128
    sizes = [(32, 48), (16, 24), (8, 12), (5, 7)]
    xs, pcs, affs = [], [], []
    for (h, w) in sizes:
        xs.append(dev(rnd(rs.randn(N, h, w, C), dtype), dtype))
        pcs.append(ops.PackedConv(dev((rs.randn(3, 3, C, C) / np.sqrt(9 * C)).astype(np.float32)), dtype))
        affs.append(ops.Affine(dev(torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)), dev(torch.tensor(rs.randn(C) * 0.5, dtype=torch.float32)), 1))
    want, wparts = [], []
    for x, pc, a, (h, w) in zip(xs, pcs, affs, sizes):
        part = torch.full((ops.conv_num_parts(N, h, w, 3), 2, C), float("nan"), device="cuda")
        want.append(ops.conv_fwd(x, pc.fwd, C, 3, a, stats_part=part))
        wparts.append(part)
    outs = [torch.full_like(t, float("nan")) for t in want]
    parts = [torch.full_like(t, float("nan")) for t in wparts]
    ops.conv_fwd_grouped(xs, [pc.fwd for pc in pcs], C, 3, affs, outs, parts)
    for a, b, pa, pb, (h, w) in zip(want, outs, wparts, parts, sizes):
        assert torch.equal(a, b)
        # the same number of rows alone and in the group; which tiles a block sums differs, the totals agree to f32 rounding
        rows = ops.conv_stats_rows(N, h, w, C, C, 3, dtype)
        assert bool(torch.isnan(pa[rows:]).all()) and bool(torch.isnan(pb[rows:]).all())
        sa, sb = pa[:rows].double().sum(0), pb[:rows].double().sum(0)
        np.testing.assert_allclose(sa.cpu().numpy(), sb.cpu().numpy(), rtol=1e-5, atol=1e-4)
    # without affine / statistics (the data-gradient use)
    want2 = [ops.conv_fwd(x, pc.bwd, C, 3) for x, pc in zip(xs, pcs)]
    outs2 = [torch.empty_like(t) for t in want2]
    ops.conv_fwd_grouped(xs, [pc.bwd for pc in pcs], C, 3, [None] * 4, outs2, [None] * 4)
    for a, b in zip(want2, outs2):
        assert torch.equal(a, b)


def test_conv_fwd_grouped_statistics_rows_with_two_channel_tiles(cuda):
    """The persistent kernel's one-row-per-block statistics when a pixel tile has TWO channel tiles (Cout = 256) and the group has
    more tiles (368) than the device has compute units: blocks walk several tiles of a job and cross from one job into the next;
    every row mpn_conv_stats_rows promises is written whole (both channel tiles), nothing behind it, and the sums are those of
    the outputs."""
    ops = _ops()
    dtype = torch.bfloat16
    rs = np.random.RandomState(23)
    N, Cin, Cout = 4, 64, 256
    sizes = [(96, 96), (48, 48), (16, 16)]
    xs, pcs, affs, outs, parts = [], [], [], [], []
    for (h, w) in sizes:
        xs.append(dev(rnd(rs.randn(N, h, w, Cin), dtype), dtype))
        pcs.append(ops.PackedConv(dev((rs.randn(3, 3, Cin, Cout) / np.sqrt(9 * Cin)).astype(np.float32)), dtype))
        affs.append(ops.Affine(dev(torch.tensor(0.5 + rs.rand(Cin), dtype=torch.float32)), dev(torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32)), 1))
        outs.append(torch.full((N, h, w, Cout), float("nan"), device="cuda", dtype=dtype))
        parts.append(torch.full((ops.conv_num_parts(N, h, w, 3), 2, Cout), float("nan"), device="cuda"))
    ops.conv_fwd_grouped(xs, [pc.fwd for pc in pcs], Cout, 3, affs, outs, parts)
    for x, pc, a, y, part, (h, w) in zip(xs, pcs, affs, outs, parts, sizes):
        assert torch.equal(y, ops.conv_fwd(x, pc.fwd, Cout, 3, a))
        rows = ops.conv_stats_rows(N, h, w, Cin, Cout, 3, dtype)
        assert 0 < rows <= part.shape[0] and bool(torch.isfinite(part[:rows]).all()) and bool(torch.isnan(part[rows:]).all())
        s = part[:rows].double().sum(0).cpu()
        yd = y.double().reshape(-1, Cout).cpu()
        np.testing.assert_allclose(s[0].numpy(), yd.sum(0).numpy(), rtol=1e-5, atol=1e-2)
        np.testing.assert_allclose(s[1].numpy(), (yd * yd).sum(0).numpy(), rtol=1e-5, atol=1e-2)


def test_conv_fwd_channel_slices_of_wider_tensors(cuda):
    """x / y pixel strides: a conv reads a channel slice of a wider NHWC tensor and writes into a slice of another one
    (phi_subnet_2/conv2 -> the first 128 channels of the 512-channel concat tensor), 3x3 tiled kernel and 1x1 GEMM kernel."""
    ops = _ops()
    rs = np.random.RandomState(3)
    dtype = torch.bfloat16
    for (k, Cin, Cout, wide_in, wide_out, off_in, off_out) in [(3, 128, 128, 128, 512, 0, 128), (1, 256, 256, 384, 512, 128, 256),
                                                                 (1, 64, 128, 128, 128, 64, 0)]:
        N, H, W = 2, 16, 24
        xw = dev(rnd(rs.randn(N, H, W, wide_in), dtype), dtype)
        w = rs.randn(k, k, Cin, Cout).astype(np.float32) / np.sqrt(k * k * Cin)
        pc = ops.PackedConv(dev(w), dtype)
        sc = dev(torch.tensor(0.5 + rs.rand(Cin), dtype=torch.float32)); sh = dev(torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32))
        aff = ops.Affine(sc, sh, 1)
        xs = xw[..., off_in:off_in + Cin]
        part_a = torch.zeros((ops.conv_num_parts(N, H, W, k), 2, Cout), device="cuda")
        part_b = torch.zeros_like(part_a)
        want = ops.conv_fwd(xs.contiguous(), pc.fwd, Cout, k, aff, stats_part=part_a)
        yw = torch.full((N, H, W, wide_out), 7.0, device="cuda", dtype=dtype)
        ops.conv_fwd(xs, pc.fwd, Cout, k, aff, out=yw[..., off_out:off_out + Cout], stats_part=part_b)
        assert torch.equal(yw[..., off_out:off_out + Cout], want) and torch.equal(part_a, part_b)
        rest = torch.ones(wide_out, dtype=torch.bool); rest[off_out:off_out + Cout] = False
        assert bool((yw[..., rest.cuda()] == 7.0).all())      # nothing outside the slice is touched
