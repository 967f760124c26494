"""GPU: backward kernels (through the C ABI) vs torch autograd over the oracle's forward restatement."""
import numpy as np
import pytest
import torch

from oracle import network as onet
from util import DTYPES, act_ref, assert_close, dev, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu
IDS = ["f32", "bf16"]


def _ops():
    from multiposenet_amd import ops
    return ops


WGRAD_CASES = [
    # N, H, W, Cin, Cout, k, act
    (2, 16, 16, 128, 128, 3, 1),
    (1, 12, 20, 128, 128, 3, 0),     # partial tiles
    (2, 16, 16, 512, 64, 3, 0),      # final_conv3x3
    (1, 4, 4, 128, 128, 3, 1),
    (2, 16, 16, 32, 64, 1, 2),       # pointwise 1 (the thin 32 x 64 block tile)
    (1, 37, 29, 32, 64, 1, 1),       # ... ragged M, many tiles per split
    (2, 10, 6, 16, 32, 1, 2),        # ... at depth_multiplier 0.5
    (2, 16, 16, 256, 128, 1, 2),     # lateral
    (3, 16, 16, 1024, 1024, 1, 2),   # pointwise 13
    (1, 10, 6, 64, 128, 1, 2),       # ragged M
    (40, 16, 16, 128, 128, 3, 1),    # many tiles per split
]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[f"{c[3]}to{c[4]}k{c[5]}_{c[0]}x{c[1]}x{c[2]}" for c in WGRAD_CASES])
def test_conv_wgrad(cuda, dtype, case):
    ops = _ops()
    N, H, W, Cin, Cout, k, act = case
    rs = np.random.RandomState(Cin + k + N)
    x = rnd(rs.randn(N, H, W, Cin), dtype)
    dy = rnd(rs.randn(N, H, W, Cout), dtype)
    aff = None
    a = x
    if act:
        sc = torch.tensor(0.5 + rs.rand(Cin), dtype=torch.float32)
        sh = torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32)
        aff = ops.Affine(dev(sc), dev(sh), act)
        a = rnd(act_ref(x * sc + sh, act), dtype)
    w = torch.zeros(k, k, Cin, Cout, requires_grad=True)
    out = onet.conv2d_same(nchw(a), w)
    out.backward(nchw(dy))
    dw = torch.full((k, k, Cin, Cout), float("nan"), device="cuda")
    ops.conv_bwd_weight(dev(x, dtype), dev(dy, dtype), k, aff, dw)
    # sum over N*H*W products of O(1) values
    assert_close(dw, w.grad, torch.float32 if dtype == torch.float32 else dtype, N * H * W,
                 scale=float(w.grad.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("N,H,W,C,stride", [(2, 16, 16, 32, 1), (2, 16, 16, 64, 2), (1, 12, 20, 128, 1), (1, 14, 10, 256, 2),
                                            (1, 9, 7, 64, 2), (2, 8, 8, 1024, 1), (1, 32, 32, 512, 2),
                                            (1, 9, 7, 64, 1), (2, 5, 13, 32, 1), (1, 70, 33, 128, 1)])   # odd widths, strips of 64 rows
def test_dwconv_backward(cuda, dtype, N, H, W, C, stride):
    ops = _ops()
    rs = np.random.RandomState(C + stride)
    x = rnd(rs.randn(N, H, W, C), dtype)
    wnp = (rs.randn(3, 3, C, 1) / 3).astype(np.float32)
    sc = torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)
    sh = torch.tensor(rs.randn(C) * 0.5, dtype=torch.float32)
    a = torch.clamp(x * sc + sh, 0, 6).requires_grad_(True)
    w = torch.tensor(wnp, requires_grad=True)
    out = onet.depthwise_conv2d_tf_same(nchw(a), w, stride)
    dy = rnd(rs.randn(*nhwc(out).shape), dtype)
    out.backward(nchw(dy))
    da = ops.dwconv_bwd_data(dev(dy, dtype), dev(wnp), (H, W), stride)
    assert_close(da, a.grad, dtype, 9)
    dw = torch.full((3, 3, C, 1), float("nan"), device="cuda")
    ops.dwconv_bwd_weight(dev(x, dtype), dev(dy, dtype), stride, ops.Affine(dev(sc), dev(sh), 2), dw)
    assert_close(dw, w.grad, torch.float32, N * H * W, scale=float(w.grad.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("C0", [32, 16, 64])
@pytest.mark.parametrize("N,H,W,u8", [(2, 32, 32, False), (1, 64, 48, False), (1, 30, 34, False), (3, 128, 128, False), (2, 70, 34, True)])
def test_stem_wgrad(cuda, dtype, N, H, W, u8, C0):
    """f32 tolerance in both builds: the bf16 build's matrix-core kernel splits the image patch into bf16 hi + lo parts.
    C0 = 16 (depth_multiplier 0.5) once overran the VALU kernel's phase buffer in LDS."""
    ops = _ops()
    rs = np.random.RandomState(H)
    if u8:
        img8 = rs.randint(0, 256, (N, H, W, 3)).astype(np.uint8)
        img = torch.tensor(img8.astype(np.float32) * np.float32(1 / 255.0))
    else:
        img = torch.tensor(rs.rand(N, H, W, 3).astype(np.float32))
    w = torch.zeros(3, 3, 3, C0, requires_grad=True)
    out = onet.conv2d_tf_same(nchw(2.0 * img - 1.0), w, 2)
    dy = rnd(rs.randn(*nhwc(out).shape), dtype)
    out.backward(nchw(dy))
    dw = torch.full((3, 3, 3, C0), float("nan"), device="cuda")
    if C0 == 64 and dtype == torch.float32:       # the f32 build's VALU kernel keeps a 256 x C0 f32 tile in 64 KB of LDS
        with pytest.raises(ValueError, match="C0 too large"):
            ops.stem_conv_bwd_weight(dev(img), dev(dy, dtype), dw)
        return
    ops.stem_conv_bwd_weight(dev(img8) if u8 else dev(img), dev(dy, dtype), dw)
    assert_close(dw, w.grad, torch.float32, N * H * W // 4, scale=float(w.grad.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("M,C,act,ch0", [(4096, 128, 1, True), (777, 32, 2, False), (50000, 64, 2, False), (300, 1024, 0, False)])
def test_bn_backward(cuda, dtype, M, C, act, ch0):
    ops = _ops()
    rs = np.random.RandomState(C + 1)
    x = rnd(rs.randn(1, 1, M, C) * 1.5 + 0.4, dtype)
    dA = rnd(rs.randn(1, 1, M, C), dtype)
    g, b = rs.rand(C).astype(np.float32) + 0.5, rs.randn(C).astype(np.float32)
    bn = ops.BNState(dev(g), dev(b), dev(np.zeros(C, np.float32)), dev(np.ones(C, np.float32)), act)
    bn.dgamma, bn.dbeta = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx = dev(x, dtype)
    part, nparts = ops.bn_stats(dx)
    ops.bn_finalize(bn, part, nparts, M)
    # oracle: autograd through act(batch_norm(x)) with gamma/beta as leaves
    xt = x.clone().double().reshape(1, M, 1, C).permute(0, 3, 1, 2).requires_grad_(True)   # NCHW [1,C,M,1]
    p = {"bn/gamma": torch.tensor(g).double().requires_grad_(True), "bn/beta": torch.tensor(b).double().requires_grad_(True),
         "bn/moving_mean": torch.zeros(C).double(), "bn/moving_variance": torch.ones(C).double()}
    y = act_ref(onet.batch_norm(xt, p, "bn", True), act)
    y.backward(dA.double().reshape(1, M, 1, C).permute(0, 3, 1, 2))
    want_dx = xt.grad.permute(0, 2, 3, 1).reshape(M, C)
    add = None
    if ch0:
        a0 = rs.randn(M).astype(np.float32)
        add = dev(a0)
        want_dx = want_dx.clone()
        want_dx[:, 0] += torch.tensor(a0).double()
    dAd = dev(dA, dtype)
    nb = ops._lib.lib().mpn_bn_stats_num_parts(M)
    part2 = torch.empty(nb * 2 * C, device="cuda")
    ops.bn_backward(bn, dAd, dx, part2, add_ch0=add)
    assert_close(dAd.reshape(M, C), want_dx, dtype, scale=float(want_dx.abs().max()))
    np.testing.assert_allclose(bn.dgamma.cpu().numpy(), p["bn/gamma"].grad.numpy(), rtol=2e-3, atol=2e-3 * float(p["bn/gamma"].grad.abs().max()))
    np.testing.assert_allclose(bn.dbeta.cpu().numpy(), p["bn/beta"].grad.numpy(), rtol=2e-3, atol=2e-3 * float(p["bn/beta"].grad.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("u", [1, 2, 4, 8])
@pytest.mark.parametrize("N,h,w,C", [(2, 6, 10, 128), (1, 3, 37, 64), (1, 4, 5, 24), (1, 2, 260, 8)],
                         ids=["128ch", "wide-64ch", "24ch", "rows-wider-than-a-walk"])
def test_bilinear_backward(cuda, dtype, u, N, h, w, C):
    """The transposed legacy resize on a channel slice of the 512-channel concat gradient (one partial block per row, several
    blocks per row, a channel count that is not a power of two) against autograd through the oracle's forward."""
    ops = _ops()
    rs = np.random.RandomState(u)
    a = torch.zeros(N, C, h, w, requires_grad=True)
    out = onet.resize_bilinear_legacy(a, h * u, w * u)
    dyfull = rnd(rs.randn(N, h * u, w * u, 512), dtype)
    out.backward(nchw(dyfull[..., 128:128 + C]))
    got = ops.bilinear_up_bwd(dev(dyfull, dtype), u, 128, C)
    assert_close(got, nhwc(a.grad), dtype, 4 * u * u)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("u", [4, 8])
@pytest.mark.parametrize("N,h", [(40, 19), (150, 16)], ids=["strips-of-2-rows-last-of-1", "strips-of-4-rows"])
def test_bilinear_backward_walks_strips_of_several_input_rows(cuda, dtype, u, N, h):
    """Enough images that the walking kernel (upsample 4 and 8) takes strips of R > 1 input rows - the subnet's real geometry
    has R = 2 and 4 - so the sum carried from one input row of a strip into the next, the U - 1 rows above a strip and a last
    strip that is shorter than the others are all on the path; against autograd through the oracle's forward."""
    ops = _ops()
    w, C, ctot, coff = 4, 8, 16, 8
    rs = np.random.RandomState(u + h)
    a = torch.zeros(N, C, h, w, requires_grad=True)
    out = onet.resize_bilinear_legacy(a, h * u, w * u)
    dyfull = rnd(rs.randn(N, h * u, w * u, ctot), dtype)
    out.backward(nchw(dyfull[..., coff:coff + C]))
    got = ops.bilinear_up_bwd(dev(dyfull, dtype), u, coff, C)
    assert_close(got, nhwc(a.grad), dtype, 4 * u * u)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("M", [128 * 5, 1000, 128 * 700])
def test_heatmap_head_backward(cuda, dtype, M):
    ops = _ops()
    rs = np.random.RandomState(M % 1000)
    x = rnd(rs.randn(1, 1, M, 64), dtype)
    wnp = (rs.randn(1, 1, 64, 18) * 0.1).astype(np.float32)
    sc = torch.tensor(0.5 + rs.rand(64), dtype=torch.float32)
    sh = torch.tensor(rs.randn(64) * 0.5, dtype=torch.float32)
    a = torch.relu(x * sc + sh).reshape(M, 64).requires_grad_(True)
    w = torch.tensor(wnp).reshape(64, 18).requires_grad_(True)
    b = torch.zeros(18, requires_grad=True)
    out = a @ w + b
    dl = torch.tensor(rs.randn(M, 18).astype(np.float32))
    out.backward(dl)
    dA = torch.empty((1, 1, M, 64), dtype=dtype, device="cuda")
    dwdb = torch.full((64 * 18 + 18,), float("nan"), device="cuda")
    ops.heatmap_head_bwd(dev(x, dtype), dev(dl).reshape(1, 1, M, 18), dev(wnp), ops.Affine(dev(sc), dev(sh), 1), dA, dwdb)
    assert_close(dA.reshape(M, 64), a.grad, dtype, 18)
    assert_close(dwdb[:64 * 18].reshape(64, 18), w.grad, torch.float32, M, scale=float(w.grad.abs().max()))
    assert_close(dwdb[64 * 18:], b.grad, torch.float32, M, scale=float(b.grad.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("N,H,W,C,stride", [(2, 16, 16, 32, 1), (2, 16, 16, 64, 2), (1, 12, 20, 128, 1), (1, 32, 32, 512, 2),
                                            (2, 8, 8, 1024, 1), (1, 48, 40, 256, 2), (1, 9, 7, 64, 1), (1, 70, 33, 128, 1)])
def test_dwconv_bwd_data_with_fused_bn_reduction(cuda, dtype, N, H, W, C, stride):
    """mpn_dwconv_bwd_data_bn: the same dx as mpn_dwconv_bwd_data, and partial rows that finalize to the dgamma / dbeta
    of mpn_bn_bwd_reduce on (dx, x_bn)."""
    ops = _ops()
    rs = np.random.RandomState(C + stride + H)
    OH, OW = ops.dwconv_out_hw(H, W, stride)
    dy = dev(rnd(rs.randn(N, OH, OW, C), dtype), dtype)
    w = dev((rs.randn(3, 3, C) / 3).astype(np.float32))
    xbn = dev(rnd(rs.randn(N, H, W, C), dtype), dtype)
    def mkbn():
        one = lambda: torch.tensor((0.5 + rs.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), 2)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((rs.randn(C) * 0.5).astype(np.float32)).cuda()); bn.mean.copy_(torch.tensor((rs.randn(C) * 0.3).astype(np.float32)).cuda())
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn
    st = rs.get_state()
    bn_a = mkbn(); rs.set_state(st); bn_b = mkbn()
    rows = ops.dwconv_bwd_data_bn_num_parts(N, H, W, C, stride, dy.dtype)
    assert rows > 0
    want = ops.dwconv_bwd_data(dy, w, (H, W), stride)
    got, r2 = ops.dwconv_bwd_data(dy, w, (H, W), stride, bn=bn_a, x_bn=xbn)
    assert r2 == rows
    assert torch.equal(got, want)
    # reference: separate reduction of the same dA
    M = N * H * W
    part = torch.empty(ops._lib.lib().mpn_bn_stats_num_parts(M) * 2 * C, device="cuda")
    dA_sep = want.clone()
    ops.bn_backward(bn_b, dA_sep, xbn, part)
    part2 = torch.empty(rows * 2 * C, device="cuda")
    dA_fused, _ = ops.dwconv_bwd_data(dy, w, (H, W), stride, bn=bn_a, x_bn=xbn, part=part2)
    ops.bn_backward(bn_a, dA_fused, xbn, part2, reduced_parts=rows)
    scale = float(bn_b.dgamma.abs().max()) + 1e-6
    assert float((bn_a.dgamma - bn_b.dgamma).abs().max()) <= 2e-5 * scale * max(1.0, M ** 0.5 / 16)
    assert float((bn_a.dbeta - bn_b.dbeta).abs().max()) <= 2e-5 * (float(bn_b.dbeta.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 16)
    assert_close(dA_fused, dA_sep.float().cpu(), dtype, 4)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 64), (1, 32, 32, 512), (1, 48, 40, 256), (2, 64, 64, 128), (1, 6, 10, 1024)])
def test_dwconv_bwd_data_add(cuda, dtype, N, H, W, C):
    """mpn_dwconv_bwd_data_add (stride 2): dx = round(gradient + addend) - the f32 gradient of mpn_dwconv_bwd_data plus the
    addend with ONE rounding (within one storage ulp of the two-pass result, which rounds twice); with the batch-norm
    reduction the partial rows finalize to the dgamma / dbeta of mpn_bn_bwd_reduce on that sum."""
    ops = _ops()
    rs = np.random.RandomState(C + H)
    OH, OW = ops.dwconv_out_hw(H, W, 2)
    assert ops.dwconv_bwd_data_add_supported(N, H, W, C, 2, dtype)
    assert not ops.dwconv_bwd_data_add_supported(N, H, W, C, 1, dtype)
    assert not ops.dwconv_bwd_data_add_supported(N, H + 1, W, C, 2, dtype)
    dy = dev(rnd(rs.randn(N, OH, OW, C), dtype), dtype)
    w = dev((rs.randn(3, 3, C) / 3).astype(np.float32))
    add = dev(rnd(rs.randn(N, H, W, C), dtype), dtype)
    xbn = dev(rnd(rs.randn(N, H, W, C), dtype), dtype)
    g32 = ops.dwconv_bwd_data(dy.float(), w, (H, W), 2)            # the f32 kernel on the same (storage-rounded) values
    want = (g32 + add.float()).cpu()
    got = ops.dwconv_bwd_data(dy, w, (H, W), 2, addend=add)
    assert_close(got, want, dtype, 1)
    two_pass = ops.add_inplace(ops.dwconv_bwd_data(dy, w, (H, W), 2), add)
    assert_close(got, two_pass.float().cpu(), dtype, 2)
    with pytest.raises(Exception):
        ops.dwconv_bwd_data(dy, w, (H, W), 2, out=add, addend=add)   # aliasing is refused

    def mkbn():
        one = lambda: torch.tensor((0.5 + rs.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), 2)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((rs.randn(C) * 0.5).astype(np.float32)).cuda()); bn.mean.copy_(torch.tensor((rs.randn(C) * 0.3).astype(np.float32)).cuda())
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn
    st = rs.get_state()
    bn_a = mkbn(); rs.set_state(st); bn_b = mkbn()
    rows = ops.dwconv_bwd_data_bn_num_parts(N, H, W, C, 2, dy.dtype)
    part2 = torch.empty(rows * 2 * C, device="cuda")
    dA_fused, r2 = ops.dwconv_bwd_data(dy, w, (H, W), 2, bn=bn_a, x_bn=xbn, part=part2, addend=add)
    assert r2 == rows and torch.equal(dA_fused, got)
    M = N * H * W
    part = torch.empty(ops._lib.lib().mpn_bn_stats_num_parts(M) * 2 * C, device="cuda")
    dA_sep = got.clone()
    ops.bn_backward(bn_b, dA_sep, xbn, part)
    ops.bn_backward(bn_a, dA_fused, xbn, part2, reduced_parts=rows)
    scale = float(bn_b.dgamma.abs().max()) + 1e-6
    assert float((bn_a.dgamma - bn_b.dgamma).abs().max()) <= 2e-5 * scale * max(1.0, M ** 0.5 / 16)
    assert float((bn_a.dbeta - bn_b.dbeta).abs().max()) <= 2e-5 * (float(bn_b.dbeta.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 16)
    assert_close(dA_fused, dA_sep.float().cpu(), dtype, 4)


def test_batched_bn_finalizes_equal_per_layer_launches(cuda):
    """mpn_bn_finalize_batched / mpn_bn_bwd_finalize_batched: several layers in one launch, bit for bit the per-layer results
    (scale, shift, saved mean / invstd, moving statistics; dgamma, dbeta, k1, k2)."""
    ops = _ops()
    rs = np.random.RandomState(31)

    def mk(C):
        one = lambda: torch.tensor((0.5 + rs.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), 2)
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn

    # (C, nparts, count); below the 4096 rows from which the per-layer finalizes compact the slab in place first
    shapes = [(128, 4000, 512000), (128, 37, 1000), (64, 1024, 8192), (24, 5, 77)]
    parts = [torch.tensor(rs.randn(n, 2, C).astype(np.float32)).cuda() for (C, n, _) in shapes]
    st = rs.get_state()
    a = [mk(C) for (C, _, _) in shapes]
    rs.set_state(st)
    b = [mk(C) for (C, _, _) in shapes]
    for bn, part, (C, n, cnt) in zip(a, parts, shapes):
        ops.bn_finalize(bn, part, n, cnt)
        call = ops.call
        call("mpn_bn_bwd_finalize", ops.ptr(part), n, C, cnt, ops.ptr(bn.dgamma), ops.ptr(bn.dbeta), ops.ptr(bn.k1), ops.ptr(bn.k2),
             ops.stream_ptr())
    jobs = [(bn, part, n, cnt) for bn, part, (C, n, cnt) in zip(b, parts, shapes)]
    ops.BnFinalizeBatch(jobs, "cuda:0").run()
    ops.BnBwdFinalizeBatch(jobs, "cuda:0").run()
    for x, y in zip(a, b):
        for f in ("scale", "shift", "mean", "invstd", "moving_mean", "moving_var", "dgamma", "dbeta", "k1", "k2"):
            assert torch.equal(getattr(x, f), getattr(y, f)), f


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_grouped_bn_backward_passes_equal_per_layer_launches(cuda, dtype):
    """mpn_bn_bwd_reduce_grouped / mpn_bn_bwd_apply_grouped: four layers (pyramid levels) per grid, bit for bit the per-layer
    partial rows, dgamma / dbeta and dx (with the extra channel-0 gradient on some jobs)."""
    ops = _ops()
    rs = np.random.RandomState(23)
    C = 128
    Ms = [2 * 32 * 48, 2 * 16 * 24, 2 * 8 * 12, 77]

    def mk():
        one = lambda: torch.tensor((0.5 + rs.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), 1)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((rs.randn(C) * 0.5).astype(np.float32)).cuda())
        bn.mean.copy_(torch.tensor((rs.randn(C) * 0.3).astype(np.float32)).cuda())
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn

    xs = [dev(rnd(rs.randn(M, C), dtype), dtype) for M in Ms]
    dAs = [dev(rnd(rs.randn(M, C), dtype), dtype) for M in Ms]
    add = [torch.tensor(rs.randn(M).astype(np.float32)).cuda() if j % 2 == 0 else None for j, M in enumerate(Ms)]
    st = rs.get_state()
    a = [mk() for _ in Ms]
    rs.set_state(st)
    b = [mk() for _ in Ms]
    nparts = [ops._lib.lib().mpn_bn_stats_num_parts(M) for M in Ms]
    pa = [torch.full((n * 2 * C,), float("nan"), device="cuda") for n in nparts]
    pb = [torch.full((n * 2 * C,), float("nan"), device="cuda") for n in nparts]
    da_a = [t.clone() for t in dAs]
    da_b = [t.clone() for t in dAs]
    for bn, d, x, p, ad in zip(a, da_a, xs, pa, add):
        ops.bn_backward(bn, d, x, p, add_ch0=ad)
    ops.bn_bwd_reduce_grouped(b, da_b, xs, pb)
    ops.BnBwdFinalizeBatch([(bn, p, n, M) for bn, p, n, M in zip(b, pb, nparts, Ms)], "cuda:0").run()
    ops.bn_bwd_apply_grouped(b, da_b, xs, add)
    for j in range(len(Ms)):
        assert torch.equal(pa[j], pb[j]), j
        assert torch.equal(a[j].dgamma, b[j].dgamma) and torch.equal(a[j].dbeta, b[j].dbeta), j
        assert torch.equal(da_a[j], da_b[j]), j


def test_grouped_bn_backward_and_conv_wgrad_on_channel_slices(cuda):
    """Row strides: the grouped batch-norm backward passes and the conv weight gradient on tensors that live in channel
    slices of wider ones (level 2 of the concat tensor and of its gradient) = the same calls on dense copies, bit for bit."""
    ops = _ops()
    rs = np.random.RandomState(5)
    dtype = torch.bfloat16
    N, H, W, C, WIDE = 2, 16, 24, 128, 512
    xw = dev(rnd(rs.randn(N, H, W, WIDE), dtype), dtype)
    dw_ = dev(rnd(rs.randn(N, H, W, WIDE), dtype), dtype)
    x_s, d_s = xw[..., :C], dw_[..., :C]
    x_d, d_d = x_s.contiguous(), d_s.contiguous()

    def mk():
        r2 = np.random.RandomState(9)
        one = lambda: torch.tensor((0.5 + r2.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), 1)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((r2.randn(C) * 0.5).astype(np.float32)).cuda())
        bn.mean.copy_(torch.tensor((r2.randn(C) * 0.3).astype(np.float32)).cuda())
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn
    M = N * H * W
    npart = ops._lib.lib().mpn_bn_stats_num_parts(M)
    res = []
    for (d, x) in ((d_d, x_d), (d_s, x_s)):
        bn = mk()
        part = torch.zeros(npart * 2 * C, device="cuda")
        ops.bn_bwd_reduce_grouped([bn], [d], [x], [part])
        ops.BnBwdFinalizeBatch([(bn, part, npart, M)], "cuda:0").run()
        ops.bn_bwd_apply_grouped([bn], [d], [x])
        res.append((part.clone(), bn.dgamma.clone(), bn.dbeta.clone(), d.contiguous().clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # conv weight gradient with a strided dy (the slice holds the gradient w.r.t. the raw conv output now)
    a_in = dev(rnd(rs.randn(N, H, W, C), dtype), dtype)
    dW_a, dW_b = torch.zeros(3, 3, C, C, device="cuda"), torch.zeros(3, 3, C, C, device="cuda")
    ops.conv_bwd_weight(a_in, d_s.contiguous(), 3, None, dW_a)
    ops.conv_bwd_weight(a_in, d_s, 3, None, dW_b)
    assert torch.equal(dW_a, dW_b)
    ops.conv_bwd_weight(x_s.contiguous(), d_d, 1, None, dW_a[0, 0])   # strided x, 1x1
    ops.conv_bwd_weight(x_s, d_d, 1, None, dW_b[0, 0])
    assert torch.equal(dW_a, dW_b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "fp16", "f32"])
@pytest.mark.parametrize("k,Cin,Cout", [(3, 128, 128), (1, 256, 128)])
def test_conv_wgrad_grouped_equals_separate_launches(cuda, dtype, k, Cin, Cout):
    """mpn_conv_bwd_weight_grouped: the weight gradients of four independent layers (pyramid levels) from ONE grid = the four
    separate launches up to the f32 summation order (the grid divides its blocks among the jobs: other split counts)."""
    ops = _ops()
    rs = np.random.RandomState(23 + k)
    N = 4
    sizes = [(40, 48), (20, 24), (10, 12), (5, 6)]
    xs = [dev(rnd(rs.randn(N, h, w, Cin), dtype), dtype) for h, w in sizes]
    dys = [dev(rnd(rs.randn(N, h, w, Cout), dtype), dtype) for h, w in sizes]
    affs = [ops.Affine(dev(torch.tensor(0.5 + rs.rand(Cin), dtype=torch.float32)), dev(torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32)), 1)
            for _ in sizes]
    want = []
    for x, dy, a in zip(xs, dys, affs):
        dw = torch.full((k, k, Cin, Cout), float("nan"), device="cuda")
        ops.conv_bwd_weight(x, dy, k, a, dw)
        want.append(dw)
    nps = ops.conv_wgrad_grouped_num_parts(N, sizes, Cin, Cout, k, dtype)
    n = k * k * Cin * Cout
    parts = [torch.full((np_ * n,), float("nan"), device="cuda") for np_ in nps]
    ops.conv_bwd_weight_grouped(xs, dys, k, affs, parts)
    if dtype != torch.float32:
        assert sum(nps) * max(1, Cin // (64 if k == 3 else 128)) * max(1, Cout // 128) <= 256 + 8 * len(sizes)   # one grid of ~256 blocks
    for w_, part, np_ in zip(want, parts, nps):
        got = torch.empty(n, device="cuda")
        ops.reduce_partials(part, np_, n, got)
        scale = float(w_.abs().max())
        assert float((got.view_as(w_) - w_).abs().max()) <= 2e-5 * scale * (N * sizes[0][0] * sizes[0][1]) ** 0.5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("act,K,C", [(1, 128, 128), (2, 128, 128), (0, 128, 128), (1, 64, 64), (1, 128, 64), (2, 64, 192), (1, 128, 256),
                                          (1, 8, 64), (1, 24, 64), (2, 24, 72)],
                         ids=["relu", "relu6", "none", "relu-64to64", "relu-128to64", "relu6-64to192", "relu-128to256",
                              "thin-8to64", "thin-24to64", "thin-24to72"])
def test_conv_dgrad_with_fused_bn_reduction(cuda, dtype, act, K, C):
    """mpn_conv_bwd_data_bn_grouped on three jobs (ragged tiles, a channel-slice raw tensor): dx = the plain data gradient
    masked by the fed layer's activation, BIT FOR BIT; the slab's sums = sum g and sum g * x over all pixels (f64 reference
    on the kernel's own rounded outputs); the raw finalize gives dgamma / dbeta / k1 / k2 of mpn_bn_bwd_reduce + finalize."""
    from multiposenet_amd import ops
    rs = np.random.RandomState(5 + act + K + C)
    N = 2
    if K < 64 and dtype != torch.bfloat16:
        assert not ops.conv_bwd_data_bn_supported(K, C, 3, dtype)      # thin K: the tiled kernel, bf16 only
        pytest.skip("thin-K fused reduction is a bf16 path")
    assert ops.conv_bwd_data_bn_supported(K, C, 3, dtype) and not ops.conv_bwd_data_bn_supported(K, 68, 3, dtype)
    sizes = [(37, 21), (16, 16), (9, 5)]
    w = (rs.randn(3, 3, C, K) / np.sqrt(9 * C)).astype(np.float32)        # forward conv C -> K; its data gradient maps K -> C
    pc = ops.PackedConv(dev(w), dtype)
    dys, xs, bns, want_dx, outs, parts = [], [], [], [], [], []
    for j, (h, w_) in enumerate(sizes):
        dys.append(dev(rnd(rs.randn(N, h, w_, K), dtype), dtype))
        xw = dev(rnd(rs.randn(N, h, w_, C + 64) * 1.5 + 0.3, dtype), dtype)
        xs.append(xw[..., 32:32 + C] if j == 0 else xw[..., :C].contiguous())          # job 0: a channel slice (pixel stride C + 64)
        bn = ops.BNState(dev(torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)), dev(torch.tensor(rs.randn(C) * 0.3, dtype=torch.float32)),
                         torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), act)
        # batch statistics of x -> scale / shift / mean / invstd as the forward pass leaves them
        xf = xs[j].float().reshape(-1, C)
        mean, var = xf.mean(0), xf.var(0, unbiased=False)
        bn.mean.copy_(mean); bn.invstd.copy_(1.0 / torch.sqrt(var + 1e-3))
        bn.scale.copy_(bn.gamma * bn.invstd); bn.shift.copy_(bn.beta - mean * bn.scale)
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        bns.append(bn)
        want_dx.append(ops.conv_fwd(dys[j], pc.bwd, C, 3))
        outs.append(torch.full((N, h, w_, C), float("nan"), device="cuda", dtype=dtype))
        parts.append(torch.full((ops.conv_num_parts(N, h, w_, 3) * 2 * C,), float("nan"), device="cuda"))
    rows = ops.conv_bwd_data_bn_grouped(dys, [pc.bwd] * 3, C, bns, xs, outs, parts)
    for j, (h, w_) in enumerate(sizes):
        bn, x = bns[j], xs[j].float()
        pre = (x.double() * bn.scale.double() + bn.shift.double()).float()   # (the kernels' fused multiply-add: one rounding)
        ok = torch.ones_like(pre, dtype=torch.bool)
        if act != 0:
            ok = pre > 0
        if act == 2:
            ok = ok & (pre < 6)
        g = torch.where(ok, want_dx[j], torch.zeros_like(want_dx[j]))
        diff = (outs[j] != g)
        # an element whose pre-activation sits within one rounding of the threshold may flip with the FMA contraction: none expected
        assert int(diff.sum()) <= 2, int(diff.sum())
        assert bool(torch.isnan(parts[j][rows[j] * 2 * C:]).all())           # exactly the rows the op returned are written
        part = parts[j][:rows[j] * 2 * C].view(rows[j], 2, C).double().sum(0).cpu()
        gd, xd = outs[j].double().reshape(-1, C).cpu(), x.double().reshape(-1, C).cpu()
        np.testing.assert_allclose(part[0].numpy(), gd.sum(0).numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(part[1].numpy(), (gd * xd).sum(0).numpy(), rtol=1e-4, atol=2e-3)
    # finalize (raw) against reduce + finalize on the masked gradient
    cnts = [N * h * w_ for (h, w_) in sizes]
    fin = ops.BnBwdFinalizeBatch([(bns[j], parts[j], rows[j], cnts[j], True) for j in range(3)], "cuda")
    fin.run()
    got = [(b.dgamma.clone(), b.dbeta.clone(), b.k1.clone(), b.k2.clone()) for b in bns]
    for j in range(3):
        bn = bns[j]
        # f64 reference on the kernel's own masked gradient: dbeta = sum g, dgamma = sum g * xhat, k = sum / count
        gd, xd = outs[j].double().reshape(-1, C), xs[j].double().reshape(-1, C)
        xhat = (xd - bn.mean.double()) * bn.invstd.double()
        ref = ((gd * xhat).sum(0), gd.sum(0), gd.sum(0) / cnts[j], (gd * xhat).sum(0) / cnts[j])
        for a, b_, name in zip(got[j], ref, ("dgamma", "dbeta", "k1", "k2")):
            np.testing.assert_allclose(a.cpu().numpy(), b_.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(b_.abs().max() + 1e-6), err_msg=f"job {j} {name}")
        if dtype != torch.bfloat16:
            continue          # (the separate batch-norm passes take f32 / bf16 storage)
        p2 = torch.empty(ops._lib.lib().mpn_bn_stats_num_parts(cnts[j]) * 2 * C, device="cuda")
        xj = xs[j].contiguous()
        nparts = ops.bn_bwd_reduce(bn, outs[j], xj, p2)
        ops.call("mpn_bn_bwd_finalize", ops.ptr(p2), nparts, C, cnts[j], ops.ptr(bn.dgamma), ops.ptr(bn.dbeta), ops.ptr(bn.k1), ops.ptr(bn.k2),
                 ops.stream_ptr())
        for a, b_, name in zip(got[j], (bn.dgamma, bn.dbeta, bn.k1, bn.k2), ("dgamma", "dbeta", "k1", "k2")):
            np.testing.assert_allclose(a.cpu().numpy(), b_.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(b_.abs().max() + 1e-6), err_msg=f"job {j} {name}")


@pytest.mark.parametrize("N,H,W,K,C,act", [(2, 16, 16, 512, 512, 2), (1, 13, 11, 256, 256, 2), (8, 96, 96, 512, 256, 1), (3, 16, 16, 1024, 1024, 0),
                                           (9, 85, 83, 256, 256, 2),
                                           # thin layers: the tiled kernel (Conv2d_1..4_pointwise's data gradients, the lateral of c5)
                                           (2, 32, 32, 64, 32, 2), (3, 21, 19, 128, 64, 2), (2, 16, 24, 128, 128, 2), (2, 16, 16, 256, 128, 1),
                                           (1, 8, 8, 128, 1024, 2), (2, 9, 7, 72, 40, 2)],
                         ids=["512to512", "ragged-256", "big-tiles", "1024-none", "big-ragged", "thin-64to32", "thin-ragged-128to64",
                              "thin-128to128", "thin-256to128", "thin-128to1024", "thin-partial-tiles"])
def test_pointwise_dgrad_with_fused_bn_reduction(cuda, N, H, W, K, C, act):
    """mpn_conv_bwd_data_bn, 1x1 through the GEMM kernel (the data gradients of Conv2d_5..13_pointwise feed the depthwise
    batch-norms): dx = the plain data gradient masked by the fed layer's activation, bit for bit; slab sums and the raw finalize
    against f64 references on the kernel's own outputs; 128- and 256-pixel tiles, ragged M, a channel-slice raw tensor."""
    from multiposenet_amd import ops
    dtype = torch.bfloat16
    rs = np.random.RandomState(K + C + H)
    assert ops.conv_bwd_data_bn_supported(K, C, 1, dtype) and not ops.conv_bwd_data_bn_supported(100, 64, 1, dtype)
    assert not ops.conv_bwd_data_bn_supported(K, C, 1, torch.float32)
    w = (rs.randn(1, 1, C, K) / np.sqrt(C)).astype(np.float32)           # forward conv C -> K; its data gradient maps K -> C
    pc = ops.PackedConv(dev(w), dtype)
    dy = dev(rnd(rs.randn(N, H, W, K), dtype), dtype)
    xw = dev(rnd(rs.randn(N, H, W, C + 64) * 1.5 + 0.3, dtype), dtype)
    x = xw[..., 64:64 + C]                                                # a channel slice (pixel stride C + 64)
    bn = ops.BNState(dev(torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)), dev(torch.tensor(rs.randn(C) * 0.3, dtype=torch.float32)),
                     torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), act)
    xf = x.float().reshape(-1, C)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    bn.mean.copy_(mean); bn.invstd.copy_(1.0 / torch.sqrt(var + 1e-3))
    bn.scale.copy_(bn.gamma * bn.invstd); bn.shift.copy_(bn.beta - mean * bn.scale)
    bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    want = ops.conv_fwd(dy, pc.bwd, C, 1)
    out = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=dtype)
    part = torch.full((ops.conv_num_parts(N, H, W, 1) * 2 * C,), float("nan"), device="cuda")
    rows = ops.conv_bwd_data_bn(dy, pc.bwd, C, 1, bn, x, out, part)
    pre = (x.double() * bn.scale.double() + bn.shift.double()).float()
    ok = torch.ones_like(pre, dtype=torch.bool)
    if act != 0:
        ok = pre > 0
    if act == 2:
        ok = ok & (pre < 6)
    g = torch.where(ok, want, torch.zeros_like(want))
    assert int((out != g).sum()) <= 2
    cnt = N * H * W
    ps = part.view(rows, 2, C).double().sum(0).cpu()
    gd, xd = out.double().reshape(-1, C).cpu(), x.double().reshape(-1, C).cpu()
    np.testing.assert_allclose(ps[0].numpy(), gd.sum(0).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(ps[1].numpy(), (gd * xd).sum(0).numpy(), rtol=1e-4, atol=3e-3)
    # in place through bn_backward (raw finalize + apply) against the separate three passes on the masked gradient
    ref = out.clone()
    sp = torch.empty(ops._lib.lib().mpn_bn_stats_num_parts(cnt) * 2 * C, device="cuda")
    xc = x.contiguous()
    ops.bn_backward(bn, ref, xc, sp)
    want_dg, want_db = bn.dgamma.clone(), bn.dbeta.clone()
    ops.bn_backward(bn, out, xc, part, reduced_parts=rows, raw=True)
    np.testing.assert_allclose(bn.dgamma.cpu().numpy(), want_dg.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(want_dg.abs().max()))
    np.testing.assert_allclose(bn.dbeta.cpu().numpy(), want_db.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(want_db.abs().max()))
    assert float((out.float() - ref.float()).abs().max()) <= 2e-2 * float(ref.float().abs().max())


@pytest.mark.parametrize("nparts,C", [(16384, 32), (4096, 64), (5000, 1024), (4097, 8)], ids=["16384x32", "4096x64", "5000x1024", "4097x8"])
def test_finalizes_over_thousands_of_partial_rows(cuda, nparts, C):
    """From 4096 partial rows on (a 1x1 layer at 256 x 256 leaves 16 384) the finalizes first add groups of 32 rows in place:
    forward affine / moving statistics and backward sums (plain and raw) against f64 numpy on the same slab; twice the same
    result (no atomics)."""
    from multiposenet_amd import ops
    rs = np.random.RandomState(nparts + C)
    count = nparts * 128
    slab = (rs.rand(nparts, 2, C) * 100 + 20).astype(np.float32)
    slab[:, 1] = slab[:, 0] ** 2 / 128 + rs.rand(nparts, C).astype(np.float32) * 50          # sum x^2 >= (sum x)^2 / n
    s64 = slab.astype(np.float64).sum(0)
    mean = s64[0] / count
    var = np.maximum(s64[1] / count - mean * mean, 0)

    def run_fwd():
        bn = ops.BNState(torch.ones(C, device="cuda") * 1.5, torch.ones(C, device="cuda") * 0.25, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), 1)
        ops.bn_finalize(bn, torch.tensor(slab).cuda().view(-1), nparts, count, training=True)
        return bn
    bn = run_fwd()
    np.testing.assert_allclose(bn.mean.cpu().numpy(), mean, rtol=1e-5)
    np.testing.assert_allclose(bn.invstd.cpu().numpy(), 1 / np.sqrt(var + 1e-3), rtol=2e-4)
    np.testing.assert_allclose(bn.scale.cpu().numpy(), 1.5 / np.sqrt(var + 1e-3), rtol=2e-4)
    b2 = run_fwd()
    assert torch.equal(bn.scale, b2.scale) and torch.equal(bn.moving_var, b2.moving_var)
    # backward: plain (second row = sum g * xhat) and raw (second row = sum g * x with saved mean / invstd)
    g = (rs.randn(nparts, 2, C) * 3).astype(np.float32)
    g64 = g.astype(np.float64).sum(0)
    bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.call("mpn_bn_bwd_finalize", ops.ptr(torch.tensor(g).cuda()), nparts, C, count, ops.ptr(bn.dgamma), ops.ptr(bn.dbeta), ops.ptr(bn.k1), ops.ptr(bn.k2),
             ops.stream_ptr())
    tol = 1e-5 * float(np.abs(g).sum(0).max())
    np.testing.assert_allclose(bn.dbeta.cpu().numpy(), g64[0], atol=tol)
    np.testing.assert_allclose(bn.dgamma.cpu().numpy(), g64[1], atol=tol)
    np.testing.assert_allclose(bn.k2.cpu().numpy(), g64[1] / count, atol=tol / count)
    m_, i_ = bn.mean.double().cpu().numpy(), bn.invstd.double().cpu().numpy()
    ops.call("mpn_bn_bwd_finalize_raw", ops.ptr(torch.tensor(g).cuda()), nparts, C, count, ops.ptr(bn.dgamma), ops.ptr(bn.dbeta), ops.ptr(bn.k1), ops.ptr(bn.k2),
             ops.ptr(bn.mean), ops.ptr(bn.invstd), ops.stream_ptr())
    want = (g64[1] - m_ * g64[0]) * i_
    np.testing.assert_allclose(bn.dgamma.cpu().numpy(), want, atol=1e-5 * float(np.abs(want).max()) + tol * float(np.abs(m_ * i_).max() + i_.max()))


def test_batched_and_per_layer_finalize_from_4096_rows(cuda):
    """from 4096 partial rows on the per-layer finalizes compact the slab in place first (group sums rounded to
    f32, the slab destroyed), the batched ones do not - include/mpn.h says so now. Here: exactly 4096 rows (the subnet's level-2
    3x3 layers at batch 32 @ 512x512), both paths on copies of one slab: equal to f32 rounding of the group sums, the per-layer
    call leaves its slab changed (compacted), the batched call leaves it untouched."""
    ops = _ops()
    rs = np.random.RandomState(77)
    C, nparts = 128, 4096
    count = nparts * 256
    slab = (rs.rand(nparts, 2, C) * 50 + 5).astype(np.float32)
    slab[:, 1] = slab[:, 0] ** 2 / 256 + rs.rand(nparts, C).astype(np.float32) * 20

    def mk():
        bn = ops.BNState(torch.full((C,), 1.25, device="cuda"), torch.full((C,), -0.5, device="cuda"), torch.zeros(C, device="cuda"),
                         torch.ones(C, device="cuda"), 1)
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn
    a, b = mk(), mk()
    pa, pb = torch.tensor(slab).cuda().view(-1), torch.tensor(slab).cuda().view(-1)
    ops.bn_finalize(a, pa, nparts, count, training=True)
    ops.BnFinalizeBatch([(b, pb, nparts, count)], "cuda:0").run()
    torch.cuda.synchronize()
    assert torch.equal(pb.cpu(), torch.tensor(slab).view(-1))           # batched: the slab is read only
    assert not torch.equal(pa.cpu(), torch.tensor(slab).view(-1))       # per layer: compacted in place (scratch)
    for f in ("scale", "shift", "mean", "invstd", "moving_mean", "moving_var"):
        np.testing.assert_allclose(getattr(a, f).cpu().numpy(), getattr(b, f).cpu().numpy(), rtol=3e-6, atol=1e-7, err_msg=f)
    s64 = slab.astype(np.float64).sum(0)
    np.testing.assert_allclose(b.mean.cpu().numpy(), s64[0] / count, rtol=1e-6)


def test_axpy_batched_equals_per_tensor_axpy(cuda):
    """mpn_axpy_batched (the weight-decay gradient of every regularised variable in one launch, keypoints_model.py:129-138)
    on 70 tensors of ragged sizes (two launches of <= 64 jobs): bit for bit the per-tensor mpn_axpy, nothing outside."""
    ops = _ops()
    rs = np.random.RandomState(3)
    sizes = [int(s) for s in rs.randint(1, 20000, size=68)] + [4096, 1]
    xs = [dev(rs.randn(n).astype(np.float32)) for n in sizes]
    ys = [dev(rs.randn(n + 8).astype(np.float32)) for n in sizes]          # 8 guard elements behind each
    want = [y.clone() for y in ys]
    for x, w in zip(xs, want):
        ops.axpy(5e-5, x, w[:x.numel()])
    ops.AxpyBatch(xs, [y[:x.numel()] for x, y in zip(xs, ys)]).run(5e-5)
    for y, w in zip(ys, want):
        assert torch.equal(y, w)
    assert not torch.equal(ys[0][:sizes[0]], dev(np.zeros(sizes[0], np.float32)))


@pytest.mark.parametrize("M,C,act", [(128 * 5, 64, 1), (1000, 64, 2), (128 * 300 + 77, 64, 1), (900, 32, 1), (333, 16, 0)])
def test_heatmap_head_backward_with_fused_bn_reduction(cuda, M, C, act):
    """mpn_heatmap_head_bwd_bn (bf16, matrix cores): the same dW / db slab as mpn_heatmap_head_bwd; dA = its dA masked by
    the activation of the batch-norm the head reads through, bit for bit; the second slab's sums = sum g and sum g * x over
    all pixels (f64 on the kernel's rounded outputs); the raw finalize gives the dgamma / dbeta of the separate passes."""
    ops = _ops()
    dtype = torch.bfloat16
    assert ops.heatmap_head_bwd_bn_supported(C, dtype) and not ops.heatmap_head_bwd_bn_supported(48, dtype)
    assert not ops.heatmap_head_bwd_bn_supported(C, torch.float32)
    rs = np.random.RandomState(M % 1000 + C)
    x = dev(rnd(rs.randn(1, 1, M, C) * 1.5 + 0.3, dtype), dtype)
    wnp = dev((rs.randn(1, 1, C, 18) * 0.1).astype(np.float32))
    dl = dev(rs.randn(1, 1, M, 18).astype(np.float32))
    bn = ops.BNState(dev(torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)), dev(torch.tensor(rs.randn(C) * 0.3, dtype=torch.float32)),
                     torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), act)
    xf = x.float().reshape(-1, C)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    bn.mean.copy_(mean); bn.invstd.copy_(1.0 / torch.sqrt(var + 1e-3))
    bn.scale.copy_(bn.gamma * bn.invstd); bn.shift.copy_(bn.beta - mean * bn.scale)
    bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    want_dA = torch.empty((1, 1, M, C), dtype=dtype, device="cuda")
    want_dw = torch.full((C * 18 + 18,), float("nan"), device="cuda")
    ops.heatmap_head_bwd(x, dl, wnp, bn.affine, want_dA, want_dw)
    got_dA = torch.full((1, 1, M, C), float("nan"), dtype=dtype, device="cuda")
    got_dw = torch.full((C * 18 + 18,), float("nan"), device="cuda")
    rows_max = ops._lib.lib().mpn_heatmap_head_bwd_num_parts(M)
    part = torch.full((rows_max * 2 * C,), float("nan"), device="cuda")
    rows = ops.heatmap_head_bwd(x, dl, wnp, bn.affine, got_dA, got_dw, bn_part=part)
    assert rows == rows_max and torch.equal(got_dw, want_dw)
    pre = (x.double() * bn.scale.double() + bn.shift.double()).float()
    ok = torch.ones_like(pre, dtype=torch.bool)
    if act != 0:
        ok = pre > 0
    if act == 2:
        ok = ok & (pre < 6)
    assert torch.equal(got_dA, torch.where(ok, want_dA, torch.zeros_like(want_dA)))
    ps = part.view(rows, 2, C).double().sum(0).cpu()
    gd, xd = got_dA.double().reshape(-1, C).cpu(), x.double().reshape(-1, C).cpu()
    np.testing.assert_allclose(ps[0].numpy(), gd.sum(0).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(ps[1].numpy(), (gd * xd).sum(0).numpy(), rtol=1e-4, atol=3e-3)
    ref = got_dA.clone()
    sp = torch.empty(ops._lib.lib().mpn_bn_stats_num_parts(M) * 2 * C, device="cuda")
    ops.bn_backward(bn, ref, x, sp)
    want_dg, want_db = bn.dgamma.clone(), bn.dbeta.clone()
    ops.bn_backward(bn, got_dA, x, part, reduced_parts=rows, raw=True)
    np.testing.assert_allclose(bn.dgamma.cpu().numpy(), want_dg.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(want_dg.abs().max()))
    np.testing.assert_allclose(bn.dbeta.cpu().numpy(), want_db.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(want_db.abs().max()))
    assert float((got_dA.float() - ref.float()).abs().max()) <= 2e-2 * float(ref.float().abs().max())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_adam_step_with_operand_casts(cuda, dtype):
    """mpn_adam_step_cast: the same theta / m / v as mpn_adam_step bit for bit, and the named ranges of the UPDATED arena as
    16-bit copies (what a separate cast pass over the f32 master would give), nothing written outside them."""
    ops = _ops()
    rs = np.random.RandomState(9)
    n = 4 * 50_021
    mk = lambda s: dev((rs.randn(n) * s).astype(np.float32))
    p0, g, m0 = mk(1.0), mk(0.01), mk(0.01)
    v0 = dev((rs.rand(n) * 1e-4).astype(np.float32))
    hyper = dev(np.array([1e-3, 1e-3, 0, 0], np.float32))
    p1, m1, v1 = p0.clone(), m0.clone(), v0.clone()
    ops.adam_step(p1, g, m1, v1, hyper, grad_scale=0.5, clip=float("inf"))
    ranges = [(0, 1024), (4096, 4 * 30_001), (n - 8, 8)]
    dsts = [torch.full((c + 8,), 7.0, dtype=dtype, device="cuda") for _, c in ranges]      # 8 guard elements behind each
    jobs = ops.AdamCastJobs([(o, c, d[:c]) for (o, c), d in zip(ranges, dsts)])
    p2, m2, v2 = p0.clone(), m0.clone(), v0.clone()
    ops.adam_step_cast(p2, g, m2, v2, hyper, jobs, grad_scale=0.5, clip=float("inf"))
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2) and not torch.equal(p0, p2)
    for (o, c), d in zip(ranges, dsts):
        assert torch.equal(d[:c], p2[o:o + c].to(dtype))
        assert bool((d[c:] == 7.0).all())


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 64), (1, 32, 32, 512), (1, 37, 29, 32), (2, 64, 64, 128), (1, 130, 70, 32),
                                       (1, 9, 11, 1024), (3, 24, 24, 256)])
def test_dwconv_bwd_fused_equals_the_separate_launches(cuda, dtype, N, H, W, C):
    """mpn_dwconv_bwd_fused (stride 1): ONE walk gives the data gradient of mpn_dwconv_bwd_data_bn bit for bit, the weight gradient
    of mpn_dwconv_bwd_weight (another summation order: f32 rounding of the partial sums), and partial rows that finalize to the
    dgamma / dbeta of the separate reduction - odd sizes (strips and column pairs that end inside the map), one to many
    channel blocks."""
    ops = _ops()
    rs = np.random.RandomState(C + H + W)
    assert ops.dwconv_bwd_fused_supported(N, H, W, C, 1, dtype)
    x = dev(rnd(rs.randn(N, H, W, C), dtype), dtype)                 # raw input of the depthwise conv = raw output of the fed layer
    dy = dev(rnd(rs.randn(N, H, W, C), dtype), dtype)
    w = dev((rs.randn(3, 3, C) / 3).astype(np.float32))

    def mkbn():
        one = lambda: torch.tensor((0.5 + rs.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), 2)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((rs.randn(C) * 0.5).astype(np.float32)).cuda()); bn.mean.copy_(torch.tensor((rs.randn(C) * 0.3).astype(np.float32)).cuda())
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn
    st = rs.get_state()
    bn_a = mkbn(); rs.set_state(st); bn_b = mkbn()
    # the separate launches
    dw_want = torch.zeros(3, 3, C, device="cuda")
    ops.dwconv_bwd_weight(x, dy, 1, bn_b.affine, dw_want)
    dA_want, rows_b = ops.dwconv_bwd_data(dy, w, (H, W), 1, bn=bn_b, x_bn=x)
    part_b = torch.empty(rows_b * 2 * C, device="cuda")
    dA_b, _ = ops.dwconv_bwd_data(dy, w, (H, W), 1, bn=bn_b, x_bn=x, part=part_b)
    # the fused launch
    dw_got = torch.zeros(3, 3, C, device="cuda")
    dA_got, rows = ops.dwconv_bwd_fused(x, dy, w, bn_a, dw_got)
    assert rows == ops.dwconv_wgrad_num_parts(N, H, W, C, 1, dtype)
    assert torch.equal(dA_got, dA_want)
    scale = float(dw_want.abs().max()) + 1e-6
    assert float((dw_got - dw_want).abs().max()) <= 3e-5 * scale * max(1.0, (N * H * W) ** 0.5 / 16)
    part_a = torch.empty(rows * 2 * C, device="cuda")
    dA_a, _ = ops.dwconv_bwd_fused(x, dy, w, bn_a, dw_got, bn_part=part_a)
    ops.bn_backward(bn_a, dA_a, x, part_a, reduced_parts=rows)
    ops.bn_backward(bn_b, dA_b, x, part_b, reduced_parts=rows_b)
    M = N * H * W
    assert float((bn_a.dgamma - bn_b.dgamma).abs().max()) <= 2e-5 * (float(bn_b.dgamma.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 16)
    assert float((bn_a.dbeta - bn_b.dbeta).abs().max()) <= 2e-5 * (float(bn_b.dbeta.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 16)
    assert_close(dA_a, dA_b.float().cpu(), dtype, 4)       # (after the apply pass: k1 / k2 come from differently grouped partial sums)
    # without the reduction: the same gradients
    dw_n = torch.zeros(3, 3, C, device="cuda")
    dA_n, r0 = ops.dwconv_bwd_fused(x, dy, w, bn_a, dw_n, reduce_bn=False)
    assert r0 == 0 and torch.equal(dA_n, dA_want) and torch.equal(dw_n, dw_got)


@pytest.mark.parametrize("dtype", [torch.bfloat16], ids=["bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,act", [(2, 16, 16, 32, 64, 2), (1, 37, 29, 32, 64, 1), (2, 24, 24, 64, 128, 2), (1, 10, 6, 16, 32, 2),
                                                 (3, 40, 40, 64, 128, 1), (1, 9, 7, 48, 96, 2), (2, 24, 24, 128, 128, 2), (1, 37, 29, 96, 128, 1),
                                                 (1, 20, 20, 128, 64, 2)])
def test_conv1x1_bwd_fused_equals_the_two_pass_backward(cuda, dtype, N, H, W, Cin, Cout, act):
    """mpn_conv1x1_bwd_fused: ONE pass over x and dy gives the weight-gradient slab of mpn_conv_bwd_weight (bit for bit: the same
    tiles, the same MFMA order), the masked data gradient of mpn_conv_bwd_data_bn (same products, the accumulation order of another
    kernel: within a storage ulp) and partial rows that finalize like that kernel's - ragged pixel counts, channels below the tile."""
    ops = _ops()
    rs = np.random.RandomState(Cin + Cout + H)
    x = dev(rnd(rs.randn(N, H, W, Cin), dtype), dtype)
    dy = dev(rnd(rs.randn(N, H, W, Cout), dtype), dtype)
    w = dev((rs.randn(1, 1, Cin, Cout) / np.sqrt(Cout)).astype(np.float32))
    assert ops.conv1x1_bwd_fused_supported(Cin, Cout, dtype) and not ops.conv1x1_bwd_fused_supported(128, 256, dtype)
    assert not ops.conv1x1_bwd_fused_supported(Cin, Cout, torch.float32) and not ops.conv1x1_bwd_fused_supported(Cin, Cout, torch.float16)

    def mkbn(seed):
        r2 = np.random.RandomState(seed)
        one = lambda: torch.tensor((0.5 + r2.rand(Cin)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), act)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((r2.randn(Cin) * 0.5).astype(np.float32)).cuda()); bn.mean.copy_(torch.tensor((r2.randn(Cin) * 0.3).astype(np.float32)).cuda())
        bn.dgamma, bn.dbeta = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        return bn
    bn_a, bn_b = mkbn(5), mkbn(5)
    rows = ops.conv_wgrad_num_parts(N, H, W, Cin, Cout, 1, dtype)
    M = N * H * W
    # two passes
    wp_b = torch.zeros(rows * Cin * Cout, device="cuda")
    dW_b = torch.zeros(1, 1, Cin, Cout, device="cuda")
    ops.conv_bwd_weight(x, dy, 1, bn_b.affine, dW_b, wp_b, reduce=True)
    pc = ops.PackedConv(w, dtype)
    g_b = torch.empty_like(x)
    if ops.conv_bwd_data_bn_supported(Cout, Cin, 1, dtype):
        sp_b = torch.zeros(max(ops.conv_num_parts(N, H, W, 1), rows) * 2 * Cin, device="cuda")
        rows_b = ops.conv_bwd_data_bn(dy, pc.bwd, Cin, 1, bn_b, x, g_b, sp_b)
        ops.bn_backward(bn_b, g_b, x, sp_b, reduced_parts=rows_b, raw=True)
    else:   # (shapes the two-pass fused reduction does not take: plain data gradient + separate reduction)
        ops.conv_fwd(dy, pc.bwd, Cin, 1, None, out=g_b)
        ops.bn_backward(bn_b, g_b, x, torch.zeros(ops._lib.lib().mpn_bn_stats_num_parts(M) * 2 * Cin, device="cuda"))
    # one pass
    wp_a = torch.zeros(rows * Cin * Cout, device="cuda")
    sp_a = torch.zeros(rows * 2 * Cin, device="cuda")
    g_a = torch.empty_like(x)
    r = ops.conv1x1_bwd_fused(x, dy, w, bn_a, g_a, wp_a, sp_a)
    assert r == rows
    dW_a = torch.zeros(1, 1, Cin, Cout, device="cuda")
    ops.call("mpn_reduce_partials", ops.ptr(wp_a), rows, Cin * Cout, ops.ptr(dW_a), 0, 1.0, ops.stream_ptr())
    scale = float(dW_b.abs().max()) + 1e-6
    assert float((dW_a - dW_b).abs().max()) <= 2e-6 * scale * max(1.0, M ** 0.5 / 8)
    ops.bn_backward(bn_a, g_a, x, sp_a, reduced_parts=rows, raw=True)
    assert float((bn_a.dgamma - bn_b.dgamma).abs().max()) <= 3e-5 * (float(bn_b.dgamma.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 16)
    assert float((bn_a.dbeta - bn_b.dbeta).abs().max()) <= 3e-5 * (float(bn_b.dbeta.abs().max()) + 1e-6) * max(1.0, M ** 0.5 / 16)
    assert_close(g_a, g_b.float().cpu(), dtype, 4)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 32, 64), (1, 37, 29, 32, 64), (2, 10, 6, 16, 32), (1, 64, 64, 24, 48), (2, 24, 24, 64, 128),
                                           (1, 19, 23, 48, 96)])
def test_conv1x1_bwd_fused_with_the_apply_pass_folded_in(cuda, N, H, W, Cin, Cout):
    """mpn_conv1x1_bwd_fused_apply: the layer's own batch-norm backward apply happens while dY is staged - the weight slab, the data
    gradient and the reduction partials of mpn_bn_bwd_apply followed by mpn_conv1x1_bwd_fused (to the rounding of a rare staged
    element); g and the raw output stay."""
    ops = _ops()
    dtype = torch.bfloat16
    rs = np.random.RandomState(Cin + Cout + H + 3)
    x = dev(rnd(rs.randn(N, H, W, Cin), dtype), dtype)
    g = dev(rnd(rs.randn(N, H, W, Cout), dtype), dtype)
    yraw = dev(rnd(rs.randn(N, H, W, Cout), dtype), dtype)
    w = dev((rs.randn(1, 1, Cin, Cout) / np.sqrt(Cout)).astype(np.float32))
    assert ops.conv1x1_bwd_fused_apply_supported(Cin, Cout, dtype) and not ops.conv1x1_bwd_fused_apply_supported(128, 128, dtype) and ops.conv1x1_bwd_fused_apply_supported(64, 128, dtype)

    def mkbn(C, seed, act):
        r2 = np.random.RandomState(seed)
        one = lambda: torch.tensor((0.5 + r2.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), act)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((r2.randn(C) * 0.5).astype(np.float32)).cuda()); bn.mean.copy_(torch.tensor((r2.randn(C) * 0.3).astype(np.float32)).cuda())
        bn.k1.copy_(torch.tensor((r2.randn(C) * 0.05).astype(np.float32)).cuda()); bn.k2.copy_(torch.tensor((r2.randn(C) * 0.05).astype(np.float32)).cuda())
        return bn
    below, own = mkbn(Cin, 1, 2), mkbn(Cout, 2, 2)
    rows = ops.conv_wgrad_num_parts(N, H, W, Cin, Cout, 1, dtype)
    dy = g.clone()
    ops.call("mpn_bn_bwd_apply", ops.ptr(dy), ops.ptr(yraw), N * H * W, Cout, ops._lib.dtype_code(dtype), ops.ptr(own.scale), ops.ptr(own.shift),
             ops.ptr(own.mean), ops.ptr(own.invstd), ops.ptr(own.k1), ops.ptr(own.k2), int(own.act), None, ops.stream_ptr())
    wp_b, sp_b, dx_b = torch.zeros(rows * Cin * Cout, device="cuda"), torch.zeros(rows * 2 * Cin, device="cuda"), torch.empty_like(x)
    ops.conv1x1_bwd_fused(x, dy, w, below, dx_b, wp_b, sp_b)
    g0, y0 = g.clone(), yraw.clone()
    wp_a, sp_a, dx_a = torch.zeros(rows * Cin * Cout, device="cuda"), torch.zeros(rows * 2 * Cin, device="cuda"), torch.empty_like(x)
    ops.conv1x1_bwd_fused(x, g, w, below, dx_a, wp_a, sp_a, apply_bn=own, y_raw=yraw)
    # the staged dY is the apply pass's expression with the per-channel constants folded in another translation unit: an element
    # in a few thousand rounds to the neighbouring storage value (seen: 19 of 4096 slab entries off by 1e-5 of 15), nothing more
    assert_close(dx_a, dx_b.float().cpu(), dtype, Cout)
    assert float((wp_a - wp_b).abs().max()) <= 2e-5 * (float(wp_b.abs().max()) + 1e-6)
    assert float((sp_a - sp_b).abs().max()) <= 2e-5 * (float(sp_b.abs().max()) + 1e-6)
    assert float((wp_a != wp_b).float().mean()) < 0.05
    assert torch.equal(g, g0) and torch.equal(yraw, y0)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 128, 128), (1, 37, 29, 64, 128), (1, 10, 6, 32, 64)])
def test_conv1x1_bwd_fused_without_a_reduction_is_the_plain_backward(cuda, N, H, W, Cin, Cout):
    """bn_part = NULL (an FPN lateral): the weight slab of mpn_conv_bwd_weight bit for bit and the UNMASKED data gradient of mpn_conv_fwd
    over the transposed kernel (another kernel's accumulation order: within a storage ulp)."""
    ops = _ops()
    dtype = torch.bfloat16
    rs = np.random.RandomState(Cin + Cout + H + 7)
    x = dev(rnd(rs.randn(N, H, W, Cin), dtype), dtype)
    dy = dev(rnd(rs.randn(N, H, W, Cout), dtype), dtype)
    w = dev((rs.randn(1, 1, Cin, Cout) / np.sqrt(Cout)).astype(np.float32))
    aff = ops.Affine(dev((0.5 + rs.rand(Cin)).astype(np.float32)), dev((rs.randn(Cin) * 0.5).astype(np.float32)), 2)
    rows = ops.conv_wgrad_num_parts(N, H, W, Cin, Cout, 1, dtype)
    wp_b, wp_a = torch.zeros(rows * Cin * Cout, device="cuda"), torch.zeros(rows * Cin * Cout, device="cuda")
    dW = torch.zeros(1, 1, Cin, Cout, device="cuda")
    ops.conv_bwd_weight(x, dy, 1, aff, dW, wp_b, reduce=False)
    pc = ops.PackedConv(w, dtype)
    dx_b = torch.empty_like(x)
    ops.conv_fwd(dy, pc.bwd, Cin, 1, None, out=dx_b)
    dx_a = torch.empty_like(x)
    assert ops.conv1x1_bwd_fused(x, dy, w, aff, dx_a, wp_a, None) == rows
    assert torch.equal(wp_a, wp_b)
    assert_close(dx_a, dx_b.float().cpu(), dtype, Cout)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("N,H,W,C,add", [(2, 16, 16, 64, True), (1, 32, 32, 512, False), (1, 48, 40, 256, True), (2, 64, 64, 128, False),
                                           (1, 6, 10, 1024, True), (1, 130, 70, 32, False)])
def test_dwconv_bwd_fused_stride2_equals_the_separate_launches(cuda, dtype, N, H, W, C, add):
    """mpn_dwconv_bwd_fused_s2: one walk gives the data gradient (+ addend) of mpn_dwconv_bwd_data[_add] with the fused reduction bit
    for bit (same expressions), the weight gradient of mpn_dwconv_bwd_weight (the same per-thread products in the same order: equal)
    and the same partial rows - with and without the FPN lateral's addend, strips and columns that end inside the map."""
    ops = _ops()
    rs = np.random.RandomState(C + H + W + 11)
    assert ops.dwconv_bwd_fused_supported(N, H, W, C, 2, dtype) and not ops.dwconv_bwd_fused_supported(N, H + 1, W, C, 2, dtype)
    OH, OW = ops.dwconv_out_hw(H, W, 2)
    x = dev(rnd(rs.randn(N, H, W, C), dtype), dtype)
    dy = dev(rnd(rs.randn(N, OH, OW, C), dtype), dtype)
    w = dev((rs.randn(3, 3, C) / 3).astype(np.float32))
    addend = dev(rnd(rs.randn(N, H, W, C), dtype), dtype) if add else None

    def mkbn(seed):
        r2 = np.random.RandomState(seed)
        one = lambda: torch.tensor((0.5 + r2.rand(C)).astype(np.float32)).cuda()
        bn = ops.BNState(one(), one(), one(), one(), 2)
        bn.scale.copy_(one()); bn.invstd.copy_(one())
        bn.shift.copy_(torch.tensor((r2.randn(C) * 0.5).astype(np.float32)).cuda()); bn.mean.copy_(torch.tensor((r2.randn(C) * 0.3).astype(np.float32)).cuda())
        bn.dgamma, bn.dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        return bn
    bn = mkbn(3)
    rows = ops.dwconv_wgrad_num_parts(N, H, W, C, 2, dtype)
    assert rows == ops.dwconv_bwd_data_bn_num_parts(N, H, W, C, 2, dtype)
    # the separate launches
    wp_b = torch.zeros(rows * 9 * C, device="cuda")
    ops.dwconv_bwd_weight(x, dy, 2, bn.affine, None, wp_b, reduce=False)
    sp_b = torch.zeros(rows * 2 * C, device="cuda")
    dA_b, r_b = ops.dwconv_bwd_data(dy, w, (H, W), 2, bn=bn, x_bn=x, part=sp_b, addend=addend)
    # the fused walk
    wp_a, sp_a = torch.zeros(rows * 9 * C, device="cuda"), torch.zeros(rows * 2 * C, device="cuda")
    dA_a, r_a = ops.dwconv_bwd_fused(x, dy, w, bn, None, wpart=wp_a, bn_part=sp_a, reduce=False, stride=2, addend=addend)
    assert r_a == r_b == rows
    assert torch.equal(dA_a, dA_b)
    assert float((wp_a - wp_b).abs().max()) <= 1e-6 * (float(wp_b.abs().max()) + 1e-6)
    assert float((sp_a - sp_b).abs().max()) <= 1e-6 * (float(sp_b.abs().max()) + 1e-6)
    # without the reduction
    wp_n = torch.zeros(rows * 9 * C, device="cuda")
    dA_n, r0 = ops.dwconv_bwd_fused(x, dy, w, bn, None, wpart=wp_n, reduce=False, reduce_bn=False, stride=2, addend=addend)
    assert r0 == 0 and torch.equal(dA_n, dA_b) and torch.equal(wp_n, wp_a)


def test_data_gradient_split_over_channel_tiles_of_one_packed_image(cuda):
    """Round 6: final_conv3x3's data gradient (64 -> 512 channels of the concat gradient, keypoint_subnet.py:37-38) as TWO launches over the
    channel tiles of ONE packed image - tile 0 into the concat's first slice with the reduction for phi_subnet_2/bn2 fused
    (mpn_conv_bwd_data_bn: raw x = the same slice of the forward tensor, pixel stride 512), tiles 1..3 plain into the other slices. The
    packed data-gradient image is [channel tile of 128][...] with a tile's weights one contiguous block, so the two launches take
    offsets into it. (Measured in the step and NOT adopted - 7.52 against 7.50 ms, profiles/r06_final_dgrad_split.txt: the one-chunk tiles
    of the fused-reduction variant cost what the level's share of the grouped reduction saved; the entry points' contract stays tested.) Against the single 64 -> 512 launch: slices 1..3 bit for bit; slice 0 = the same values masked by bn2's ReLU, bit
    for bit; the slab's sums = sum g and sum g * x of that slice."""
    from multiposenet_amd import ops
    dtype = torch.bfloat16
    rs = np.random.RandomState(91)
    N, H, W, K, C = 2, 37, 21, 64, 512
    w = (rs.randn(3, 3, C, K) / np.sqrt(9 * C)).astype(np.float32)        # the forward convolution 512 -> 64
    pc = ops.PackedConv(dev(w), dtype)
    dy = dev(rnd(rs.randn(N, H, W, K), dtype), dtype)
    want = ops.conv_fwd(dy, pc.bwd, C, 3)                                  # one launch, four channel tiles
    xcat = dev(rnd(rs.randn(N, H, W, C) * 1.5 + 0.3, dtype), dtype)       # the forward concat tensor: slice 0 = the raw y2 of level 2
    x0 = xcat[..., :128]
    bn = ops.BNState(dev(torch.tensor(0.5 + rs.rand(128), dtype=torch.float32)), dev(torch.tensor(rs.randn(128) * 0.3, dtype=torch.float32)),
                     torch.zeros(128, device="cuda"), torch.ones(128, device="cuda"), 1)
    xf = x0.float().reshape(-1, 128)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    bn.mean.copy_(mean); bn.invstd.copy_(1.0 / torch.sqrt(var + 1e-3))
    bn.scale.copy_(bn.gamma * bn.invstd); bn.shift.copy_(bn.beta - mean * bn.scale)
    got = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=dtype)
    part = torch.full((ops.conv_num_parts(N, H, W, 3) * 2 * 128,), float("nan"), device="cuda")
    tile_bytes = pc.bwd.numel() // 4
    assert pc.bwd.numel() % 4 == 0 and tile_bytes % 16 == 0
    rows = ops.conv_bwd_data_bn(dy, pc.bwd[:tile_bytes], 128, 3, bn, x0, got[..., :128], part)
    ops.conv_fwd(dy, pc.bwd[tile_bytes:], 384, 3, None, out=got[..., 128:])
    assert torch.equal(got[..., 128:], want[..., 128:])
    pre = (x0.double() * bn.scale.double() + bn.shift.double()).float()
    g0 = torch.where(pre > 0, want[..., :128], torch.zeros_like(want[..., :128]))
    assert int((got[..., :128] != g0).sum()) <= 2
    s = part[:rows * 2 * 128].view(rows, 2, 128).double().sum(0).cpu()
    gd, xd = got[..., :128].double().reshape(-1, 128).cpu(), x0.double().reshape(-1, 128).cpu()
    np.testing.assert_allclose(s[0].numpy(), gd.sum(0).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[1].numpy(), (gd * xd).sum(0).numpy(), rtol=1e-4, atol=2e-3)


def _batch_for_more_tiles_than_blocks():
    """Batch size at which the three maps 128^2 + 64^2 + 32^2 (84 tiles of 16 x 16 pixels per image and channel tile) give at least 1.3 x
    as many tiles as the device has compute units (= persistent blocks): 4 on an MI355X (336 tiles on 256 blocks), more on a larger part -
    so that blocks DO walk from one job into the next whatever the block count is."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = max(2, -(-(cus * 13 // 10) // 84))
    assert n * 84 > cus
    return n


@pytest.mark.parametrize("K,C", [(64, 64), (128, 64), (128, 128)], ids=["64->64", "128->64", "128->128"])
def test_grouped_fused_reduction_when_a_block_walks_from_one_job_into_the_next(cuda, K, C):
    """Round 5: with more tiles than blocks a persistent block finishes its last tile of job j and starts on
    job j + 1. Round 4's kernel loaded job j + 1's batch-norm table while that last tile was still being multiplied and masked its
    epilogue with the WRONG layer's scale / shift (the four pyramid levels of a subnet stage have four different batch-norms) - unseen by
    the ragged-tile tests, whose groups are smaller than the grid. Here 1.3 x as many tiles as blocks (336 on the 256 of an MI355X), batch-norms of opposite signs per job:
    the grouped launch equals each job launched alone, bit for bit (64-channel tiles deeper than one chunk: conv3x3.hip; 128-channel tiles
    and one-chunk 64-channel tiles - the detector's towers: conv3x3_cs.hip)."""
    from multiposenet_amd import ops
    dtype = torch.bfloat16
    rs = np.random.RandomState(77 + K + C)
    N = _batch_for_more_tiles_than_blocks()
    sizes = [(128, 128), (64, 64), (32, 32)]
    pc = ops.PackedConv(dev((rs.randn(3, 3, C, K) / np.sqrt(9 * C)).astype(np.float32)), dtype)
    dys = [dev(rnd(rs.randn(N, h, w, K), dtype), dtype) for h, w in sizes]
    xs = [dev(rnd(rs.randn(N, h, w, C) * 1.5, dtype), dtype) for h, w in sizes]
    bns = []
    for j, _ in enumerate(sizes):
        bn = ops.BNState(dev(torch.tensor(0.5 + rs.rand(C), dtype=torch.float32)), dev(torch.tensor(rs.randn(C), dtype=torch.float32)),
                         torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), 1)
        bn.scale.copy_(bn.gamma * (1.0 if j % 2 == 0 else -1.0)); bn.shift.copy_(bn.beta)
        bns.append(bn)
    outs = [torch.empty(N, h, w, C, device="cuda", dtype=dtype) for h, w in sizes]
    parts = [torch.zeros(ops.conv_num_parts(N, h, w, 3) * 2 * C, device="cuda") for h, w in sizes]
    rows = ops.conv_bwd_data_bn_grouped(dys, [pc.bwd] * 3, C, bns, xs, outs, parts)
    for j in range(3):
        o1, p1 = [torch.empty_like(outs[j])], [torch.zeros_like(parts[j])]
        r1 = ops.conv_bwd_data_bn_grouped([dys[j]], [pc.bwd], C, [bns[j]], [xs[j]], o1, p1)
        assert torch.equal(o1[0], outs[j]), (j, int((o1[0] != outs[j]).sum()))
        sg = parts[j][:rows[j] * 2 * C].view(rows[j], 2, C).double().sum(0).cpu().numpy()
        s1 = p1[0][:r1[0] * 2 * C].view(r1[0], 2, C).double().sum(0).cpu().numpy()
        np.testing.assert_allclose(sg, s1, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("Cin,Cout", [(128, 128), (128, 64), (64, 64)], ids=["128->128", "128->64", "64->64"])
def test_grouped_forward_when_a_block_walks_from_one_job_into_the_next(cuda, Cin, Cout):
    """The same walk for the forward launches: each job has its own producer affine (the table a block stages its halo with changes with
    the job) and its own statistics slab row per block. 1.3 x as many tiles as blocks; outputs bit for bit those of the jobs launched alone, the
    slabs' totals to f32 rounding of the blocks' sums."""
    from multiposenet_amd import ops
    dtype = torch.bfloat16
    rs = np.random.RandomState(5 + Cin + Cout)
    N = _batch_for_more_tiles_than_blocks()
    sizes = [(128, 128), (64, 64), (32, 32)]
    xs = [dev(rnd(rs.randn(N, h, w, Cin), dtype), dtype) for h, w in sizes]
    pcs = [ops.PackedConv(dev((rs.randn(3, 3, Cin, Cout) / np.sqrt(9 * Cin)).astype(np.float32)), dtype) for _ in sizes]
    affs = [ops.Affine(dev(torch.tensor((0.5 + rs.rand(Cin)) * (1.0 if j % 2 == 0 else -1.0), dtype=torch.float32)),
                       dev(torch.tensor(rs.randn(Cin) * 0.5, dtype=torch.float32)), 1) for j in range(3)]
    outs = [torch.empty(N, h, w, Cout, device="cuda", dtype=dtype) for h, w in sizes]
    parts = [torch.zeros(ops.conv_num_parts(N, h, w, 3), 2, Cout, device="cuda") for h, w in sizes]
    ops.conv_fwd_grouped(xs, [pc.fwd for pc in pcs], Cout, 3, affs, outs, parts)
    for j, (h, w) in enumerate(sizes):
        p1 = torch.zeros_like(parts[j])
        want = ops.conv_fwd(xs[j], pcs[j].fwd, Cout, 3, affs[j], stats_part=p1)
        assert torch.equal(want, outs[j]), (j, int((want != outs[j]).sum()))
        rows = ops.conv_stats_rows(N, h, w, Cin, Cout, 3, dtype)
        np.testing.assert_allclose(parts[j][:rows].double().sum(0).cpu().numpy(), p1[:rows].double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-2)
