"""BASELINE config 2 at its FULL size (batch 32 @ 512x512, bf16) through size-independent properties: the oracle cannot
run this size in seconds, but these hold for the reference's graph at any size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S = 32, 512


def _net(seed=0):
    from multiposenet_amd.net import KeypointNet
    return KeypointNet(dtype=torch.bfloat16, seed=seed)


def test_inference_is_batch_independent_at_full_size(cuda):
    """is_training=False: batch-norm uses the moving statistics, so every image's logits depend on that image alone -
    image i of the batch of 32 must equal the same image run alone, bit for bit (mobilenet_v1.py:29-38 with
    is_training False; any cross-image leak in a tile, a halo or a statistics row shows up here)."""
    net = _net()
    rs = np.random.RandomState(3)
    img = torch.tensor(rs.rand(B, S, S, 3).astype(np.float32)).cuda()
    logits, feats = net.forward(img, False)
    logits = logits.clone()
    p2 = feats["p2"].clone()
    assert torch.isfinite(logits).all()
    for i in (0, 13, 31):
        l1, f1 = net.forward(img[i:i + 1].contiguous(), False)
        assert torch.equal(l1[0], logits[i]), i
        assert torch.equal(f1["p2"][0], p2[i]), i


def test_training_step_is_deterministic_and_finite_at_full_size(cuda):
    """Two identically seeded replicas make the same step, bit for bit (no atomics in any reduction), every loss term and
    gradient is finite, and the batch statistics the step stores are those of the batch (stem layer, checked with torch)."""
    from multiposenet_amd.train import Trainer
    from multiposenet_amd.synthetic import synthetic_batch
    hp = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    out = []
    for rep in range(2):
        net = _net(seed=1)
        tr = Trainer(net, hp, use_graph=(rep == 1))          # eager vs hipGraph replay of the same step
        feats, labels = synthetic_batch(B, S, S, rank=0, device="cuda:0")
        losses = [tr.step(feats, labels).cpu().numpy().copy() for _ in range(2)]
        out.append((losses, net.grad.clone(), net.theta.clone(), net.moving.clone()))
    for a, b in zip(out[0][0], out[1][0]):
        assert np.all(np.isfinite(a))
        np.testing.assert_array_equal(a, b)
    for k in (1, 2, 3):
        assert torch.isfinite(out[0][k]).all()
        assert torch.equal(out[0][k], out[1][k])
    # moving statistics after one step from zero-mean / unit-variance initial values: (1 - momentum) * batch statistics
    net = _net(seed=1)
    feats, labels = synthetic_batch(B, S, S, rank=0, device="cuda:0")
    net.forward(feats["images"], True)
    bstem = net._last[0]["stem"].float()                      # raw stem conv output [B,256,256,32]
    mean = bstem.mean(dim=(0, 1, 2))
    mm = net.stats["MobilenetV1/Conv2d_0/BatchNorm/moving_mean"]
    from multiposenet_amd.ops import BN_MOMENTUM              # 0.95: detector/backbones/mobilenet_v1.py:7
    torch.testing.assert_close(mm, (1 - BN_MOMENTUM) * mean, rtol=2e-2, atol=2e-4)   # moving_mean starts at 0


# ---------------------------------------------------------------- BASELINE config 4 at its full size
DB, DH, DW = 16, 896, 1408      # 800 x 1333 padded to multiples of 128 (constants.py:4)
DHP = {"initial_learning_rate": 1e-3, "num_steps": 150000, "weight_decay": 5e-5, "localization_loss_weight": 1.0,
       "classification_loss_weight": 2.0, "gamma": 2.0, "alpha": 0.25}


def _detector_batch(seed=7, maxn=12):
    rs = np.random.RandomState(seed)
    boxes = np.zeros((DB, maxn, 4), np.float32)
    for b in range(DB):
        for n in range(maxn):
            cy, cx = rs.rand(2)
            h, w = 0.08 + 0.5 * rs.rand(2)
            boxes[b, n] = [max(cy - h / 2, 0), max(cx - w / 2, 0), min(cy + h / 2, 1), min(cx + w / 2, 1)]
    num = rs.randint(1, maxn + 1, DB).astype(np.int32)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.rand((DB, DH, DW, 3), generator=g, device="cuda"), boxes, num


def test_detector_step_at_full_size_properties(cuda):
    """RetinaNet head (cfg4: batch 16 @ 896x1408, bf16), where the oracle cannot run in seconds: anchor matching of image i is
    independent of the batch and equals the numpy restatement (training_target_creation.py:5-159) on the same boxes, bit
    for bit; the number of matched anchors equals the restatement's count; an eager step and its hipGraph replay agree bit
    for bit on losses, gradients, variables and moving statistics; everything finite."""
    from multiposenet_amd.retinanet import PersonDetectorNet, generate_anchors
    from oracle import retinanet as R
    images, boxes, num = _detector_batch()
    gt = {"boxes": torch.from_numpy(boxes).cuda(), "num_boxes": torch.from_numpy(num).cuda()}
    anchors, _ = generate_anchors(DH, DW)
    assert anchors.shape[0] == 157542
    ref = PersonDetectorNet(dtype=torch.bfloat16, seed=0)
    want = [ref.train_step(images, gt, DHP).cpu().numpy().copy() for _ in range(2)]        # eager
    bset = ref._last[0]
    matches = bset["matches"].cpu().numpy()
    targets = bset["targets"].cpu().numpy()
    total = 0
    for i in (0, 7, 15):
        wt, wm = R.get_training_targets(anchors, boxes[i, :num[i]])
        np.testing.assert_array_equal(matches[i], wm, err_msg=f"image {i}")
        pos = wm >= 0
        np.testing.assert_allclose(targets[i][pos], wt[pos], rtol=2e-6, atol=2e-6)
    for i in range(DB):
        _, wm = R.get_training_targets(anchors, boxes[i, :num[i]])
        total += int((wm >= 0).sum())
    assert int(bset["num_matched"].item()) == total > 0
    # image i alone (a batch of one, another buffer set): the same matches
    one = ref._buffers(1, DH, DW)
    ref.create_targets({"boxes": gt["boxes"][7:8].contiguous(), "num_boxes": gt["num_boxes"][7:8].contiguous()}, b=one)
    np.testing.assert_array_equal(one["matches"].cpu().numpy()[0], matches[7])
    for w_ in want:
        assert np.all(np.isfinite(w_))
    assert torch.isfinite(ref.theta).all() and torch.isfinite(ref.grad).all() and torch.isfinite(ref.moving).all()
    # the same two steps, the second one replayed from a hipGraph
    net = PersonDetectorNet(dtype=torch.bfloat16, seed=0)
    first = net.train_step(images, gt, DHP).cpu().numpy().copy()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        losses = net.train_step(images, gt, DHP)
    g.replay()
    got = [first, losses.cpu().numpy().copy()]
    for a, b_ in zip(want, got):
        np.testing.assert_array_equal(a, b_)
    assert torch.equal(ref.theta, net.theta) and torch.equal(ref.moving, net.moving) and torch.equal(ref.grad, net.grad)
    # inference at full size: batch-independent detections
    pred = ref.predict(images, 0.05, 0.5, 25)
    p1 = ref.predict(images[5:6].contiguous(), 0.05, 0.5, 25)
    assert int(p1["num_boxes"][0]) == int(pred["num_boxes"][5])
    assert torch.equal(p1["scores"][0], pred["scores"][5]) and torch.equal(p1["boxes"][0], pred["boxes"][5])


def test_prn_step_at_full_size_properties(cuda):
    """PRN (cfg5: 128 crops x 56 x 36 x 17, fp16): an eager step and its hipGraph replay agree bit for bit, every value finite,
    and the loss of a crop batch does not depend on the ORDER of the crops (a mean over crops: a permuted batch gives the
    same per-crop loss terms)."""
    from multiposenet_amd.prn import PoseResidualNet
    rs = np.random.RandomState(3)
    B, h, w, c = 128, 56, 36, 17
    x = torch.tensor(rs.rand(B, h, w, c).astype(np.float32)).cuda()
    y = torch.zeros(B, h, w, c)
    for b in range(B):
        for k in range(c):
            if rs.rand() < 0.8:
                y[b, rs.randint(h), rs.randint(w), k] = 1.0
    y = y.cuda()
    ref = PoseResidualNet(batch=B, dtype=torch.float16, seed=0)
    want = [float(ref.train_step(x, y, 1e-3, 1000)) for _ in range(2)]
    net = PoseResidualNet(batch=B, dtype=torch.float16, seed=0)
    first = float(net.train_step(x, y, 1e-3, 1000))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = net.train_step(x, y, 1e-3, 1000)
    g.replay()
    assert [first, float(loss)] == want and all(np.isfinite(v) for v in want)
    assert torch.equal(ref.theta, net.theta) and torch.isfinite(ref.theta).all() and torch.isfinite(ref.grad).all()
    # permutation of the crops: the same per-crop loss terms
    perm = torch.tensor(rs.permutation(B)).cuda()
    net.forward(x)
    net.loss(y, with_grad=False)
    a = net.loss_part.clone()
    net.forward(x[perm].contiguous())
    net.loss(y[perm].contiguous(), with_grad=False)
    torch.testing.assert_close(net.loss_part, a[perm], rtol=2e-3, atol=1e-9)


# ---------------------------------------------------------------- grouped launches at the REAL job tables
# Round 5's bug (15748a0): a persistent 3x3 block walking from one job of a grouped launch into the next masked its last tile with the
# NEXT job's batch-norm. It passed every whole-net parity test (their sizes give fewer tiles than blocks) and every full-size test
# (properties only). These tests run the grouped entry points at the job tables the two headline steps really launch - where blocks DO
# cross jobs - against the same jobs launched alone: tensors bit for bit, f32 partial sums to rounding (their split differs).
KP_LEVELS = [(128, 128), (64, 64), (32, 32), (16, 16)]                      # cfg2: p2..p5 at batch 32 (keypoint_subnet.py:30-35)
DET_LEVELS = [(112, 176), (56, 88), (28, 44), (14, 22), (7, 11)]            # cfg4: p3..p7 of 896 x 1408 at batch 16 (retinanet.py:169-217)


def _rand_bf16(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


def _bn_state(C, seed, act=1, sign=1.0):
    from multiposenet_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    bn = ops.BNState(0.5 + torch.rand(C, generator=g, device="cuda"), torch.randn(C, generator=g, device="cuda") * 0.3,
                     torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), act)
    bn.scale.copy_(bn.gamma * sign); bn.shift.copy_(bn.beta)
    bn.mean.copy_(torch.randn(C, generator=g, device="cuda") * 0.1); bn.invstd.copy_(0.5 + torch.rand(C, generator=g, device="cuda"))
    bn.k1.copy_(torch.randn(C, generator=g, device="cuda") * 0.05); bn.k2.copy_(torch.randn(C, generator=g, device="cuda") * 0.05)
    return bn


def _slab_total(part, rows, width):
    return part.reshape(-1)[:rows * width].view(rows, width).double().sum(0).cpu().numpy()


@pytest.mark.parametrize("table", ["cfg2-subnet-128to128", "cfg4-towers-64to64", "cfg4-merged-tower-128to128"])
def test_grouped_3x3_launches_equal_their_jobs_alone_at_the_real_job_tables(cuda, table):
    """mpn_conv_fwd_grouped (producer affine + statistics), mpn_conv_bwd_data_bn_grouped (fused reduction), mpn_conv_bwd_weight_grouped,
    mpn_bn_bwd_reduce_grouped and mpn_bn_bwd_apply_grouped over the four pyramid levels of a keypoint-subnet stage at batch 32, and over
    the detector's five-level tower grid at batch 16 (64 -> 64 towers; the merged 128 -> 128 first layer): every job has its own
    batch-norm (alternating signs: a tile masked or staged with a neighbour's table is wrong everywhere), its own slab; outputs bit for
    bit those of the job launched alone, slab totals to f32 rounding."""
    from multiposenet_amd import ops
    dt = torch.bfloat16
    if table.startswith("cfg2"):
        N, sizes, Cin, Cout = B, KP_LEVELS, 128, 128
    elif "64to64" in table:
        N, sizes, Cin, Cout = DB, DET_LEVELS, 64, 64
    else:
        N, sizes, Cin, Cout = DB, DET_LEVELS, 128, 128
    n = len(sizes)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = sum(N * ((h + 15) // 16) * ((w + 15) // 16) for h, w in sizes) * max(1, Cout // 128)
    assert tiles > 2 * cus                                   # blocks walk several tiles and cross jobs
    xs = [_rand_bf16((N, h, w, Cin), 11 + j) for j, (h, w) in enumerate(sizes)]
    pcs = [ops.PackedConv(torch.randn(3, 3, Cin, Cout, device="cuda") / (9 * Cin) ** 0.5, dt) for _ in sizes]
    affs = [ops.Affine((0.5 + torch.rand(Cin, device="cuda")) * (1.0 if j % 2 == 0 else -1.0), torch.randn(Cin, device="cuda") * 0.5, 1) for j in range(n)]
    # ---- forward: affine on load + statistics
    outs = [torch.empty(N, h, w, Cout, device="cuda", dtype=dt) for h, w in sizes]
    parts = [torch.zeros(ops.conv_num_parts(N, h, w, 3), 2, Cout, device="cuda") for h, w in sizes]
    ops.conv_fwd_grouped(xs, [pc.fwd for pc in pcs], Cout, 3, affs, outs, parts)
    for j, (h, w) in enumerate(sizes):
        p1 = torch.zeros_like(parts[j])
        want = ops.conv_fwd(xs[j], pcs[j].fwd, Cout, 3, affs[j], stats_part=p1)
        assert torch.equal(want, outs[j]), ("forward", j, int((want != outs[j]).sum()))
        rows = ops.conv_stats_rows(N, h, w, Cin, Cout, 3, dt)
        np.testing.assert_allclose(_slab_total(parts[j], rows, 2 * Cout), _slab_total(p1, rows, 2 * Cout), rtol=2e-5, atol=5e-2, err_msg=f"forward slab {j}")
    # ---- data gradient with the fused reduction for the batch-norm it feeds (dy: Cout channels -> dx: Cin channels)
    dys = [_rand_bf16((N, h, w, Cout), 31 + j) for j, (h, w) in enumerate(sizes)]
    xbn = [_rand_bf16((N, h, w, Cin), 51 + j, 1.5) for j, (h, w) in enumerate(sizes)]
    bns = [_bn_state(Cin, 71 + j, 1, 1.0 if j % 2 == 0 else -1.0) for j in range(n)]
    dxs = [torch.empty(N, h, w, Cin, device="cuda", dtype=dt) for h, w in sizes]
    bparts = [torch.zeros(ops.conv_num_parts(N, h, w, 3) * 2 * Cin, device="cuda") for h, w in sizes]
    assert ops.conv_bwd_data_bn_supported(Cout, Cin, 3, dt)
    rows = ops.conv_bwd_data_bn_grouped(dys, [pc.bwd for pc in pcs], Cin, bns, xbn, dxs, bparts)
    for j in range(n):
        o1, p1 = [torch.empty_like(dxs[j])], [torch.zeros_like(bparts[j])]
        r1 = ops.conv_bwd_data_bn_grouped([dys[j]], [pcs[j].bwd], Cin, [bns[j]], [xbn[j]], o1, p1)
        assert r1[0] == rows[j] and torch.equal(o1[0], dxs[j]), ("data gradient", j, int((o1[0] != dxs[j]).sum()))
        np.testing.assert_allclose(_slab_total(bparts[j], rows[j], 2 * Cin), _slab_total(p1[0], rows[j], 2 * Cin), rtol=2e-5, atol=5e-2, err_msg=f"reduction slab {j}")
    # ---- weight gradient (grouped split-K grid): slab totals
    hws = list(sizes)
    nps = ops.conv_wgrad_grouped_num_parts(N, hws, Cin, Cout, 3, dt)
    wparts = [torch.zeros(np_ * 9 * Cin * Cout, device="cuda") for np_ in nps]
    ops.conv_bwd_weight_grouped(xs, dys, 3, affs, wparts)
    for j, (h, w) in enumerate(sizes):
        np1 = ops.conv_wgrad_grouped_num_parts(N, [(h, w)], Cin, Cout, 3, dt)[0]
        w1 = [torch.zeros(np1 * 9 * Cin * Cout, device="cuda")]
        ops.conv_bwd_weight_grouped([xs[j]], [dys[j]], 3, [affs[j]], w1)
        a, b_ = _slab_total(wparts[j], nps[j], 9 * Cin * Cout), _slab_total(w1[0], np1, 9 * Cin * Cout)
        assert np.abs(a - b_).max() <= 2e-5 * np.abs(b_).max(), ("weight gradient", j, np.abs(a - b_).max(), np.abs(b_).max())
    # ---- batch-norm backward: grouped reduction and grouped apply
    nb = [ops._lib.lib().mpn_bn_stats_num_parts(N * h * w) for h, w in sizes]
    rparts = [torch.zeros(r * 2 * Cin, device="cuda") for r in nb]
    dA = [d.clone() for d in dxs]
    ops.bn_bwd_reduce_grouped(bns, dA, xbn, rparts)
    for j in range(n):
        p1 = torch.zeros_like(rparts[j])
        r1 = ops.bn_bwd_reduce(bns[j], dxs[j].clone(), xbn[j], p1)
        assert r1 == nb[j]
        np.testing.assert_allclose(_slab_total(rparts[j], nb[j], 2 * Cin), _slab_total(p1, nb[j], 2 * Cin), rtol=2e-5, atol=5e-2, err_msg=f"bn reduce {j}")
    adds = [torch.randn(N, h, w, device="cuda") * 0.2 for h, w in sizes]
    ops.bn_bwd_apply_grouped(bns, dA, xbn, adds)
    for j in range(n):
        want = ops.bn_bwd_apply(bns[j], dxs[j].clone(), xbn[j], adds[j])
        assert torch.equal(want, dA[j]), ("bn apply", j, int((want != dA[j]).sum()))


def _step_outputs(make_net, step, alone):
    """One eager train step of a freshly seeded net with every grouped / batched launch shared (alone=False) or split into its jobs."""
    from multiposenet_amd import ops
    ops.LAUNCH_JOBS_ALONE = alone
    try:
        net = make_net()
        losses = step(net)
        torch.cuda.synchronize()
        out = (losses, {k: v.clone() for k, v in net.grads.items()}, {k: v.clone() for k, v in net.stats.items()})
    finally:
        ops.LAUNCH_JOBS_ALONE = False
    del net
    torch.cuda.empty_cache()
    return out


def _compare_steps(a, b, grad_tol, what):
    """Gradients per tensor in relative L2 (the f32 partial sums of the two runs split differently, and a sum that differs in its last
    bit moves a rare 16-bit element of a later tensor by an ulp: measured at most 6.7e-7 (cfg2) / 4.0e-7 (cfg4); a tile computed with a
    neighbour job's table - round 5's bug - moves a tensor by 1e-2..1), moving statistics to f32 rounding, losses to 1e-5."""
    np.testing.assert_allclose(a[0], b[0], rtol=1e-5, atol=1e-6, err_msg=f"{what}: losses")
    worst = []
    for k, ga in a[1].items():
        gb = b[1][k]
        den = float(gb.double().norm())
        err = float((ga.double() - gb.double()).norm()) / (den + 1e-30)
        if den > 0 or float(ga.double().norm()) > 0:
            worst.append((err, k))
    worst.sort(reverse=True)
    print(f"{what}: grouped vs alone, worst gradient tensors (relative L2):", [(f"{e:.2e}", k) for e, k in worst[:4]], "of", len(worst))
    assert worst and worst[0][0] <= grad_tol, worst[:8]
    for k, ma in a[2].items():
        torch.testing.assert_close(ma, b[2][k], rtol=2e-5, atol=2e-6, msg=lambda m, k=k: f"{what}: {k}: {m}")


def test_keypoint_step_with_shared_grids_equals_the_step_with_every_job_alone(cuda):
    """cfg2 (batch 32 @ 512 x 512, bf16): one TRAIN step with the grouped launches of the subnet / FPN stages, the batched finalizes, the
    batched slab reduction and the batched packer, against the same step with every job launched alone (ops.LAUNCH_JOBS_ALONE)."""
    from multiposenet_amd.train import Trainer
    from multiposenet_amd.synthetic import synthetic_batch
    hp = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    feats, labels = synthetic_batch(B, S, S, rank=0, device="cuda:0")

    def step(net):
        tr = Trainer(net, hp, use_graph=False)
        return tr.step(feats, labels).cpu().numpy().copy()
    shared = _step_outputs(lambda: _net(seed=1), step, False)
    alone = _step_outputs(lambda: _net(seed=1), step, True)
    _compare_steps(shared, alone, 2e-5, "cfg2")


def test_detector_step_with_shared_grids_equals_the_step_with_every_job_alone(cuda):
    """cfg4 (batch 16 @ 896 x 1408, bf16): the five-level tower grids, the merged first tower layer, the grouped weight gradients and
    finalizes, against every job launched alone."""
    from multiposenet_amd.retinanet import PersonDetectorNet
    images, boxes, num = _detector_batch()
    gt = {"boxes": torch.from_numpy(boxes).cuda(), "num_boxes": torch.from_numpy(num).cuda()}

    def step(net):
        return net.train_step(images, gt, DHP).cpu().numpy().copy()
    shared = _step_outputs(lambda: PersonDetectorNet(dtype=torch.bfloat16, seed=0), step, False)
    alone = _step_outputs(lambda: PersonDetectorNet(dtype=torch.bfloat16, seed=0), step, True)
    _compare_steps(shared, alone, 2e-5, "cfg4")
