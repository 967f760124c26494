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
