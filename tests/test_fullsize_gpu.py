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
