"""Variables on disk in the reference's naming (multiposenet_amd/checkpoint.py): save / resume / warm start."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=3):
    from test_network_gpu import _params, _labels
    rs = np.random.RandomState(seed)
    B, H, W = 2, 128, 128
    img = torch.tensor(rs.rand(B, H, W, 3).astype(np.float32)).cuda()
    lab = {k: torch.tensor(v).cuda() for k, v in _labels(rs, B, H // 4, W // 4).items()}
    hp = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    return _params(seed), img, lab, hp


def test_resume_is_bit_identical_and_names_are_tensorflows(cuda, tmp_path):
    from multiposenet_amd.net import KeypointNet
    from multiposenet_amd.train import Trainer
    from multiposenet_amd import checkpoint
    params, img, lab, hp = _setup()
    net = KeypointNet(values=params, dtype=torch.float32)
    tr = Trainer(net, hp, use_graph=False)
    for _ in range(2):
        tr.step({"images": img}, lab)
    path = str(tmp_path / "model.npz")
    names = checkpoint.save_npz(path, net)
    assert "MobilenetV1/Conv2d_0/weights/Adam" in names and "MobilenetV1/Conv2d_0/weights/Adam_1" in names
    assert "MobilenetV1/Conv2d_1_depthwise/BatchNorm/moving_variance" in names
    assert "global_step" in names and "beta1_power" in names
    with np.load(path) as z:
        assert int(z["global_step"]) == 2
        assert abs(float(z["beta1_power"]) - 0.729) < 1e-6 and z["MobilenetV1/Conv2d_1_depthwise/depthwise_weights"].shape[-1] == 1
    third = tr.step({"images": img}, lab).cpu().numpy().copy()
    want = net.state_dict()
    # a fresh process: different initial values, then restore
    net2 = KeypointNet(values=None, dtype=torch.float32, seed=99)
    restored = checkpoint.load_npz(path, net2)
    assert "global_step" in restored and int(net2.global_step.item()) == 2
    tr2 = Trainer(net2, hp, use_graph=False)
    third2 = tr2.step({"images": img}, lab).cpu().numpy()
    np.testing.assert_array_equal(third2, third)
    got = net2.state_dict()
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_resume_with_a_padded_stem_width_keeps_reference_shapes(cuda, tmp_path):
    """depth_multiplier 0.75: the stem's 24 channels are 32 inside the arena (net.internal_shapes); the file holds the
    reference shapes (variables AND Adam slots), and a resumed run continues bit for bit."""
    from multiposenet_amd.net import KeypointNet
    from multiposenet_amd.train import Trainer
    from multiposenet_amd import checkpoint
    from oracle import network as onet
    _, img, lab, hp = _setup()
    hp = dict(hp, depth_multiplier=0.75)
    params = onet.randomize_bn(onet.init_params(5, depth_multiplier=0.75), 6)
    net = KeypointNet(values=params, depth_multiplier=0.75, dtype=torch.bfloat16)
    tr = Trainer(net, hp, use_graph=False)
    for _ in range(2):
        tr.step({"images": img}, lab)
    path = str(tmp_path / "model.npz")
    checkpoint.save_npz(path, net)
    with np.load(path) as z:
        for k in ("MobilenetV1/Conv2d_0/weights", "MobilenetV1/Conv2d_0/weights/Adam", "MobilenetV1/Conv2d_0/weights/Adam_1"):
            assert z[k].shape == (3, 3, 3, 24), (k, z[k].shape)
        assert z["MobilenetV1/Conv2d_1_depthwise/BatchNorm/moving_variance"].shape == (24,)
        assert z["MobilenetV1/Conv2d_1_pointwise/weights/Adam"].shape == (1, 1, 24, 48)
    third = tr.step({"images": img}, lab).cpu().numpy().copy()
    net2 = KeypointNet(values=None, depth_multiplier=0.75, dtype=torch.bfloat16, seed=99)
    checkpoint.load_npz(path, net2)
    third2 = Trainer(net2, hp, use_graph=False).step({"images": img}, lab).cpu().numpy()
    np.testing.assert_array_equal(third2, third)
    a, b = net.state_dict(), net2.state_dict()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_warm_start_restores_the_backbone_only(cuda, tmp_path):
    from multiposenet_amd.net import KeypointNet
    from multiposenet_amd import checkpoint
    params, _, _, _ = _setup(5)
    src = KeypointNet(values=params, dtype=torch.bfloat16)
    # what slim's classification checkpoint holds: MobilenetV1/* plus names the keypoint model does not have
    sd = {k: v for k, v in src.state_dict().items() if k.startswith("MobilenetV1/")}
    sd["MobilenetV1/Logits/Conv2d_1c_1x1/weights"] = np.zeros((1, 1, 1024, 1001), np.float32)
    path = str(tmp_path / "mobilenet_v1.npz")
    np.savez(path, **sd)
    net = KeypointNet(values=None, dtype=torch.bfloat16, seed=42)
    before = net.state_dict()
    restored = checkpoint.warm_start(path, net)
    after = net.state_dict()
    assert restored and all(k.startswith("MobilenetV1/") for k in restored)
    for k in after:
        if k.startswith("MobilenetV1/"):
            np.testing.assert_array_equal(after[k], sd[k], err_msg=k)
        else:
            np.testing.assert_array_equal(after[k], before[k], err_msg=k)
    assert int(net.global_step.item()) == 0
    with pytest.raises(KeyError):
        checkpoint.load_npz(path, net)               # strict restore of the whole model from a backbone-only file


def test_prn_round_trip(cuda, tmp_path):
    from multiposenet_amd.prn import PoseResidualNet
    from multiposenet_amd import checkpoint
    net = PoseResidualNet(batch=8, h=8, w=6, dtype=torch.float32, seed=1)
    x = torch.rand(8, 8, 6, 17, device="cuda")
    y = torch.zeros_like(x); y[:, 2, 3, :] = 1
    net.train_step(x, y, 1e-3, 1000)
    path = str(tmp_path / "prn.npz")
    names = checkpoint.save_npz(path, net)
    assert "PRN/fc1/weights/Adam_1" in names
    want = net.predict(x).cpu().numpy()
    net2 = PoseResidualNet(batch=8, h=8, w=6, dtype=torch.float32, seed=2)
    checkpoint.load_npz(path, net2)
    np.testing.assert_array_equal(net2.predict(x).cpu().numpy(), want)
    a = float(net.train_step(x, y, 1e-3, 1000)); b = float(net2.train_step(x, y, 1e-3, 1000))
    assert a == b


def test_train_keypoints_loop_checkpoints_and_resumes(cuda, tmp_path):
    """The estimator-loop replacement (multiposenet_amd.train_keypoints.train): warm start of MobilenetV1/* from an .npz,
    summaries every N steps, a checkpoint at the end, and a second call that RESUMES from model_dir and continues exactly
    where an uninterrupted run would be (bit-identical variables)."""
    import json
    from multiposenet_amd import keypoints_model as km
    from multiposenet_amd import checkpoint
    from multiposenet_amd import train_keypoints as tk
    from multiposenet_amd.net import KeypointNet
    pre = str(tmp_path / "pretrained.npz")
    src = KeypointNet(dtype=torch.bfloat16, seed=5)
    checkpoint.save_npz(pre, src, with_optimizer=False)
    base = dict(tk.PARAMS, pretrained_checkpoint=pre, batch_size=2, image_size=(128, 128), dtype="bf16", seed=1)
    cfg = {"save_summary_steps": 2, "log_step_count_steps": 3}

    def batches():
        return tk.synthetic_batches(2, 128, 128, distinct=3)
    logs = []
    km.reset_registry()
    a = dict(base, model_dir=str(tmp_path / "a"))
    assert tk.train(a, batches, run_config=cfg, max_steps=5, log=logs.append) == 5
    assert any("warm start" in l for l in logs) and os.path.exists(os.path.join(a["model_dir"], "model.ckpt-5.npz"))
    recs = [json.loads(l) for l in open(os.path.join(a["model_dir"], "summaries.jsonl"))]
    assert [r["step"] for r in recs] == [2, 4] and "focal_loss" in recs[0]
    net_a = km.get_trainer(a).net
    want = {k: v.copy() for k, v in net_a.state_dict().items()}
    # interrupted run: 3 steps, then a fresh process (registry cleared) resumes from model_dir and does steps 4, 5
    km.reset_registry()
    b = dict(base, model_dir=str(tmp_path / "b"))
    tk.train(b, batches, run_config=cfg, max_steps=3, log=logs.append)
    km.reset_registry()

    def batches_from_4th():
        it = batches()
        for _ in range(3):
            next(it)
        return it
    assert tk.train(b, batches_from_4th, run_config=cfg, max_steps=5, log=logs.append) == 5
    assert any("restored" in l for l in logs)
    got = km.get_trainer(b).net.state_dict()
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
