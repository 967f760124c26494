"""Tiny end-to-end invocation of the hot path used by __graft_entry__.smoke(): one forward + losses + backward +
Adam step on cuda:0 (f32 storage), checked against the oracle (imported here as the checker only)."""
import numpy as np
import torch


def run():
    from oracle import network as onet          # checker
    from multiposenet_amd.net import KeypointNet
    from multiposenet_amd.train import Trainer
    rs = np.random.RandomState(0)
    B, H, W = 1, 128, 128
    params = onet.randomize_bn(onet.init_params(0), 1)
    params["heatmaps/kernel"] = (rs.randn(1, 1, 64, 18) * 0.05).astype(np.float32)
    img = rs.rand(B, H, W, 3).astype(np.float32)
    h = H // 4
    hm = (rs.rand(B, h, h, 17) * 0.9).astype(np.float32)
    hm[0, 5, 7, 3] = 1.0
    lab = {"heatmaps": hm, "loss_masks": (rs.rand(B, h, h) < 0.9).astype(np.float32),
           "segmentation_masks": (rs.rand(B, h, h) < 0.3).astype(np.float32), "num_boxes": np.array([2], np.int32)}
    hp = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    ref = {k: v.astype(np.float64) for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in ref.items()}
    v = {k: np.zeros_like(v) for k, v in ref.items()}
    total, losses, _ = onet.train_step(ref, m, v, img, lab, 0, hp, dtype=torch.float64)
    net = KeypointNet(values=params, dtype=torch.float32)
    tr = Trainer(net, hp, use_graph=False)
    out = tr.step({"images": torch.tensor(img).cuda()}, {k: torch.tensor(x).cuda() for k, x in lab.items()}).cpu().numpy()
    np.testing.assert_allclose(out[6], total, rtol=1e-3)
    np.testing.assert_allclose(out[:6], list(losses.values()), rtol=1e-3, atol=1e-9)
    sd = net.state_dict()
    worst = max(float(np.abs(sd[k] - ref[k]).max()) for k in ref if onet.is_trainable(k))
    assert worst <= 2 * 3e-4 + 1e-6, worst      # one Adam step moves a weight by at most ~lr
    print(f"smoke: train step OK (total loss {out[6]:.5f} vs oracle {total:.5f})")
    # bf16 throughput build: one graph-captured step must run and produce a finite loss
    net16 = KeypointNet(values=params, dtype=torch.bfloat16)
    tr16 = Trainer(net16, hp, use_graph=True)
    l16 = float(tr16.step({"images": torch.tensor(img).cuda()}, {k: torch.tensor(x).cuda() for k, x in lab.items()})[6])
    # (batch 1 at 128x128: the deepest batch-norms see 16 samples, which amplifies bf16 rounding; 15% here, 2-3% at real sizes)
    assert np.isfinite(l16) and abs(l16 - total) / total < 0.15, (l16, total)
    print(f"smoke: bf16 hipGraph step OK (total loss {l16:.5f})")
