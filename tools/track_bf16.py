"""bf16 vs f32 loss tracking over a few steps on the toy batch of tests/test_network_gpu.py (diagnostic):
python tools/track_bf16.py [seed ...]"""
import os, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_network_gpu import _params, _labels
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer

seeds = [int(a) for a in sys.argv[1:]] or [4]
for seed in seeds:
    rs = np.random.RandomState(seed)
    B, H, W = 2, 128, 128
    params = _params(2)
    img = torch.tensor(rs.rand(B, H, W, 3).astype(np.float32)).cuda()
    dlab = {k: torch.tensor(val).cuda() for k, val in _labels(rs, B, H // 4, W // 4).items()}
    hp = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        net = KeypointNet(values=params, dtype=dt)
        tr = Trainer(net, hp, use_graph=False)
        out[dt] = [float(tr.step({"images": img}, dlab).cpu().numpy()[6]) for _ in range(4)]
    f, b = np.array(out[torch.float32]), np.array(out[torch.bfloat16])
    print("seed", seed, "f32", np.round(f, 3), "bf16", np.round(b, 3), "rel", np.round(b / f - 1, 4))
