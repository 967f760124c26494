"""How sensitive is the step's gradient to bf16 storage at a TRAINED point (vs random initialisation, where a conv+BN stack
amplifies any forward perturbation ~1.2x per layer)? Trains the f32 build N steps on one 2 x 128 x 128 batch, then compares, on
the trained variables: HIP bf16 step gradients (fused / unfused reductions), the f64 oracle, and the f64 oracle with bf16
storage emulation.   python tools/dbg_trained_sensitivity.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import network as onet
from test_network_gpu import _labels, _params
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(14)
B, H, W = 2, 128, 128
params = _params(8)
img = rs.rand(B, H, W, 3).astype(np.float32)
lab = _labels(rs, B, H // 4, W // 4)
hp = {"initial_learning_rate": 1e-3, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
feats = {"images": torch.tensor(img).cuda()}
dlab = {k: torch.tensor(val).cuda() for k, val in lab.items()}
net = KeypointNet(values=params, dtype=torch.float32)
tr = Trainer(net, hp, use_graph=True)
for i in range(steps):
    l = tr.step(feats, dlab)
print("trained", steps, "steps: total loss", float(l[6]))
trained = net.state_dict()
ref = {k: v.astype(np.float64) for k, v in trained.items()}
zeros = lambda: {k: np.zeros_like(v) for k, v in ref.items()}
def oracle():
    t, _, g = onet.train_step({k: v.copy() for k, v in ref.items()}, zeros(), zeros(), img, lab, 0, hp, dtype=torch.float64)
    return t, g
t_ex, g_ex = oracle()
with onet.storage_emulation(torch.bfloat16):
    t_em, g_em = oracle()
keys = sorted(g_ex)
cat = lambda g: np.concatenate([np.asarray(g[k], np.float64).ravel() for k in keys])
ex, em = cat(g_ex), cat(g_em)
print("oracle loss exact %.5f, emulated %.5f; emulated vs exact gradients rel-L2 %.4f" % (t_ex, t_em, np.linalg.norm(em - ex) / np.linalg.norm(ex)))
for fused in (False, True):
    n2 = KeypointNet(values=trained, dtype=torch.bfloat16)
    n2.fuse_conv_bn = fused
    ls = Trainer(n2, hp, use_graph=False).step(feats, dlab)
    g = cat({k: n2.grads[k].cpu().numpy() for k in keys})
    print("HIP bf16 fused=%d loss %.5f: vs emulated %.4f, vs exact %.4f" % (fused, float(ls[6]), np.linalg.norm(g - em) / np.linalg.norm(em), np.linalg.norm(g - ex) / np.linalg.norm(ex)))
n3 = KeypointNet(values=trained, dtype=torch.float32)
ls = Trainer(n3, hp, use_graph=False).step(feats, dlab)
g = cat({k: n3.grads[k].cpu().numpy() for k in keys})
print("HIP f32 loss %.5f: vs exact %.5f" % (float(ls[6]), np.linalg.norm(g - ex) / np.linalg.norm(ex)))
