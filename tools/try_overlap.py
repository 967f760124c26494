"""Same-box A/B of the step with one KeypointNet switch off / on / off / on: python tools/try_overlap.py [attribute]
(default overlap_wgrad: the weight gradients on a second HIP stream; e.g. fuse_dw_bwd, fuse_dw_bn, fuse_conv_bn)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer
from multiposenet_amd.synthetic import synthetic_batch

params = {"initial_learning_rate": 1e-4, "num_steps": 100000, "weight_decay": 0.0}
feats, labels = synthetic_batch(32, 512, 512)
attr = sys.argv[1] if len(sys.argv) > 1 else 'overlap_wgrad'
for overlap in (False, True, False, True):
    net = KeypointNet(dtype=torch.bfloat16, seed=0)
    setattr(net, attr, overlap)
    tr = Trainer(net, params, use_graph=True)
    for _ in range(8):
        tr.step(feats, labels)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        losses = tr.step(feats, labels)
    e1.record()
    torch.cuda.synchronize()
    print(f"{attr}={overlap}: {e0.elapsed_time(e1) / 30:.3f} ms per step, total loss {float(losses[-1]):.4f}", flush=True)
    del tr, net
    torch.cuda.empty_cache()
