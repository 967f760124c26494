"""`prn(x, is_training)` with the reference's name and argument meaning (detector/prn.py:5-25)."""
import numpy as np
import torch

from ..prn import PoseResidualNet

_nets = {}


def prn(x, is_training, values=None, dtype=torch.bfloat16):
    """x: float [b, h, w, c] (numpy or CUDA tensor). Returns the logits [b, h, w, c] float32 (CUDA tensor).
    `values`: variables by reference name ('PRN/fc1/weights', ...); is_training is accepted for signature parity (the
    reference's dropout is commented out, prn.py:21)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    x = x.to("cuda:0", torch.float32).contiguous()
    key = (tuple(x.shape), dtype, id(values))
    if key not in _nets:
        b, h, w, c = x.shape
        _nets[key] = PoseResidualNet(values=values, batch=b, h=h, w=w, c=c, dtype=dtype)
    return _nets[key].predict(x)
