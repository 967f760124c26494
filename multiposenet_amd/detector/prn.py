"""`prn(x, is_training)` with the reference's name and argument meaning (detector/prn.py:5-25)."""
import numpy as np
import torch

from .. import _lib
from ..prn import PoseResidualNet

_nets = {}


def prn(x, is_training, values=None, dtype=torch.bfloat16, scope="PRN"):
    """x: float [b, h, w, c] (numpy or CUDA tensor). Returns the logits [b, h, w, c] float32 (CUDA tensor).
    `values`: variables by reference name ('PRN/fc1/weights', ...); is_training is accepted for signature parity (the
    reference's dropout is commented out, prn.py:21). The network of a (shape, dtype, scope) is built once - the stand-in
    for tf.variable_scope('PRN') - and re-loaded whenever a different `values` mapping is passed (the cache holds a
    reference to the mapping it loaded, so identities cannot be recycled under it)."""
    x = _lib.to_device_f32(x)      # (the process's device - one process per GPU -, or the tensor's own)
    key = (tuple(x.shape), dtype, scope, x.device.index)
    entry = _nets.get(key)
    if entry is None:
        b, h, w, c = x.shape
        entry = _nets[key] = [PoseResidualNet(values=values, batch=b, h=h, w=w, c=c, dtype=dtype, device=x.device), values]
    elif values is not None and values is not entry[1]:
        entry[0].load_state_dict(values)
        entry[1] = values
    return entry[0].predict(x)
