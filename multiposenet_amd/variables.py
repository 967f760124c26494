"""Process-wide default KeypointNet - the stand-in for TensorFlow's variable scopes.

The reference builds its layers with tf.get_variable under global name scopes (`MobilenetV1/...`,
`keypoint_fpn/...`, detector/backbones/mobilenet_v1.py:43, detector/fpn.py:36, detector/keypoint_subnet.py:31); the functional
API in multiposenet_amd.detector resolves the same names in the net registered here.
"""
import torch

_DEFAULT = {}


def get_default_net(depth_multiplier=1.0, dtype=torch.bfloat16):
    key = (float(depth_multiplier), dtype)
    if key not in _DEFAULT:
        from .net import KeypointNet
        _DEFAULT[key] = KeypointNet(depth_multiplier=depth_multiplier, dtype=dtype)
    return _DEFAULT[key]


def set_default_net(net):
    _DEFAULT[(float(net.dm), net.dtype)] = net
    return net


def reset_default_nets():
    _DEFAULT.clear()
