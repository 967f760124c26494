"""conv2d_same / batch_norm_relu with the reference's signatures (detector/utils/layer_utils.py:9,19).

The reference creates a variable per call under the current tf scope; here `name` must be the full name of a variable
of the default net (e.g. 'keypoint_fpn/p3' -> variable 'keypoint_fpn/p3/kernel')."""
import torch

from ... import ops, variables
from ..._lib import ACT_NONE, ACT_RELU
from ..feature_map import FeatureMap

BATCH_NORM_MOMENTUM = 0.95
BATCH_NORM_EPSILON = 1e-3


def _fm(x):
    return x if isinstance(x, FeatureMap) else FeatureMap(x)


def conv2d_same(x, num_filters, kernel_size=3, stride=1, name=None, net=None):
    assert kernel_size in [1, 3]
    assert stride in [1, 2]
    if stride != 1:
        raise NotImplementedError("stride 2 is only used by the RetinaNet p6/p7 branch (out of scope)")
    net = net or variables.get_default_net()
    conv = next((c for c in net.convs if c.name == f"{name}/kernel"), None)
    if conv is None or conv.cout != num_filters or conv.ksize != kernel_size:
        raise KeyError(f"no {kernel_size}x{kernel_size} conv variable '{name}/kernel' with {num_filters} filters")
    x = _fm(x)
    return FeatureMap(ops.conv_fwd(x.raw, conv.packed.fwd, conv.cout, conv.ksize, x.affine))


def batch_norm_relu(x, is_training, use_relu=True, name=None, net=None):
    net = net or variables.get_default_net()
    bn = next((b for b in net.all_bn if b.name == name), None)
    if bn is None:
        raise KeyError(f"no batch-norm variables '{name}/gamma' ...")
    x = _fm(x)
    raw = x.tensor()
    if is_training:
        part, nparts = ops.bn_stats(raw)
        ops.bn_finalize(bn, part, nparts, raw.numel() // raw.shape[3])
    else:
        ops.bn_inference_affine(bn)
    return FeatureMap(raw, ops.Affine(bn.scale, bn.shift, ACT_RELU if use_relu else ACT_NONE))
