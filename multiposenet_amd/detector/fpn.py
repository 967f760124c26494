"""feature_pyramid_network with the reference's signature (detector/fpn.py:6-55), keypoint configuration only."""
from .. import ops, variables
from ..net import DEPTH
from .feature_map import FeatureMap


def feature_pyramid_network(features, is_training, depth, min_level=3, add_coarse_features=True, scope='fpn', net=None):
    """Only the configuration the keypoint path uses is on the hot path (detector/keypoint_subnet.py:20-23):
    depth=128, min_level=2, add_coarse_features=False, scope='keypoint_fpn'. The RetinaNet variant (min_level=3 with the
    coarse p6 / p7 branch, detector/fpn.py:42-46, scope 'fpn') lives in multiposenet_amd.retinanet.PersonDetectorNet /
    detector.RetinaNet. Returns {'p2'..'p5': FeatureMap} (raw outputs, no batch-norm)."""
    if add_coarse_features or min_level != 2 or depth != DEPTH or scope != 'keypoint_fpn':
        raise NotImplementedError("this function builds the keypoint FPN (depth=128, min_level=2, add_coarse_features=False, "
                                  "scope='keypoint_fpn'); the detector's FPN is part of detector.RetinaNet")
    c5 = features["c5"]
    net = net or getattr(c5, "_net", None) or variables.get_default_net()
    n, _, h5, w5 = c5.shape
    b = net._buffers(n, h5 * 32, w5 * 32)
    prev = None
    for l in (5, 4, 3, 2):
        f = features[f"c{l}"]
        ops.conv_fwd(f.raw, net.lateral[l].packed.fwd, DEPTH, 1, f.affine, out=b["x"][l], up_res=prev)
        prev = b["x"][l]
        ops.conv_fwd(prev, net.pconv[l].packed.fwd, DEPTH, 3, None, out=b["p"][l])
    return {f"p{l}": FeatureMap(b["p"][l]) for l in (2, 3, 4, 5)}
