"""KeypointSubnet with the reference's constructor contract (detector/keypoint_subnet.py:10-62)."""
from .. import variables

DEPTH = 128  # detector/keypoint_subnet.py:7


class KeypointSubnet:
    def __init__(self, backbone_features, is_training, params, net=None):
        """
        Arguments:
            backbone_features: a dict with FeatureMaps, keys ['c2', 'c3', 'c4', 'c5'] (from mobilenet_v1).
            is_training: a boolean.
            params: a dict (unused by the reference as well).
        Attributes (NHWC, like the reference after its final transposes, keypoint_subnet.py:56-62):
            heatmaps: f32 [b, h/4, w/4, 18] logits (17 keypoints + 1 person segmentation).
            enriched_features: {'p2'..'p5'}: the PRE-batch-norm FPN outputs [b, h/2^l, w/2^l, 128].
        """
        c5 = backbone_features["c5"]
        net = net or getattr(c5, "_net", None) or variables.get_default_net()
        n, _, h5, w5 = c5.shape
        b = net._buffers(n, h5 * 32, w5 * 32)
        feats = {k: (f.raw, f.affine) for k, f in backbone_features.items()}
        self.heatmaps = net.subnet_forward(feats, is_training, b)
        self.enriched_features = {f"p{l}": b["p"][l] for l in (2, 3, 4, 5)}
        net._last = (b, feats, getattr(c5, "_images", None))
