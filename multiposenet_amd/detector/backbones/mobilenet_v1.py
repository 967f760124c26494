"""mobilenet_v1 with the reference's signature (detector/backbones/mobilenet_v1.py:11)."""
import torch

from ... import variables
from ..feature_map import FeatureMap


def mobilenet_v1(images, is_training, depth_multiplier=1.0, net=None):
    """
    Arguments:
        images: a float tensor with shape [b, h, w, 3], RGB with pixel values in [0, 1] (torch CUDA tensor or numpy);
            uint8 is accepted too and scaled by 1/255 on load (create_pb.py:167).
        is_training: a boolean.
        depth_multiplier: a float number, multiplier for the number of filters in a layer.
    Returns:
        a dict with four FeatureMaps: 'c2', 'c3', 'c4', 'c5' (outputs of pointwise 3, 5, 11, 13; mobilenet_v1.py:76-79).
    """
    net = net or variables.get_default_net(depth_multiplier)
    if not torch.is_tensor(images):
        images = torch.from_numpy(images)
    images = images.to(net.device).contiguous()
    feats = net.backbone_forward(images, is_training)
    out = {k: FeatureMap(raw, aff) for k, (raw, aff) in feats.items()}
    for f in out.values():
        f._net, f._images = net, images
    return out
