"""RetinaNet with the reference's constructor and method contract (detector/retinanet.py:13-166)."""
import torch

from .. import variables
from ..retinanet import LOSS_NAMES  # noqa: F401


class RetinaNet:
    def __init__(self, backbone_features, image_shape, is_training, params, net=None):
        """
        Arguments:
            backbone_features: a dict with FeatureMaps, keys ['c2', 'c3', 'c4', 'c5'] (from mobilenet_v1 with is_training=False).
            image_shape: [b, h, w, 3] (ints or a tensor), as tf.shape(images).
            is_training: a boolean.
            params: a dict (read by `loss`: gamma, alpha).
        Attributes: anchors (float32 CUDA tensor [num_anchors, 4]), raw_predictions {'encoded_boxes': [b, num_anchors, 4],
        'class_predictions': [b, num_anchors]}.
        """
        net = net or variables.get_default_detector(params.get("depth_multiplier", 1.0) if params else 1.0)
        shape = [int(v) for v in (image_shape.tolist() if torch.is_tensor(image_shape) else image_shape)]
        n, h, w = shape[0], shape[1], shape[2]
        b = net._buffers(n, h, w)
        feats = {k: (f.raw, f.affine) for k, f in backbone_features.items()}
        net.head_forward(feats, b, is_training)
        self._net, self._b = net, b
        self.anchors = b["anchors"]
        self.raw_predictions = net.raw_predictions(b)

    def get_predictions(self, score_threshold=0.05, iou_threshold=0.5, max_detections=25):
        """{'boxes': [b, N, 4], 'scores': [b, N], 'num_boxes': [b]} with N = max_detections (retinanet.py:60-84)."""
        pred = self._net.check_nms(self._net.nms(self._b, score_threshold, iou_threshold, max_detections))
        pred.pop("overflow")                                            # (checked here: the reference's three keys)
        return pred

    def loss(self, groundtruth, params):
        """{'localization_loss', 'classification_loss'}: scalar device tensors (retinanet.py:86-144)."""
        # THIS object's forward pass: another RetinaNet built on the shared detector in between (another image shape)
        # must not redirect the loss to its outputs and anchors
        self._net.create_targets(groundtruth, b=self._b)
        losses = self._net.compute_losses(dict(params, weight_decay=0.0, localization_loss_weight=1.0, classification_loss_weight=1.0),
                                          with_grad=False, b=self._b)
        return {"localization_loss": losses[0], "classification_loss": losses[1]}
