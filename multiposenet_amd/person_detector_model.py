"""`model_fn` with the reference's contract (person_detector_model.py:8-81): the RetinaNet person detector on the frozen
MobileNet backbone - forward, anchor matching, losses and, in TRAIN mode, the whole optimizer step on the HIP kernels.

    spec = model_fn(features, labels, mode, params)

features {'images': [b,H,W,3] f32 in [0,1]} (H, W multiples of 128), labels {'boxes': f32 [b,max_boxes,4] normalised
(ymin, xmin, ymax, xmax), 'num_boxes': int32 [b]} (detector_pipeline.py), mode in ModeKeys.{TRAIN, EVAL}; params = the
PARAMS dict of train_person_detector.py:7-30 (keys read: depth_multiplier, weight_decay, score_threshold, iou_threshold,
max_boxes, localization_loss_weight, classification_loss_weight, gamma, alpha, num_steps, initial_learning_rate; optional
'dtype', 'seed', 'backbone_values', 'head_values').
"""
import torch

from .keypoints_model import EstimatorSpec, ModeKeys, _as_device
from .retinanet import LOSS_NAMES, PersonDetectorNet

_REGISTRY = {}


def get_detector(params):
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}[params.get("dtype", "bf16")]
    key = (params.get("model_dir"), float(params["depth_multiplier"]), dt, int(params.get("seed", 0)))
    if key not in _REGISTRY:
        _REGISTRY[key] = PersonDetectorNet(backbone_values=params.get("backbone_values"), head_values=params.get("head_values"),
                                           depth_multiplier=params["depth_multiplier"], dtype=dt, seed=int(params.get("seed", 0)))
    return _REGISTRY[key]


def reset_registry():
    _REGISTRY.clear()


def model_fn(features, labels, mode, params):
    assert mode != ModeKeys.PREDICT                                    # person_detector_model.py:10
    net = get_detector(params)
    images = _as_device(features["images"], device=net.device)
    gt = {"boxes": _as_device(labels["boxes"], torch.float32, net.device), "num_boxes": _as_device(labels["num_boxes"], torch.int32, net.device)}
    if images.shape[1] % 128 or images.shape[2] % 128:
        raise ValueError("image height and width must be multiples of 128 (detector/constants.py:4)")
    if mode == ModeKeys.TRAIN:
        losses = net.train_step(images, gt, params)
        named = {n: losses[i] for i, n in enumerate(LOSS_NAMES)}
        return EstimatorSpec(mode, named["total_loss"], "applied", None, named)
    assert images.shape[0] == 1                                        # person_detector_model.py:49-51 (evaluation: batch size 1)
    b = net.forward(images, False)
    predictions = net.check_nms(net.nms(b, params["score_threshold"], params["iou_threshold"], params["max_boxes"]))
    predictions.pop("overflow")                                        # (checked: the spec carries the reference's three keys)
    net.create_targets(gt)
    losses = net.compute_losses(params, with_grad=False)
    named = {n: losses[i] for i, n in enumerate(LOSS_NAMES)}
    return EstimatorSpec(mode, named["total_loss"], None, predictions, named)
