"""`model_fn` with the reference's contract (keypoints_model.py:6): forward, losses, and - in TRAIN mode -
the whole optimizer step, executed on the device by the HIP kernels.

    spec = model_fn(features, labels, mode, params)

features {'images': [b,H,W,3] f32 in [0,1]}, labels as produced by KeypointPipeline
(keypoints_detector_pipeline.py:104-110), mode in ModeKeys.{TRAIN, EVAL} (PREDICT is refused exactly like
keypoints_model.py:8), params = the PARAMS dict of train_keypoints.py:7-23 (keys read: depth_multiplier,
weight_decay, initial_learning_rate, num_steps; optional build-specific keys: 'dtype' ('bf16'|'f32'),
'seed', 'use_graph', 'distributed').

tf.estimator owns the variables in the reference; here a process-wide registry keeps one KeypointNet
(+ optimizer state) per distinct model configuration, so repeated calls continue training.
"""
from collections import namedtuple

import torch

from . import ops
from .net import KeypointNet
from .train import Trainer


class ModeKeys:   # values of tf.estimator.ModeKeys
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"


EstimatorSpec = namedtuple("EstimatorSpec", ["mode", "loss", "train_op", "eval_metric_ops", "losses"])

_REGISTRY = {}


def _as_device(t, dtype=None, device=None):
    if not torch.is_tensor(t):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(t))
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if device is not None and t.device != device:
        t = t.to(device)
    elif not t.is_cuda:
        t = t.cuda()
    return t.contiguous()


def get_trainer(params):
    """The (net, trainer) pair model_fn uses for `params` (created on first use)."""
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}[params.get("dtype", "bf16")]
    key = (params.get("model_dir"), float(params["depth_multiplier"]), dt, int(params.get("seed", 0)))
    if key not in _REGISTRY:
        distributed = bool(params.get("distributed", False))
        device = torch.device("cuda", torch.cuda.current_device())
        if distributed:
            # one process per GPU under torch.distributed.run: join the process group and bind this rank's device BEFORE
            # the network is built (identical replicas from the same seed; gradients averaged by the Trainer's reducer)
            from .parallel import init_distributed
            _, local_rank, _ = init_distributed()
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
        net = KeypointNet(values=params.get("initial_values"), depth_multiplier=params["depth_multiplier"], dtype=dt,
                          seed=int(params.get("seed", 0)), device=device)
        _REGISTRY[key] = Trainer(net, params, use_graph=bool(params.get("use_graph", True)), distributed=distributed)
    return _REGISTRY[key]


def reset_registry():
    _REGISTRY.clear()


def model_fn(features, labels, mode, params):
    assert mode != ModeKeys.PREDICT                                    # keypoints_model.py:8
    is_training = mode == ModeKeys.TRAIN
    trainer = get_trainer(params)
    dev = trainer.net.device
    feats = {"images": _as_device(features["images"], device=dev)}
    labs = {"heatmaps": _as_device(labels["heatmaps"], torch.float32, dev),
            "loss_masks": _as_device(labels["loss_masks"], torch.float32, dev),
            "segmentation_masks": _as_device(labels["segmentation_masks"], torch.float32, dev),
            "num_boxes": _as_device(labels["num_boxes"], torch.int32, dev)}
    if feats["images"].shape[1] % 128 or feats["images"].shape[2] % 128:
        raise ValueError("image height and width must be multiples of 128 (detector/constants.py:4)")
    if is_training:
        losses = trainer.step(feats, labs)
    else:
        losses = trainer.eval_step(feats, labs)
    named = {n: losses[i] for i, n in enumerate(ops.LOSS_NAMES)}
    total = named["total_loss"]
    if mode == ModeKeys.EVAL:                                           # keypoints_model.py:92-105
        metrics = {
            "eval_regression_loss": named["regression_loss"],
            "eval_focal_loss": named["focal_loss"],
            "eval_per_pixel_reg_loss": named["per_pixel_reg_loss"],
            "eval_segmentation_loss_at_level_2": named["segmentation_loss_at_level_2"],
            "eval_segmentation_loss_at_level_5": named["segmentation_loss_at_level_5"],
        }
        return EstimatorSpec(mode, total, None, metrics, named)
    return EstimatorSpec(mode, total, "applied", None, named)
