"""`model_fn(features, labels, mode, params)` of the reference's prn_model.py:5-57 on the HIP kernels.

features: float32 [b, 56, 36, 17] person-crop heatmaps (numpy or CUDA tensor), labels: the same shape (one-hot peaks per
visible keypoint, prn_pipeline.py); mode: ModeKeys.TRAIN | EVAL; params: {'initial_learning_rate', 'num_steps'} (+ optional
'dtype': 'bf16' | 'f32', 'values': initial variables by reference name). TRAIN performs the optimizer step.
"""
import numpy as np
import torch

from .keypoints_model import EstimatorSpec, ModeKeys
from . import _lib
from .prn import PoseResidualNet

_models = {}


def _model(params, batch, shape, device=None):
    """One set of variables (+ optimizer state) per model configuration: tf.estimator keys its variables by `model_dir`
    (train_prn.py: RunConfig(model_dir=...)), so does this registry - never by object identity, which is recycled, and
    never by batch size: the reference's eval pipeline ends on a partial batch (`dataset.repeat(1).batch(b)`), which must
    score the TRAINED variables. Activation buffers are per batch size (`PoseResidualNet.for_batch`)."""
    # storage type of the GEMM operands and activations (masters, accumulators and Adam stay f32); "fp16" is the type
    # BASELINE.json config 5 names
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16}[params.get("dtype", "bf16")]
    dev = _lib.current_device() if device is None else device
    key = (params.get("model_dir"), dt, int(params.get("seed", 0)), shape, dev.index)
    if key not in _models:
        _models[key] = PoseResidualNet(values=params.get("values"), batch=batch, h=shape[0], w=shape[1], c=shape[2], dtype=dt,
                                       device=dev, seed=int(params.get("seed", 0)))
    return _models[key].for_batch(batch)


def reset_registry():
    _models.clear()


def _dev(a):
    return _lib.to_device_f32(a)      # (the process's device - one process per GPU -, or the tensor's own)


def model_fn(features, labels, mode, params):
    assert mode != ModeKeys.PREDICT                       # prn_model.py:7
    x, y = _dev(features), _dev(labels)
    if x.dim() != 4 or x.shape != y.shape:
        raise ValueError(f"features and labels must be [b, h, w, c] of equal shape, got {tuple(x.shape)} / {tuple(y.shape)}")
    net = _model(params, x.shape[0], tuple(x.shape[1:]), x.device)
    if mode == ModeKeys.TRAIN:
        loss = net.train_step(x, y, float(params["initial_learning_rate"]), int(params["num_steps"]))
        return EstimatorSpec(mode=mode, loss=loss, train_op=net.global_step, eval_metric_ops=None, losses={"logloss": loss})
    net.forward(x)
    loss = net.loss(y, with_grad=False)
    return EstimatorSpec(mode=mode, loss=loss, train_op=None, eval_metric_ops={"eval_loss": loss}, losses={"logloss": loss})
