"""Oracle for the pose residual network (TEST INFRASTRUCTURE, not shipped code).

torch-CPU restatement (f32 / f64, backward by autograd) of `prn` (reference detector/prn.py:5-25) and of the loss /
optimizer of prn_model.py:5-57 in TensorFlow-1.15 semantics (slim.fully_connected = matmul + bias + ReLU with variables
`weights` [in,out] / `biases`; tf.nn.softmax over axis 1; tf.losses.log_loss with epsilon 1e-7, Reduction.NONE, then
reduce_mean; cosine_decay alpha 1e-4; AdamOptimizer defaults, NO gradient clipping).
PARITY UNPINNED: the arithmetic lives in tensorflow==1.15, which cannot be imported here (see oracle/network.py).
"""
import math

import numpy as np
import torch

from .network import adam_step, cosine_decay

H, W, C = 56, 36, 17            # CROP_SIZE (detector/constants.py) x 17 keypoints
HIDDEN = 1024                    # prn.py:20


def param_shapes(h=H, w=W, c=C, hidden=HIDDEN):
    n = h * w * c
    return {"PRN/fc1/weights": (n, hidden), "PRN/fc1/biases": (hidden,),
            "PRN/fc2/weights": (hidden, n), "PRN/fc2/biases": (n,)}


def init_params(seed=0, h=H, w=W, c=C, hidden=HIDDEN, dtype=np.float32):
    """tf.variance_scaling_initializer() (scale 1, fan_in, truncated normal) for weights (prn.py:19), zeros for biases."""
    rs = np.random.RandomState(seed)
    p = {}
    for k, shp in param_shapes(h, w, c, hidden).items():
        if k.endswith("weights"):
            std = math.sqrt(1.0 / shp[0]) / 0.87962566103423978
            p[k] = np.clip(rs.randn(*shp), -2, 2).astype(dtype) * dtype(std)
        else:
            p[k] = np.zeros(shp, dtype)
    return p


def prn(x, p):
    """prn.py:5-25.  x [b,h,w,c] -> [b,h,w,c] (logits = x + relu(fc2(relu(fc1(flatten x)))))."""
    b = x.shape[0]
    flat = x.reshape(b, -1)
    y = torch.relu(flat @ p["PRN/fc1/weights"] + p["PRN/fc1/biases"])
    y = torch.relu(y @ p["PRN/fc2/weights"] + p["PRN/fc2/biases"])
    return (flat + y).reshape(x.shape)


def log_loss(labels, logits):
    """prn_model.py:16-30: softmax over the h*w axis per channel, tf.losses.log_loss(eps 1e-7), mean."""
    b, h, w, c = logits.shape
    lab = labels.reshape(b, h * w, c)
    prob = torch.softmax(logits.reshape(b, h * w, c), dim=1)
    eps = 1e-7
    losses = -lab * torch.log(prob + eps) - (1.0 - lab) * torch.log(1.0 - prob + eps)
    return losses.mean()


def train_step(params_np, m_np, v_np, x, labels, global_step, hp, dtype=torch.float32):
    """One TRAIN step of prn_model.model_fn on numpy state, in place. Returns (loss, grads dict, logits)."""
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params_np.items()}
    logits = prn(torch.as_tensor(x, dtype=dtype), p)
    loss = log_loss(torch.as_tensor(labels, dtype=dtype), logits)
    loss.backward()
    lr = cosine_decay(hp["initial_learning_rate"], global_step, hp["num_steps"])
    grads = {}
    for k in params_np:
        g = p[k].grad.detach().numpy().astype(np.float64)
        grads[k] = g
        adam_step(params_np[k], g.astype(params_np[k].dtype), m_np[k], v_np[k], lr, global_step + 1, clip=float("inf"))
    return float(loss), grads, logits.detach().numpy()
