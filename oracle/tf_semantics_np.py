"""Independent plain-numpy (explicit loops) restatement of the TensorFlow-1.15 op semantics the
oracle depends on. TEST INFRASTRUCTURE ONLY; used to cross-check oracle/network.py on small cases.

Each function follows TF's documented behaviour for the op the reference calls:
  conv_same       - tf.nn.conv2d / slim.conv2d padding='SAME'   (mobilenet_v1.py:56,73)
  depthwise_same  - tf.nn.depthwise_conv2d 'SAME'                 (mobilenet_v1.py:101)
  nearest_up2     - tf.image.resize_nearest_neighbor              (fpn.py:71)
  bilinear_legacy - tf.image.resize_bilinear (align_corners=False, no half-pixel) (keypoint_subnet.py:86)
All arrays NHWC, weights HWIO.
The second half of the file restates, loop by loop and without torch, the remaining TF arithmetic of the TRAIN step
(fused batch-norm + moving-average update, Adam, cosine decay, sigmoid cross-entropy / focal loss, l2_loss, clipping,
resize_bilinear as used to halve the loss masks), each citing the TF 1.15 source file it follows.
"""
import math

import numpy as np


def _same_pad(size, k, s):
    out = int(math.ceil(size / s))
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def conv_same(x, w, stride):
    n, h, ww, ci = x.shape
    k, _, _, co = w.shape
    oh, pt = _same_pad(h, k, stride)
    ow, pl = _same_pad(ww, k, stride)
    y = np.zeros((n, oh, ow, co), np.float64)
    for oy in range(oh):
        for ox in range(ow):
            for ky in range(k):
                for kx in range(k):
                    iy, ix = oy * stride + ky - pt, ox * stride + kx - pl
                    if 0 <= iy < h and 0 <= ix < ww:
                        y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[ky, kx].astype(np.float64)
    return y


def depthwise_same(x, w, stride):
    n, h, ww, c = x.shape
    k = w.shape[0]
    oh, pt = _same_pad(h, k, stride)
    ow, pl = _same_pad(ww, k, stride)
    y = np.zeros((n, oh, ow, c), np.float64)
    for oy in range(oh):
        for ox in range(ow):
            for ky in range(k):
                for kx in range(k):
                    iy, ix = oy * stride + ky - pt, ox * stride + kx - pl
                    if 0 <= iy < h and 0 <= ix < ww:
                        y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) * w[ky, kx, :, 0].astype(np.float64)
    return y


def nearest_up2(x):
    n, h, w, c = x.shape
    y = np.zeros((n, 2 * h, 2 * w, c), x.dtype)
    for i in range(2 * h):
        for j in range(2 * w):
            y[:, i, j] = x[:, i // 2, j // 2]
    return y


def bilinear_legacy(x, oh, ow):
    n, h, w, c = x.shape
    y = np.zeros((n, oh, ow, c), np.float64)
    sy, sx = h / oh, w / ow
    for i in range(oh):
        fy = i * sy
        y0 = int(math.floor(fy)); y1 = min(y0 + 1, h - 1); ly = fy - y0
        for j in range(ow):
            fx = j * sx
            x0 = int(math.floor(fx)); x1 = min(x0 + 1, w - 1); lx = fx - x0
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
            y[:, i, j] = top + (bot - top) * ly
    return y


# =====================================================================================================================
# Second part: loop-level, torch-free restatements of the TensorFlow-1.15 arithmetic the keypoint TRAIN step goes
# through besides the convolutions - each following the TF 1.15 source file named in its docstring (paths relative to
# the tensorflow/ source tree at tag v1.15.0; TensorFlow itself is absent from /root/reference and from this image, so
# these are restated from the published sources, NOT executed against them: PARITY UNPINNED).
# tests/test_oracle_network.py pins each one with hand-computed known-answer vectors and then checks oracle/network.py
# (the torch restatement the HIP kernels are compared with) against them on random inputs.
# =====================================================================================================================

def fused_batch_norm_op_train(x, gamma, beta, eps):
    """The FusedBatchNorm op in training mode, NHWC view [rows, C] per channel.

    tensorflow/core/kernels/fused_batch_norm_op.cc, `struct FusedBatchNorm<CPUDevice, T, U>`:
        mean        = sum(x) / rest_size
        variance    = sum((x - mean)^2) / rest_size                      (biased; normalises the batch)
        y           = (x - mean) * rsqrt(variance + epsilon) * scale + offset
        batch_mean  = mean
        batch_var   = variance * rest_size / (rest_size - 1)             (Bessel-corrected: what the op RETURNS;
                                                                          rest_size_adjust = rest_size / max(rest_size - 1, 1))
    Returns (y, batch_mean, batch_var_unbiased, variance_biased)."""
    x = np.asarray(x, np.float64)
    rows, C = x.reshape(-1, x.shape[-1]).shape
    flat = x.reshape(rows, C)
    y = np.zeros_like(flat)
    mean = np.zeros(C)
    var = np.zeros(C)
    for c in range(C):
        s = 0.0
        for r in range(rows):
            s += flat[r, c]
        mean[c] = s / rows
        q = 0.0
        for r in range(rows):
            q += (flat[r, c] - mean[c]) ** 2
        var[c] = q / rows
        inv = 1.0 / math.sqrt(var[c] + eps)
        for r in range(rows):
            y[r, c] = (flat[r, c] - mean[c]) * inv * gamma[c] + beta[c]
    adjust = rows / max(rows - 1, 1)
    return y.reshape(x.shape), mean, var * adjust, var


def batch_norm_layer(x, gamma, beta, moving_mean, moving_var, momentum, eps, training):
    """tf.layers.batch_normalization(..., fused=True) as the reference calls it (mobilenet_v1.py:29-38,
    layer_utils.py:9-16) = keras BatchNormalization in TF 1.15.

    tensorflow/python/keras/layers/normalization.py, `BatchNormalizationBase._fused_batch_norm`:
        training : output, mean, variance = nn.fused_batch_norm(inputs, gamma, beta, epsilon)   (variance = the op's
                   Bessel-corrected batch_var; `self._bessels_correction_test_only` is True by default, so the
                   "remove Bessel's correction" branch does NOT run)
                   new moving_mean = _assign_moving_average(moving_mean, mean, momentum)
                   new moving_var  = _assign_moving_average(moving_variance, variance, momentum)
        inference: nn.fused_batch_norm(inputs, gamma, beta, mean=moving_mean, variance=moving_variance, is_training=False)
    `_assign_moving_average` (same file; tensorflow/python/training/moving_averages.py for the v1 path):
        decay = 1 - momentum;  variable -= (variable - value) * decay
    Returns (y, new_moving_mean, new_moving_var)."""
    if training:
        y, mean, var_unbiased, _ = fused_batch_norm_op_train(x, gamma, beta, eps)
        decay = 1.0 - momentum
        new_mean = np.array([moving_mean[c] - (moving_mean[c] - mean[c]) * decay for c in range(len(mean))])
        new_var = np.array([moving_var[c] - (moving_var[c] - var_unbiased[c]) * decay for c in range(len(mean))])
        return y, new_mean, new_var
    x = np.asarray(x, np.float64)
    C = x.shape[-1]
    flat = x.reshape(-1, C)
    y = np.zeros_like(flat)
    for c in range(C):
        inv = 1.0 / math.sqrt(moving_var[c] + eps)     # fused_batch_norm_op.cc inference: (x - est_mean) * rsqrt(est_var + eps)
        for r in range(flat.shape[0]):
            y[r, c] = (flat[r, c] - moving_mean[c]) * inv * gamma[c] + beta[c]
    return y.reshape(x.shape), np.asarray(moving_mean, np.float64), np.asarray(moving_var, np.float64)


def clip_by_value(t, lo, hi):
    """tensorflow/python/ops/clip_ops.py `clip_by_value`: minimum(maximum(t, clip_value_min), clip_value_max)
    (keypoints_model.py:119 clips every gradient to [-200, 200])."""
    return [min(max(float(v), lo), hi) for v in np.asarray(t, np.float64).reshape(-1)]


def adam_apply(var, m, v, beta1_power, beta2_power, lr, grad, beta1=0.9, beta2=0.999, eps=1e-8):
    """One tf.train.AdamOptimizer apply on flat lists (keypoints_model.py:117-120).

    tensorflow/core/kernels/training_ops.cc, `struct ApplyAdam<CPUDevice, T>` (use_nesterov = false):
        alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)
        m    += (g - m) * (1 - beta1)
        v    += (g * g - v) * (1 - beta2)
        var  -= (m * alpha) / (sqrt(v) + epsilon)          (epsilon OUTSIDE the bias correction: "epsilon hat")
    tensorflow/python/training/adam.py: `_create_slots` creates beta1_power = beta1, beta2_power = beta2 (so the FIRST
    apply sees beta^1), `_finish` multiplies both by beta after every apply.
    Returns (var, m, v, beta1_power, beta2_power) after the apply."""
    alpha = lr * math.sqrt(1.0 - beta2_power) / (1.0 - beta1_power)
    out_var, out_m, out_v = [], [], []
    for i in range(len(var)):
        g = float(grad[i])
        mi = m[i] + (g - m[i]) * (1.0 - beta1)
        vi = v[i] + (g * g - v[i]) * (1.0 - beta2)
        out_var.append(var[i] - (mi * alpha) / (math.sqrt(vi) + eps))
        out_m.append(mi)
        out_v.append(vi)
    return out_var, out_m, out_v, beta1_power * beta1, beta2_power * beta2


def cosine_decay(initial_learning_rate, global_step, decay_steps, alpha=0.0):
    """tf.train.cosine_decay (keypoints_model.py:109-112, called there with alpha=1e-4; constants train_keypoints.py:17-18).

    tensorflow/python/training/learning_rate_decay.py `cosine_decay` -> tensorflow/python/keras/optimizer_v2/
    learning_rate_schedule.py `CosineDecay.__call__`:
        global_step_recomp = minimum(global_step, decay_steps)
        completed_fraction = global_step_recomp / decay_steps
        cosine_decayed     = 0.5 * (1.0 + cos(pi * completed_fraction))
        decayed            = (1 - alpha) * cosine_decayed + alpha
        return initial_learning_rate * decayed"""
    step = min(float(global_step), float(decay_steps))
    completed = step / float(decay_steps)
    cosine_decayed = 0.5 * (1.0 + math.cos(math.pi * completed))
    return initial_learning_rate * ((1.0 - alpha) * cosine_decayed + alpha)


def sigmoid_cross_entropy_with_logits(labels, logits):
    """tensorflow/python/ops/nn_impl.py `sigmoid_cross_entropy_with_logits`:
        relu_logits    = where(logits >= 0, logits, 0)
        neg_abs_logits = where(logits >= 0, -logits, logits)
        return relu_logits - logits * labels + log1p(exp(neg_abs_logits))"""
    out = []
    for z, x in zip(np.asarray(labels, np.float64).reshape(-1), np.asarray(logits, np.float64).reshape(-1)):
        relu = x if x >= 0 else 0.0
        neg_abs = -x if x >= 0 else x
        out.append(relu - x * z + math.log1p(math.exp(neg_abs)))
    return np.array(out).reshape(np.shape(logits))


def l2_loss(t):
    """tensorflow/core/kernels/l2loss_op.cc `L2LossOp<CPUDevice, T>`: output = sum(t * t) / 2."""
    s = 0.0
    for v in np.asarray(t, np.float64).reshape(-1):
        s += v * v
    return s / 2.0


def focal_loss(heatmaps, num_boxes, predictions, alpha=2.0, beta=4.0):
    """The reference's own loss, keypoints_model.py:141-178, element by element (NHWC, [b,h,w,c] -> [b,h,w]):
        is_extreme = (y == 1.0); ce = sigmoid_cross_entropy_with_logits(float(is_extreme), x); p = sigmoid(x)
        weights = is_extreme ? (1 - p)^alpha : (1 - y)^beta * p^alpha
        loss = sum_c(weights * ce) / (num_boxes + 1)"""
    y = np.asarray(heatmaps, np.float64)
    x = np.asarray(predictions, np.float64)
    b, h, w, c = y.shape
    out = np.zeros((b, h, w))
    for n in range(b):
        for i in range(h):
            for j in range(w):
                s = 0.0
                for k in range(c):
                    pos = y[n, i, j, k] == 1.0
                    ce = float(sigmoid_cross_entropy_with_logits([1.0 if pos else 0.0], [x[n, i, j, k]])[0])
                    p = 1.0 / (1.0 + math.exp(-x[n, i, j, k]))
                    wgt = (1.0 - p) ** alpha if pos else (1.0 - y[n, i, j, k]) ** beta * p ** alpha
                    s += wgt * ce
                out[n, i, j] = s / (float(num_boxes[n]) + 1.0)
    return out


def resize_bilinear_tf(x, oh, ow):
    """tf.image.resize_bilinear(align_corners=False) as TF 1.15 computes it (half_pixel_centers does not exist in the
    v1 Python signature the reference uses, keypoint_subnet.py:86 / keypoints_model.py:73-74).

    tensorflow/core/kernels/image_resizer_state.h: `CalculateResizeScale(in, out, align_corners=false) = in / float(out)`,
    `LegacyScaler: in = out_index * scale`;  tensorflow/core/kernels/resize_bilinear_op.cc:
        compute_interpolation_weights: lower = floor(in); upper = min(lower + 1, in_size - 1); lerp = in - lower
        compute_lerp(top_left, top_right, bottom_left, bottom_right, x_lerp, y_lerp):
            top    = top_left + (top_right - top_left) * x_lerp
            bottom = bottom_left + (bottom_right - bottom_left) * x_lerp
            return top + (bottom - top) * y_lerp                       (x first, then y)
    Scale and source coordinates are float32 in TF; so are they here."""
    x = np.asarray(x, np.float64)
    n, h, w, c = x.shape
    sy, sx = np.float32(h) / np.float32(oh), np.float32(w) / np.float32(ow)
    y = np.zeros((n, oh, ow, c))
    for i in range(oh):
        fy = np.float32(i) * sy
        y0 = int(math.floor(fy)); y1 = min(y0 + 1, h - 1); ly = float(fy - np.float32(y0))
        for j in range(ow):
            fx = np.float32(j) * sx
            x0 = int(math.floor(fx)); x1 = min(x0 + 1, w - 1); lx = float(fx - np.float32(x0))
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
            y[:, i, j] = top + (bot - top) * ly
    return y


def keypoint_losses(logits, enriched_ch0, labels):
    """keypoints_model.py:31-79 with the TF ops above, loops only. logits [b,h,w,18]; enriched_ch0: {level: [b,h_l,w_l]}
    = channel 0 of p2..p5; labels: heatmaps [b,h,w,17], loss_masks / segmentation_masks [b,h,w], num_boxes [b].
    The masks are halved per level by resize_bilinear_tf (:73-74), NOT by slicing - the [::2, ::2] identity that
    oracle/network.py relies on is what the tests check against this."""
    hm = np.asarray(labels["heatmaps"], np.float64)
    b = hm.shape[0]
    seg = np.asarray(labels["segmentation_masks"], np.float64)[..., None]
    lm = np.asarray(labels["loss_masks"], np.float64)[..., None]
    logits = np.asarray(logits, np.float64)
    out = {}
    fl = focal_loss(hm, labels["num_boxes"], logits[..., :17])
    out["focal_loss"] = float((lm[..., 0] * fl).sum()) / b
    out["regression_loss"] = 1e-3 * l2_loss(lm * (logits[..., 17:18] - seg)) / b
    for level in range(2, 6):
        x = np.asarray(enriched_ch0[level], np.float64)[..., None]
        out[f"segmentation_loss_at_level_{level}"] = 1e-5 * l2_loss(lm * (x - seg)) / b
        h, w = seg.shape[1], seg.shape[2]
        seg = resize_bilinear_tf(seg, h // 2, w // 2)
        lm = resize_bilinear_tf(lm, h // 2, w // 2)
    out["total_loss"] = sum(out.values())
    return out
