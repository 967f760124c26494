"""Pose residual network on the HIP kernels - `prn` (reference detector/prn.py:5-25) and the TRAIN / EVAL step of
prn_model.py:5-57 (softmax-over-space log loss, cosine LR, TF-Adam without clipping).

GEMM mapping (b = number of crops, n = h*w*c = 34272, hidden = 1024):
    fc1 forward   pre1[b,1024] = X[b,n] W1[n,1024]        K = n: split-K "weight gradient" of a 1x1 conv whose pixel
    fc2 dgrad     dH[b,1024]   = dPre2[b,n] W2^T[n,1024]   axis is K (operands K-major: X^T / dPre2^T, W1 / W2^T)
    fc2 forward   pre2[b,n]    = H[b,1024] W2[1024,n]      1x1 conv, 1024 -> n channels over b "pixels"
    fc1 / fc2 wgrad                                        1x1 conv weight gradients over b "pixels" (one slab each)
"""
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, ops
from ._lib import call, ptr, stream_ptr
from .net import _Arena

NUM_KEYPOINTS = 17
CROP_SIZE = (56, 36)     # detector/constants.py
HIDDEN = 1024


def variable_shapes(h=CROP_SIZE[0], w=CROP_SIZE[1], c=NUM_KEYPOINTS, hidden=HIDDEN):
    n = h * w * c
    return OrderedDict([("PRN/fc1/weights", (n, hidden)), ("PRN/fc1/biases", (hidden,)),
                        ("PRN/fc2/weights", (hidden, n)), ("PRN/fc2/biases", (n,))])


def initial_values(seed=0, **kw):
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for k, shp in variable_shapes(**kw).items():
        if k.endswith("weights"):       # tf.variance_scaling_initializer(): truncated normal, fan_in
            std = np.sqrt(1.0 / shp[0]) / 0.87962566103423978
            out[k] = (np.clip(rs.randn(*shp), -2, 2) * std).astype(np.float32)
        else:
            out[k] = np.zeros(shp, np.float32)
    return out


@_lib.device_guarded("_init", "load_state_dict", "refresh_operands", "forward", "loss", "predict", "backward",
                     "optimizer_step", "train_step")
class PoseResidualNet:
    def __init__(self, values=None, batch=128, h=CROP_SIZE[0], w=CROP_SIZE[1], c=NUM_KEYPOINTS, hidden=HIDDEN,
                 dtype=torch.bfloat16, device="cuda:0", seed=0, share_variables_of=None):
        """share_variables_of: another PoseResidualNet of the same geometry and dtype - this instance then runs at ITS OWN
        batch size on the OTHER's variables, Adam slots, step counter and operand copies (one model, several batch
        sizes: a partial last batch, EVAL at batch 1 after TRAIN at batch 128)."""
        self.device = torch.device(device) if share_variables_of is None else share_variables_of.device
        self._init(values, batch, h, w, c, hidden, dtype, seed, share_variables_of)

    def for_batch(self, batch):
        """The instance that runs this model at `batch` crops (activation buffers are per batch size, variables shared)."""
        if int(batch) == self.valid:
            return self
        sib = self._siblings.get(int(batch))
        if sib is None:
            sib = PoseResidualNet(batch=int(batch), h=self.h, w=self.w, c=self.c, hidden=self.hidden, dtype=self.dtype,
                                  share_variables_of=self)
        return sib

    def _init(self, values, batch, h, w, c, hidden, dtype, seed, share=None):
        _lib.lib()   # fail loudly without the HIP library
        # the crop axis is a GEMM dimension (a multiple of 8 elements): `valid` crops run in buffers of B = valid rounded up,
        # the padding rows stay zero (inputs, loss gradient) and never enter the loss
        self.valid = int(batch)
        self.B, self.h, self.w, self.c, self.hidden = (int(batch) + 7) // 8 * 8, h, w, c, hidden
        self.n = h * w * c
        if self.n % 8 or hidden % 8:
            raise ValueError("h*w*c and hidden must be multiples of 8")
        self.dtype = dtype
        # static loss scale of the fp16 build: the gradient of a mean over batch*h*w*c (~4.4e6) terms sits below fp16's
        # normal range once it is stored as a GEMM operand (dpre2, dpre1), so the backward pass carries
        # loss_scale * gradient (a power of two: exact) and Adam divides it out. bf16 / f32 have the exponent range.
        self.loss_scale = float(2 ** int(np.floor(np.log2(batch * h * w * c)))) if dtype == torch.float16 else 1.0
        dev, B, n = self.device, self.B, self.n
        if share is None:
            shapes = variable_shapes(h, w, c, hidden)
            self._arena = _Arena(shapes, self.device)
            self.theta, self.grad = self._arena.new(), self._arena.new()
            self.adam_m, self.adam_v = self._arena.new(), self._arena.new()
            self.vars, self.grads = self._arena.views(self.theta), self._arena.views(self.grad)
            self.global_step = torch.zeros(1, dtype=torch.int64, device=self.device)
            self.hyper = torch.zeros(4, dtype=torch.float32, device=self.device)
            self.load_state_dict(values if values is not None else initial_values(seed, h=h, w=w, c=c, hidden=hidden))
            W1, W2 = self.vars["PRN/fc1/weights"], self.vars["PRN/fc2/weights"]
            # operand copies in the storage dtype: W1 [n,1024] and W2 [1024,n] as stored (written by the Adam kernel itself,
            # mpn_adam_step_cast), W2^T [n,1024] (one transposing pass per step)
            self.w1_op = W1 if dtype == torch.float32 else torch.empty((n, hidden), dtype=dtype, device=dev)
            self.w2_op = W2 if dtype == torch.float32 else torch.empty((hidden, n), dtype=dtype, device=dev)
            # W2^T [n,1024]: only the f32 build's data gradient of fc2 still contracts over rows of a transposed copy; the
            # 16-bit builds read W2 as stored (mpn_gemm_nt)
            self.w2t_op = torch.empty((n, hidden), dtype=dtype, device=dev) if dtype == torch.float32 else None
            self._adam_cast = None
            if dtype != torch.float32:
                o1, n1, _ = self._arena.offsets["PRN/fc1/weights"]
                o2, n2, _ = self._arena.offsets["PRN/fc2/weights"]
                self._adam_cast = ops.AdamCastJobs([(o1, n1, self.w1_op), (o2, n2, self.w2_op)])
            self._siblings = {self.valid: self}
        else:
            if (share.h, share.w, share.c, share.hidden, share.dtype) != (h, w, c, hidden, dtype):
                raise ValueError("share_variables_of: geometry / dtype differ")
            for a in ("_arena", "theta", "grad", "adam_m", "adam_v", "vars", "grads", "global_step", "hyper", "w1_op",
                      "w2_op", "w2t_op", "_adam_cast", "_siblings"):
                setattr(self, a, getattr(share, a))
            self._siblings[self.valid] = self
            # the fp16 loss scale is a function of the batch size; Adam divides by the scale of the instance that steps
        f32 = torch.float32
        self.xt = torch.empty((n, B), dtype=dtype, device=dev)            # X^T
        self.x_op = torch.empty((B, n), dtype=dtype, device=dev)          # X in the storage dtype (fc1 wgrad operand)
        self.pre1 = torch.empty((B, hidden), dtype=f32, device=dev)
        self.hid = torch.empty((B, hidden), dtype=dtype, device=dev)
        self.hidt = torch.empty((hidden, B), dtype=dtype, device=dev)     # H^T: fc2 as a contraction over its 1024 rows
        self.pre2 = torch.empty((B, n), dtype=f32, device=dev)
        self.y2 = torch.empty((B, n), dtype=dtype, device=dev)
        self.logits = torch.empty((B, n), dtype=f32, device=dev)
        self.dlogits = torch.zeros((B, n), dtype=f32, device=dev)     # (rows >= valid are never written: stay zero)
        self._xpad = torch.zeros((B, h, w, c), dtype=f32, device=dev) if self.valid != B else None
        self.dpre2 = torch.empty((B, n), dtype=dtype, device=dev)
        self.dpre2t = torch.empty((n, B), dtype=dtype, device=dev) if dtype == torch.float32 else None
        self.nt_slab = None if dtype == torch.float32 else torch.empty(ops.gemm_nt_num_parts(n) * B * hidden, dtype=f32, device=dev)
        self.dhid = torch.empty((B, hidden), dtype=f32, device=dev)
        self.dpre1 = torch.empty((B, hidden), dtype=dtype, device=dev)
        self.loss_part = torch.zeros(B, dtype=f32, device=dev)
        self._loss = torch.zeros(1, dtype=f32, device=dev)
        nparts = ops.conv_wgrad_num_parts(1, 1, n, B, hidden, 1, dtype)   # the two K = n contractions
        self.kslab = torch.empty(nparts * B * hidden, dtype=f32, device=dev)
        self._kparts = nparts
        for name, cin, cout in (("PRN/fc1/weights", n, hidden), ("PRN/fc2/weights", hidden, n)):
            if ops.conv_wgrad_num_parts(1, 1, B, cin, cout, 1, dtype) != 1:
                raise RuntimeError("weight-gradient geometry changed: expected one slab for " + name)
        # fc2 forward: K = 1024 rows, output [B, n] - one slab (the output itself) in the 16-bit builds; the f32 kernel splits K
        self._fc2_parts = ops.conv_wgrad_num_parts(1, 1, hidden, B, n, 1, dtype)
        self.fc2_slab = torch.empty(self._fc2_parts * B * n, dtype=f32, device=dev) if self._fc2_parts != 1 else None
        if share is None:
            self.refresh_operands()

    # ---------------------------------------------------------------- state
    def state_dict(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.vars.items())

    def load_state_dict(self, values, strict=True):
        for k, v in self.vars.items():
            if k not in values:
                if strict:
                    raise KeyError(f"missing variable {k}")
                continue
            a = np.asarray(values[k], np.float32)
            if a.shape != tuple(v.shape):
                raise ValueError(f"{k}: shape {a.shape} != {tuple(v.shape)}")
            v.copy_(torch.from_numpy(a))
        if hasattr(self, "w2t_op"):
            self.refresh_operands()

    def refresh_operands(self, casts=True):
        """Operand copies of the f32 masters. casts=False (after an optimizer step): the Adam kernel has already written the
        two plain casts (mpn_adam_step_cast) - in the 16-bit builds nothing remains to refresh."""
        dc, f32c = _lib.dtype_code(self.dtype), _lib.dtype_code(torch.float32)
        W1, W2 = self.vars["PRN/fc1/weights"], self.vars["PRN/fc2/weights"]
        if casts and self.dtype != torch.float32:
            call("mpn_cast", ptr(W1), f32c, ptr(self.w1_op), dc, W1.numel(), stream_ptr())
            call("mpn_cast", ptr(W2), f32c, ptr(self.w2_op), dc, W2.numel(), stream_ptr())
        if self.w2t_op is not None:
            call("mpn_transpose_cast", ptr(W2), f32c, ptr(self.w2t_op), dc, self.hidden, self.n, stream_ptr())

    # ---------------------------------------------------------------- forward / loss / backward
    def _kgemm(self, at, bmat, out):
        """out[B? rows = at.shape[1], cols = bmat.shape[1]] = at^T @ bmat with the contraction over the n rows."""
        n, rows = at.shape
        cols = bmat.shape[1]
        ops.conv_bwd_weight(at.view(1, 1, n, rows), bmat.view(1, 1, n, cols), 1, None, out.view(1, 1, rows, cols), self.kslab)

    def forward(self, x):
        """x: f32 [b,h,w,c] device tensor (b == batch). Returns y2 = relu(fc2(relu(fc1(x)))) [b, n] (storage dtype);
        logits = x + y2 are formed by `loss` / `predict`."""
        B, n, dc = self.B, self.n, _lib.dtype_code(self.dtype)
        if tuple(x.shape) != (self.valid, self.h, self.w, self.c) or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError(f"x must be contiguous float32 [{self.valid},{self.h},{self.w},{self.c}]")
        if self._xpad is not None:
            self._xpad[:self.valid].copy_(x)
            x = self._xpad
        f32c = _lib.dtype_code(torch.float32)
        call("mpn_transpose_cast", ptr(x), f32c, ptr(self.xt), dc, B, n, stream_ptr())
        call("mpn_cast", ptr(x), f32c, ptr(self.x_op), dc, B * n, stream_ptr())
        self._kgemm(self.xt, self.w1_op, self.pre1)
        call("mpn_bias_relu_fwd", ptr(self.pre1), f32c, ptr(self.vars["PRN/fc1/biases"]), ptr(self.hid), dc, B, self.hidden, stream_ptr())
        # fc2 as the same kind of contraction over rows (here the 1024 hidden units): W2 is read as stored - its operand copy is
        # a plain cast, which the Adam kernel writes itself; the packed image of a 1x1 convolution was a 145 us pass per step
        call("mpn_transpose_cast", ptr(self.hid), dc, ptr(self.hidt), dc, B, self.hidden, stream_ptr())
        if self.fc2_slab is None:
            ops.conv_bwd_weight(self.hidt.view(1, 1, self.hidden, B), self.w2_op.view(1, 1, self.hidden, n), 1, None,
                                self.pre2.view(1, 1, B, n), self.pre2.view(-1), reduce=False)
        else:
            ops.conv_bwd_weight(self.hidt.view(1, 1, self.hidden, B), self.w2_op.view(1, 1, self.hidden, n), 1, None,
                                self.pre2.view(1, 1, B, n), self.fc2_slab)
        call("mpn_bias_relu_fwd", ptr(self.pre2), f32c, ptr(self.vars["PRN/fc2/biases"]), ptr(self.y2), dc, B, n, stream_ptr())
        self._x = x
        return self.y2

    def loss(self, labels, with_grad=True):
        """labels f32 [b,h,w,c]. Returns the device scalar loss (prn_model.py:29); fills logits (and dlogits)."""
        B = self.valid      # one block per crop; the mean runs over the valid crops only
        if tuple(labels.shape) != (B, self.h, self.w, self.c) or labels.dtype != torch.float32 or not labels.is_contiguous():
            raise ValueError(f"labels must be contiguous float32 [{B},{self.h},{self.w},{self.c}]")
        call("mpn_prn_loss", ptr(self._x), ptr(self.y2), _lib.dtype_code(self.dtype), ptr(labels), B, self.h * self.w, self.c,
             ptr(self.logits), ptr(self.dlogits) if with_grad else None, ptr(self.loss_part), self.loss_scale, stream_ptr())
        ops.reduce_partials(self.loss_part, B, 1, self._loss)      # fixed-order sum of the per-crop terms
        return self._loss[0]

    def predict(self, x):
        """Inference: logits [b,h,w,c] f32 (create_pb.py:112)."""
        self.forward(x)
        v = self.valid
        out = torch.empty((v, self.h, self.w, self.c), dtype=torch.float32, device=self.device)
        call("mpn_prn_residual", ptr(self._x), ptr(self.y2), _lib.dtype_code(self.dtype), v * self.n, ptr(out), stream_ptr())
        return out

    def backward(self):
        B, n, hidden, dc = self.B, self.n, self.hidden, _lib.dtype_code(self.dtype)
        g = self.grads
        call("mpn_bias_relu_bwd", ptr(self.y2), dc, ptr(self.dlogits), ptr(self.dpre2), dc, ptr(g["PRN/fc2/biases"]), B, n, stream_ptr())
        # fc2 weight gradient: dW2[1024,n] = H^T dPre2 (one slab = the gradient itself)
        ops.conv_bwd_weight(self.hid.view(1, 1, B, hidden), self.dpre2.view(1, 1, B, n), 1, None,
                            g["PRN/fc2/weights"].view(1, 1, hidden, n), g["PRN/fc2/weights"].view(-1), reduce=False)
        # fc2 data gradient: dH = dPre2 W2^T (K = n). 16-bit builds: both operands as stored, K contiguous (mpn_gemm_nt) - the
        # transposed copy of W2 (a 69-90 us pass per step) and of dPre2 are gone
        if self.nt_slab is not None:
            ops.gemm_nt(self.dpre2, self.w2_op, self.dhid, self.nt_slab)
        else:
            call("mpn_transpose_cast", ptr(self.dpre2), dc, ptr(self.dpre2t), dc, B, n, stream_ptr())
            self._kgemm(self.dpre2t, self.w2t_op, self.dhid)
        call("mpn_bias_relu_bwd", ptr(self.hid), dc, ptr(self.dhid), ptr(self.dpre1), dc, ptr(g["PRN/fc1/biases"]), B, hidden, stream_ptr())
        # fc1 weight gradient: dW1[n,1024] = X^T dPre1
        ops.conv_bwd_weight(self.x_op.view(1, 1, B, n), self.dpre1.view(1, 1, B, hidden), 1, None,
                            g["PRN/fc1/weights"].view(1, 1, n, hidden), g["PRN/fc1/weights"].view(-1), reduce=False)

    def optimizer_step(self, initial_learning_rate, num_steps):
        ops.adam_prepare(self.global_step, self.hyper, initial_learning_rate, num_steps)
        if self._adam_cast is not None:
            ops.adam_step_cast(self.theta, self.grad, self.adam_m, self.adam_v, self.hyper, self._adam_cast,
                               grad_scale=1.0 / self.loss_scale, clip=float("inf"))
        else:
            ops.adam_step(self.theta, self.grad, self.adam_m, self.adam_v, self.hyper, grad_scale=1.0 / self.loss_scale, clip=float("inf"))
        self.refresh_operands(casts=False)

    def train_step(self, x, labels, initial_learning_rate, num_steps):
        self.forward(x)
        loss = self.loss(labels)
        self.backward()
        self.optimizer_step(initial_learning_rate, num_steps)
        return loss
