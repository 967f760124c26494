"""Training-step driver: forward + losses + backward + (all-reduce) + Adam, replayed from a hipGraph.

Equivalent of the `train_op` that keypoints_model.py:115-120 hands to tf.estimator: one call =
one optimizer step (batch-norm moving statistics included, as under UPDATE_OPS).
"""
import torch

from . import _lib, ops
from .parallel import GradientAllReducer


@_lib.device_guarded("step", "eval_step", "input_buffers")
class Trainer:
    def __init__(self, net, params, use_graph=True, distributed=False, bucket_bytes=16 << 20):
        self.net = net
        self.device = net.device
        self.lr0 = float(params["initial_learning_rate"])
        self.num_steps = int(params["num_steps"])
        self.weight_decay = float(params.get("weight_decay", 0.0))
        self.use_graph = use_graph
        self.reducer = GradientAllReducer(net.grad, bucket_bytes=bucket_bytes) if distributed else None
        self.measure_comm = False   # bench: record (end of backward, exchange complete) event pairs per step
        self.comm_events = []
        self._static = None
        self._losses = None       # the f32[8] loss tensor of the buffer set the train step runs on
        self._graph_fb = None
        self._graph_bb = None
        self._graph_bb2 = None
        self._graph_opt = None

    # ---- pieces
    def _fwd_bwd(self, part=None):
        """part None: everything; 0: forward + losses + head/FPN backward; 1: backward of the deep backbone blocks; 2: of the
        shallow blocks and the stem (+ weight decay)."""
        s = self._static
        if part in (None, 0):
            self.net.forward(s["images"], True)
            self.net.compute_losses(s["labels"])
            if self.weight_decay > 0.0:
                self.net.add_weight_decay_loss(self.weight_decay)
        self.net.backward(part)
        if part in (None, 2) and self.weight_decay > 0.0:
            self.net.add_weight_decay_gradients(self.weight_decay)

    def _opt(self):
        scale = self.reducer.grad_scale if self.reducer else 1.0
        self.net.optimizer_step(self.lr0, self.num_steps, grad_scale=scale)

    def input_buffers(self, features, labels):
        """The static device buffers a replayed step reads: (features, labels) dicts shaped like the arguments.
        A loader that writes its batches straight into them (and passes them to `step`) saves the device-to-device
        copy that `step` otherwise makes of every batch (100 MB of images + 38 MB of labels at bs32 @ 512x512)."""
        self._bind(features, labels)
        return {"images": self._static["images"]}, self._static["labels"]

    def _bind(self, features, labels):
        imgs = features["images"]
        s = self._static
        if s is None or s["images"].shape != imgs.shape or s["images"].dtype != imgs.dtype:
            self._static = {"images": imgs.clone(), "labels": {k: v.clone() for k, v in labels.items()}}
            self._graph_fb = self._graph_bb = self._graph_bb2 = self._graph_opt = None
            # the loss tensor of THIS shape's buffer set: a graph replay does not touch net._last, which an eval_step at
            # another batch size rebinds in between
            N, H, W, _ = imgs.shape
            self._losses = self.net._buffers(N, H, W)["losses"]
            return
        if s["images"].data_ptr() != imgs.data_ptr():
            s["images"].copy_(imgs)
        for k, v in labels.items():
            if s["labels"][k].data_ptr() != v.data_ptr():
                s["labels"][k].copy_(v)

    def _capture(self):
        # eager warm-up on a side stream (sets kernel attributes, fills caches), then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph_fb = torch.cuda.CUDAGraph()
        if self.reducer is None:
            with torch.cuda.graph(self._graph_fb):
                self._fwd_bwd()
                self._opt()
        else:
            # data parallel: four graphs around the three RCCL exchanges (head/FPN gradients travel while the backbone's backward
            # runs, the deep backbone blocks' - 2.9 M of its 3.2 M parameters - while the shallow blocks' backward runs: what stays
            # exposed is the exchange of ~1.3 MB; weight decay touches every gradient, so with it the exchange waits for the end)
            with torch.cuda.graph(self._graph_fb):
                self._fwd_bwd(0)
            self._graph_bb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_bb):
                self._fwd_bwd(1)
            self._graph_bb2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_bb2):
                self._fwd_bwd(2)
            self._graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_opt):
                self._opt()

    def step(self, features, labels):
        """One optimizer step on this rank's shard. Returns the device tensor f32[8] of losses (ops.LOSS_NAMES)."""
        self._bind(features, labels)
        if not self.use_graph:
            self._fwd_bwd()
            if self.reducer:
                self.reducer.all_reduce()
            self._opt()
        else:
            if self._graph_fb is None:
                # the warm-up pass inside _capture must not count as a step: snapshot & restore state
                snap = [t.clone() for t in (self.net.theta, self.net.adam_m, self.net.adam_v, self.net.moving,
                                            self.net.global_step)]
                self._capture()
                for dst, src in zip((self.net.theta, self.net.adam_m, self.net.adam_v, self.net.moving,
                                     self.net.global_step), snap):
                    dst.copy_(src)
                self.net.repack_weights()
            self._graph_fb.replay()
            if self.reducer is not None:
                split, deep = self.net.backbone_grad_end, self.net.backbone_deep_begin
                early = self.weight_decay == 0.0
                if early:
                    self.reducer.start(split, None)      # head + FPN gradients: overlapped with the backbone's backward
                self._graph_bb.replay()
                if early:
                    self.reducer.start(deep, split)      # deep backbone blocks: overlapped with the shallow blocks' backward
                self._graph_bb2.replay()
                if self.measure_comm:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self.reducer.start(0, deep if early else None)
                self.reducer.finish()
                if self.measure_comm:
                    e1.record()
                    self.comm_events.append((e0, e1))
                self._graph_opt.replay()
            self.net.mark_variables_changed()    # (a replay does not run the Python that does this in the eager path)
        return self._losses

    def _static_bufs(self):
        """The buffer set of the bound (static) input shape."""
        N, H, W, _ = self._static["images"].shape
        return self.net._buffers(N, H, W)

    def eval_step(self, features, labels):
        self.net.forward(features["images"], False)
        losses = self.net.compute_losses(labels, with_grad=False)
        if self.weight_decay > 0.0:
            self.net.add_weight_decay_loss(self.weight_decay)
        return losses
