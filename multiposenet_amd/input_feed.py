"""Host -> device feed for the training step: the place of `dataset.prefetch` + the estimator's feed in the reference
(detector/input_pipeline/keypoints_detector_pipeline.py:60-63 builds the tf.data pipeline that `train_keypoints.py:26-45`
hands to `tf.estimator`; TensorFlow copies batch i+1 to the device while step i runs).

`HostBatchFeeder` owns `depth` slots, each a PINNED host batch + a device staging batch. The loader fills the pinned
arrays of a slot in place (`slot_arrays`), `submit` starts the asynchronous copy on a copy stream, and `train_step`
runs one optimizer step on the oldest submitted slot: the main stream waits for that slot's copy only, the step's own
static buffers are refreshed device-to-device (Trainer._bind), and the slot is handed back once the step has read it.
At bs32 @ 512x512 a batch is 100.7 MB of f32 images (25.2 MB as uint8, which the stem kernel standardises on load)
+ 38 MB of labels; over PCIe Gen5 that is 2-3 ms per step, hidden behind the previous step's 9.5 ms.

PyTorch is used as what it is in this repo: allocator of pinned / device memory, streams and events. No arithmetic.
"""
import collections

import torch

LABEL_DTYPES = {"heatmaps": torch.float32, "loss_masks": torch.float32, "segmentation_masks": torch.float32,
                "num_boxes": torch.int32}


class HostBatchFeeder:
    def __init__(self, trainer, batch_size, height, width, depth=2, image_dtype=torch.float32, device=None):
        if height % 128 or width % 128:
            raise ValueError("image height and width must be multiples of 128 (detector/constants.py:4)")
        if image_dtype not in (torch.float32, torch.uint8):
            raise ValueError("images travel as float32 in [0,1] or as uint8")
        if depth < 2:
            raise ValueError("depth >= 2: one slot is being consumed while the next one is copied")
        self.trainer = trainer
        self.device = torch.device(device if device is not None else trainer.net.device)
        h, w = height // 4, width // 4
        shapes = {"images": ((batch_size, height, width, 3), image_dtype),
                  "heatmaps": ((batch_size, h, w, 17), LABEL_DTYPES["heatmaps"]),
                  "loss_masks": ((batch_size, h, w), LABEL_DTYPES["loss_masks"]),
                  "segmentation_masks": ((batch_size, h, w), LABEL_DTYPES["segmentation_masks"]),
                  "num_boxes": ((batch_size,), LABEL_DTYPES["num_boxes"])}
        self._host = [{k: torch.empty(s, dtype=d).pin_memory() for k, (s, d) in shapes.items()} for _ in range(depth)]
        self._dev = [{k: torch.empty(s, dtype=d, device=self.device) for k, (s, d) in shapes.items()} for _ in range(depth)]
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._copied = [torch.cuda.Event() for _ in range(depth)]     # slot's H2D copy complete
        self._consumed = [None] * depth                                # slot's device batch read by its step
        self._free = collections.deque(range(depth))
        self._ready = collections.deque()
        self.bytes_per_batch = sum(t.numel() * t.element_size() for t in self._host[0].values())

    # ---- loader side
    def acquire(self):
        """Index of a free slot (raises if every slot is filled or in flight: call train_step first)."""
        if not self._free:
            raise RuntimeError("HostBatchFeeder: no free slot; run train_step() before acquiring another")
        slot = self._free.popleft()
        if self._consumed[slot] is not None:
            # the previous occupant's copy left the pinned arrays long ago (train_step waited for it on the device);
            # the host may only overwrite them once that copy has completed
            self._copied[slot].synchronize()
        return slot

    def slot_arrays(self, slot):
        """The slot's pinned host batch as numpy views {'images', 'heatmaps', 'loss_masks', 'segmentation_masks',
        'num_boxes'} - fill them in place."""
        return {k: t.numpy() for k, t in self._host[slot].items()}

    def submit(self, slot):
        """Start the slot's host -> device copy on the copy stream (returns immediately)."""
        cs = self._copy_stream
        if self._consumed[slot] is not None:
            cs.wait_event(self._consumed[slot])       # the step that read this slot's device batch has done so
        with torch.cuda.stream(cs):
            for k, src in self._host[slot].items():
                self._dev[slot][k].copy_(src, non_blocking=True)
            self._copied[slot].record(cs)
        self._ready.append(slot)

    def feed(self, features, labels):
        """Convenience for batches that live in ordinary (pageable) host memory: one host-side copy into a pinned slot,
        then submit. A loader that writes into slot_arrays() directly saves that copy."""
        slot = self.acquire()
        dst = self._host[slot]
        src = dict(labels)
        src["images"] = features["images"]
        for k, t in dst.items():
            v = src[k]
            v = v if torch.is_tensor(v) else torch.from_numpy(v)
            t.copy_(v)       # casts to the slot's dtype where the loader's differs (e.g. int64 num_boxes)
        self.submit(slot)
        return slot

    # ---- training side
    def pending(self):
        return len(self._ready)

    def train_step(self):
        """One optimizer step on the oldest submitted batch. Returns the device tensor f32[8] of losses."""
        if not self._ready:
            raise RuntimeError("HostBatchFeeder: no submitted batch")
        slot = self._ready.popleft()
        main = torch.cuda.current_stream(self.device)
        main.wait_event(self._copied[slot])
        d = self._dev[slot]
        labels = {k: v for k, v in d.items() if k != "images"}
        losses = self.trainer.step({"images": d["images"]}, labels)
        ev = torch.cuda.Event()
        ev.record(main)
        self._consumed[slot] = ev
        self._free.append(slot)
        return losses
