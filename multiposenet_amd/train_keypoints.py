"""Runnable replacement of the reference's `train_keypoints.py` (:7-61): PARAMS, warm start of `MobilenetV1/*`, the
tf.estimator train loop with its cadence (checkpoint every `save_checkpoints_secs`, summaries every `save_summary_steps`,
steps/sec log every `log_step_count_steps`, evaluation every `throttle_secs`), resume from the newest checkpoint in
`model_dir` - on the HIP kernels, one process per GPU.

    python -m multiposenet_amd.train_keypoints [--synthetic] [--steps N] [--model-dir DIR] [--batch B] [--size S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m multiposenet_amd.train_keypoints --synthetic

Data: the reference reads TFRecords through tf.data (`KeypointPipeline`, out of scope here - SURVEY.md section 2); `train()`
takes any iterator of (features, labels) dicts in the pipeline's output contract (keypoints_detector_pipeline.py:104-110),
`--synthetic` feeds the on-device generator of multiposenet_amd.synthetic. Checkpoints are `.npz` files keyed by the TF
variable names (multiposenet_amd.checkpoint), `model.ckpt-<step>.npz`, like the estimator's `model.ckpt-<step>`.
"""
import argparse
import glob
import json
import os
import re
import time

import numpy as np
import torch

from . import checkpoint
from .keypoints_model import ModeKeys, get_trainer, model_fn

PARAMS = {   # train_keypoints.py:7-23
    'model_dir': 'models/run00/',
    'train_dataset': '/home/dan/datasets/COCO/multiposenet/train/',
    'val_dataset': '/home/dan/datasets/COCO/multiposenet/val/',
    'pretrained_checkpoint': 'pretrained/mobilenet_v1_1.0_224.npz',   # the slim checkpoint exported by tools/tf_checkpoint_to_npz.py

    'backbone': 'mobilenet',
    'depth_multiplier': 1.0,
    'weight_decay': 0.0,

    'num_steps': 200000,
    'initial_learning_rate': 3e-4,

    'min_dimension': 512,
    'batch_size': 16,
    'image_size': (512, 512),
}
RUN_CONFIG = {'save_summary_steps': 200, 'save_checkpoints_secs': 7200, 'log_step_count_steps': 1000,   # train_keypoints.py:46-50
              'eval_start_delay_secs': 7200, 'eval_throttle_secs': 7200}                                 # :60


def _readable(path):
    try:
        with np.load(path) as z:
            return "global_step" in z.files or len(z.files) > 0
    except Exception:
        return False


def latest_checkpoint(model_dir):
    """(step, path) of the newest checkpoint numpy can open; files a killed writer left truncated are skipped."""
    found = []
    for f in glob.glob(os.path.join(model_dir, "model.ckpt-*.npz")):
        m = re.search(r"model\.ckpt-(\d+)\.npz$", f)
        if m:
            found.append((int(m.group(1)), f))
    for step, f in sorted(found, reverse=True):
        if _readable(f):
            return (step, f)
    return None


def train(params, train_batches, val_batches=None, run_config=None, max_steps=None, log=print):
    """tf.estimator.train_and_evaluate for the keypoint model. train_batches / val_batches: iterators (or callables returning
    iterators) of (features, labels). Returns the global step reached."""
    cfg = dict(RUN_CONFIG, **(run_config or {}))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    params = dict(params)
    if world > 1:
        params["distributed"] = True
    trainer = get_trainer(params)            # (joins the process group and binds this rank's device when distributed)
    net = trainer.net
    model_dir = params["model_dir"]
    if rank == 0:
        os.makedirs(model_dir, exist_ok=True)
    last = latest_checkpoint(model_dir)
    if last is not None:                     # the estimator resumes from model_dir before it looks at warm_start_from
        checkpoint.load_npz(last[1], net)
        log(f"[train] restored {last[1]} (global_step {int(net.global_step.item())})")
    elif params.get("pretrained_checkpoint") and os.path.exists(params["pretrained_checkpoint"]):
        names = checkpoint.warm_start(params["pretrained_checkpoint"], net, scopes=("MobilenetV1/",))   # train_keypoints.py:55
        log(f"[train] warm start: {len(names)} variables from {params['pretrained_checkpoint']}")
    max_steps = int(max_steps if max_steps is not None else params["num_steps"])
    it = iter(train_batches() if callable(train_batches) else train_batches)
    step = int(net.global_step.item())
    t_ckpt = t_log = t_start = time.time()
    t_eval = t_start + cfg["eval_start_delay_secs"] - cfg["eval_throttle_secs"]
    step_log = step
    summaries = os.path.join(model_dir, "summaries.jsonl")

    def save():
        if rank == 0:     # replicas are identical; moving statistics are per replica (rank 0's are kept)
            path = os.path.join(model_dir, f"model.ckpt-{step}.npz")
            checkpoint.save_npz(path, net)
            log(f"[train] saved {path}")
    while step < max_steps:
        features, labels = next(it)
        spec = model_fn(features, labels, ModeKeys.TRAIN, params)
        step += 1
        if step % cfg["save_summary_steps"] == 0 and rank == 0:
            rec = {"step": step, **{k: float(v) for k, v in spec.losses.items()}}
            with open(summaries, "a") as f:
                f.write(json.dumps(rec) + "\n")
        if step % cfg["log_step_count_steps"] == 0:
            torch.cuda.synchronize()
            now = time.time()
            log(f"[train] step {step}: loss {float(spec.loss):.4f}, {(step - step_log) / (now - t_log):.2f} steps/s")
            t_log, step_log = now, step
        now = time.time()
        if now - t_ckpt >= cfg["save_checkpoints_secs"]:
            save()
            t_ckpt = now
        if val_batches is not None and now - t_eval >= cfg["eval_throttle_secs"]:
            evaluate(params, val_batches, log=log, step=step)
            t_eval = now
    save()
    return step


def evaluate(params, val_batches, log=print, step=None):
    """EvalSpec(steps=None): one pass over the validation batches; means of the eval metrics of keypoints_model.py:92-105."""
    sums, n = {}, 0
    for features, labels in (val_batches() if callable(val_batches) else val_batches):
        spec = model_fn(features, labels, ModeKeys.EVAL, params)
        for k, v in spec.eval_metric_ops.items():
            sums[k] = sums.get(k, 0.0) + float(v)
        n += 1
    out = {k: v / max(n, 1) for k, v in sums.items()}
    log(f"[eval] step {step}: " + ", ".join(f"{k} {v:.5f}" for k, v in sorted(out.items())))
    return out


def synthetic_batches(batch_size, height, width, device=None, distinct=8):
    """An endless stream of `distinct` device-resident synthetic batches (seeded per rank) in the pipeline's contract."""
    from .synthetic import synthetic_batch
    rank = int(os.environ.get("RANK", "0"))
    dev = device or f"cuda:{torch.cuda.current_device()}"
    pool = [synthetic_batch(batch_size, height, width, rank=rank * 1000 + i, device=dev) for i in range(distinct)]
    i = 0
    while True:
        yield pool[i % distinct]
        i += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", action="store_true", help="train on generated batches (no TFRecord reader in this build)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()
    params = dict(PARAMS, dtype=args.dtype)
    if args.model_dir:
        params["model_dir"] = args.model_dir
    if args.batch:
        params["batch_size"] = args.batch
    if args.size:
        params["image_size"] = (args.size, args.size)
    if not args.synthetic:
        raise SystemExit("this build has no TFRecord reader (SURVEY.md section 2: input pipelines are out of scope): pass --synthetic, "
                         "or call multiposenet_amd.train_keypoints.train(PARAMS, your_batch_iterator)")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from .parallel import init_distributed
        _, local_rank, _ = init_distributed()
        torch.cuda.set_device(local_rank)
    h, w = params["image_size"]
    train(params, lambda: synthetic_batches(params["batch_size"], h, w), max_steps=args.steps)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
