"""Variables on disk in the reference's naming: `.npz` files whose keys are TensorFlow variable names.

The reference keeps its state in TF checkpoints: `train_keypoints.py:55` warm-starts `MobilenetV1/*` from slim's
`mobilenet_v1_1.0_224.ckpt`, `tf.estimator` saves/restores everything else (variables, the Adam slots
`<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power`, `global_step`), `create_pb.py:170-185` freezes the variables
into the graph. Reading a TF checkpoint needs TensorFlow, which is not in this image: `tools/tf_checkpoint_to_npz.py` is the
one-off converter to run where it is (`tf.train.load_checkpoint` -> `np.savez`, names unchanged); everything here works on
the resulting `.npz` - HWIO kernels, [3,3,C,1] depthwise kernels, f32 - for KeypointNet and PoseResidualNet alike.
"""
import os

import numpy as np
import torch

ADAM_SLOTS = ("Adam", "Adam_1")   # tf.train.AdamOptimizer: first / second moment slot names


def slot_name(var, slot):
    return f"{var}/{ADAM_SLOTS[slot]}"


def save_npz(path, net, with_optimizer=True, beta1=0.9, beta2=0.999):
    """Everything `tf.train.Saver` would write for `net` (a KeypointNet or PoseResidualNet): variables, moving statistics,
    and (with_optimizer) the Adam slots, beta powers and global_step."""
    out = dict(net.state_dict())
    if with_optimizer:
        step = int(net.global_step.item())
        views = net._train_arena if hasattr(net, "_train_arena") else net._arena
        for slot, flat in enumerate((net.adam_m, net.adam_v)):
            for k, v in views.views(flat).items():
                v = net.unpad(k, v) if hasattr(net, "unpad") else v        # (reference shapes: net.internal_shapes)
                out[slot_name(k, slot)] = v.detach().cpu().numpy().copy()
        out["global_step"] = np.int64(step)
        # tf.train.AdamOptimizer creates the powers as beta and multiplies them after every apply: after `step` applies the
        # checkpoint holds beta^(step+1) (consistent with csrc/optim.hip using t = global_step + 1)
        out["beta1_power"] = np.float32(beta1 ** (step + 1))
        out["beta2_power"] = np.float32(beta2 ** (step + 1))
    # a kill during the write must not leave a truncated file under the final name (it would be the newest checkpoint
    # of model_dir): write beside it under a name no checkpoint glob matches, then rename atomically
    d, base = os.path.split(os.path.abspath(path))
    tmp = os.path.join(d, ".partial-" + (base if base.endswith(".npz") else base + ".npz"))
    np.savez(tmp, **out)
    os.replace(tmp, path if path.endswith(".npz") else path + ".npz")
    return sorted(out)


def _read(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def load_npz(path, net, scopes=None, strict=True, with_optimizer=True):
    """Restore `net` from an `.npz`. scopes: only variables whose name starts with one of these prefixes (the reference's
    warm start is scopes=("MobilenetV1/",), train_keypoints.py:55); strict applies to the selected variables only.
    Returns the list of restored names."""
    values = _read(path)
    own = set(net.vars) | set(getattr(net, "stats", {}))
    sel = {k: v for k, v in values.items() if k in own and (scopes is None or any(k.startswith(s) for s in scopes))}
    if strict:
        want = [k for k in own if scopes is None or any(k.startswith(s) for s in scopes)]
        missing = [k for k in want if k not in sel]
        if missing:
            raise KeyError(f"{path}: missing variables {missing[:5]}{'...' if len(missing) > 5 else ''}")
    net.load_state_dict(sel, strict=False)
    restored = sorted(sel)
    if with_optimizer and "global_step" in values:
        views = net._train_arena if hasattr(net, "_train_arena") else net._arena
        for slot, flat in enumerate((net.adam_m, net.adam_v)):
            for k, dst in views.views(flat).items():
                name = slot_name(k, slot)
                if name in values and (scopes is None or any(k.startswith(s) for s in scopes)):
                    v = np.asarray(values[name], np.float32)
                    if hasattr(net, "unpad") and k in getattr(net, "_pads", {}):
                        dst.zero_()
                        dst = net.unpad(k, dst)
                    if tuple(v.shape) != tuple(dst.shape):
                        raise ValueError(f"{name}: shape {v.shape} != {tuple(dst.shape)}")
                    dst.copy_(torch.from_numpy(v))
                    restored.append(name)
        if scopes is None:
            step = int(values["global_step"])
            for name, beta in (("beta1_power", 0.9), ("beta2_power", 0.999)):
                if name in values and not np.isclose(float(values[name]), beta ** (step + 1), rtol=1e-3):
                    raise ValueError(f"{path}: {name} = {float(values[name])} does not belong to global_step {step} "
                                     f"(expected {beta ** (step + 1)})")
            net.global_step.fill_(step)
            restored.append("global_step")
    return restored


def warm_start(path, net, scopes=("MobilenetV1/",)):
    """train_keypoints.py:55 (`tf.estimator.WarmStartSettings(ckpt, 'MobilenetV1/*')`): backbone variables and moving
    statistics only; optimizer state and global_step untouched. Variables absent from the file (a classification
    checkpoint has no keypoint head) are skipped, as TF's warm start does for names outside the regex."""
    return load_npz(path, net, scopes=scopes, strict=False, with_optimizer=False)
