"""TF checkpoint -> .npz with the variable names unchanged. Run where TensorFlow is installed (it is not in this image):

    python tools/tf_checkpoint_to_npz.py pretrained/mobilenet_v1_1.0_224.ckpt mobilenet_v1.npz
    python tools/tf_checkpoint_to_npz.py models/run00 run00.npz        # a model_dir: the latest checkpoint

The result loads with multiposenet_amd.checkpoint.load_npz / warm_start (train_keypoints.py:55, create_pb.py:170-185).
"""
import sys

import numpy as np


def main(src, dst):
    import tensorflow as tf   # noqa: needs TensorFlow (1.x or 2.x)
    ckpt = tf.train.latest_checkpoint(src) or src
    reader = tf.train.load_checkpoint(ckpt)
    out = {}
    for name in sorted(reader.get_variable_to_shape_map()):
        out[name] = np.asarray(reader.get_tensor(name))
    np.savez(dst, **out)
    print(f"{ckpt}: {len(out)} variables -> {dst}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
