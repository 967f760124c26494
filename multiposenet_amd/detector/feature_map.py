"""FeatureMap: what the network builders hand to each other instead of a tf.Tensor."""
from .. import ops


class FeatureMap:
    """A conv output as it lives in HBM: the RAW NHWC tensor plus the batch-norm affine + activation its consumers apply
    on load. `tensor()` / `numpy()` materialise the activated values (the tf.Tensor the reference would return)."""

    def __init__(self, raw, affine=None):
        self.raw, self.affine = raw, affine

    @property
    def shape(self):
        """Logical NCHW shape, as the reference's channels_first tensors report it."""
        n, h, w, c = self.raw.shape
        return (n, c, h, w)

    def tensor(self):
        """Activated values, NHWC torch tensor (storage dtype)."""
        if self.affine is None:
            return self.raw
        return ops.bn_act_apply(self.raw, self.affine)

    def numpy(self, layout="NCHW"):
        t = self.tensor().float().cpu()
        return (t.permute(0, 3, 1, 2) if layout == "NCHW" else t).numpy()
