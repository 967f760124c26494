from .layer_utils import conv2d_same, batch_norm_relu  # noqa: F401
