from .mobilenet_v1 import mobilenet_v1  # noqa: F401
