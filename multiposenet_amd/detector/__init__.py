"""Network builders with the reference's names (detector/__init__.py:1-3 exports KeypointSubnet, RetinaNet, prn)."""
from .keypoint_subnet import KeypointSubnet  # noqa: F401
from .fpn import feature_pyramid_network  # noqa: F401
from .feature_map import FeatureMap  # noqa: F401
from .retinanet import RetinaNet  # noqa: F401
from .anchor_generator import AnchorGenerator  # noqa: F401
