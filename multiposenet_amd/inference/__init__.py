from .utils import get_keypoints, get_keypoints_batch, KeypointDecoder  # noqa: F401
from .detector import Detector  # noqa: F401
