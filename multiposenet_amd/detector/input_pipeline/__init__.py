"""Label producers of the keypoint path (reference detector/input_pipeline/): only target-heatmap rendering is built."""
from .heatmap_creation import get_heatmaps, get_heatmaps_batch, HeatmapRenderer  # noqa: F401
