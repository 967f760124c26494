"""Constants of detector/constants.py:4-13 that the keypoint hot path reads."""
# all image sizes must be divisible by this value (detector/constants.py:4)
DIVISOR = 128
# The reference computes in 'channels_first' (detector/constants.py:7) and transposes at the API edge; this build keeps
# NHWC end to end (channels innermost = contiguous MFMA K dimension, 16-byte channel vectors), so no transposes exist.
DATA_FORMAT = 'channels_last'
# number of body landmarks that will be predicted (detector/constants.py:10)
NUM_KEYPOINTS = 17
# all heatmaps and masks are downsampled (detector/constants.py:13)
DOWNSAMPLE = 4
