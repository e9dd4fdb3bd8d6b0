"""Constants of the SE3DS path (values of reference constants.py:18-29)."""

INVALID_SEM_VALUE = 0   # MP3D void class (constants.py:21)
INVALID_RGB_VALUE = -1  # negative to avoid collision with black pixels (constants.py:22)

PI = 3.1415926535897932384626433
HFOV = 90 * PI / 180
DEPTH_SCALE = 20.0

NUM_MP3D_CLASSES = 42
PANO_VIDEO_LENGTH = 8

CKPT_UNSEEN = 'data/se3ds_ckpt'
CKPT_RE10K = 'data/se3ds_re10k_ckpt'
