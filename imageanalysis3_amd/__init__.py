"""imageanalysis3_amd — MI355X-native per-FOV spot-calling hot path of ImageAnalysis3.

Package globals mirror the reference's (/root/reference/__init__.py:8-20) so callers that
read them (e.g. ``classes/preprocess.py``) find the same names.
"""
import numpy as np

_distance_zxy = [200, 108, 108]
_sigma_zxy = [1.35, 1.9, 1.9]
_image_size = [30, 2048, 2048]
_allowed_colors = ['750', '647', '561', '488', '405']
_corr_channels = ['750', '647', '561']
_num_buffer_frames = 0
_num_empty_frames = 0
_image_dtype = np.uint16
_correction_folder = ''

__version__ = "0.1.0"
