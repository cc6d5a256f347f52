"""spot_tools — seeding / fitting operators (reference: spot_tools/__init__.py, fitting.py, matching.py)."""
from .. import _sigma_zxy, _image_size, _allowed_colors, _corr_channels, _distance_zxy  # noqa: F401

# default seeding thresholds per channel (reference: classes/batch_functions.py:10-17)
_seed_th = {'750': 400, '647': 600, '561': 400}
