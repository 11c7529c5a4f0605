"""Numeric constants and dictionary keys shared by the host layer.

Values follow the reference so that checkpoints and RenderOut.extra dictionaries are
interchangeable (reference: thre3d_atom/utils/constants.py:1-27 and
thre3d_atom/thre3d_reprs/constants.py:1-11).
"""

NUM_COORD_DIMENSIONS = 3
NUM_COLOUR_CHANNELS = 3

SEED = 42
ZERO_PLUS = 1e-10
INFINITY = 1e10

# RenderOut.extra keys
EXTRA_DISPARITY = "disparity"
EXTRA_ACCUMULATED_WEIGHTS = "accumulated_weight"
# ... the per-sample debug outputs of the accumulator (reference utils/constants.py:14-18, accumulate.py:96-107)
EXTRA_POINT_DENSITIES = "point_densities"
EXTRA_POINT_OCCUPANCIES = "point_occupancies"
EXTRA_SAMPLE_INTERVALS = "deltas"
EXTRA_POINT_WEIGHTS = "point_weights"
EXTRA_POINT_DEPTHS = "point_depths"

# checkpoint dictionary keys
THRE3D_REPR = "thre3d_repr"
RENDER_PROCEDURE = "render_procedure"
RENDER_CONFIG = "render_config"
RENDER_CONFIG_TYPE = "render_config_type"
STATE_DICT = "state_dict"
CONFIG_DICT = "config_dict"
EXTRA_INFO = "extra_info"
u_DENSITIES = "_densities"
u_FEATURES = "_features"
CAMERA_BOUNDS = "camera_bounds"
CAMERA_INTRINSICS = "camera_intrinsics"
HEMISPHERICAL_RADIUS = "hemispherical_radius"
