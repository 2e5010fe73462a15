"""Drop-in module name: ``import image_utils as iu`` keeps working."""
from reflectance_filtering_amd.image_utils import *  # noqa: F401,F403
from reflectance_filtering_amd.image_utils import (  # noqa: F401
    colorize, imread, imwrite, normalize, rgb_to_srgb, srgb_byte_lut, srgb_to_rgb)
