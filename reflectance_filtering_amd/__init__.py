"""reflectance_filtering_amd -- MI355X-native reflectance-adaptive filtering.

Host-side mirror of the reference's hot path (tnestmeyer/reflectance-filtering):
``filter_reflectance`` / ``decompose_with_trained_CNN`` / ``image_utils`` keep the reference's
function names and CLI; the OpenCV-ximgproc and Caffe calls underneath are hand-written HIP
kernels for gfx950 reached through the C ABI in include/reflectance_filtering.h.
Importing the package does not touch the GPU; calling an operator without the built
extension or without a GPU raises (there is no CPU fallback).
"""
__version__ = "0.1.0"

from . import image_utils  # noqa: F401
from . import weights  # noqa: F401
from . import _ffi  # noqa: F401
from . import ops  # noqa: F401
from . import ximgproc  # noqa: F401
from . import filter_reflectance  # noqa: F401
from . import decompose_with_trained_CNN  # noqa: F401
from . import whdr  # noqa: F401
from .filter_reflectance import apply_filter, apply_filter_batch, read_filter_write  # noqa: F401
from .decompose_with_trained_CNN import (  # noqa: F401
    decompose_and_filter_batch, decompose_batch, decompose_image, get_reflectance_batch,
    get_reflectance_caffe)
