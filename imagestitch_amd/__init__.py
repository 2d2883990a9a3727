"""imagestitch_amd — MI355X-native warp + multi-band blend hot path of mhhai/ImageStitch.

The package is a thin host-side mirror of the reference's call surface
(cv::detail::RotationWarper / cv::detail::Blender) over the C-ABI HIP library
imagestitch_amd/csrc/libimagestitch_hip.so (include/imagestitch_hip.h).  No CPU fallback exists.
"""
from ._lib import (BORDER_CONSTANT, BORDER_REFLECT, BORDER_REFLECT_101, BORDER_REPLICATE, INTER_LINEAR,  # noqa: F401
                   INTER_NEAREST, INTER_TIES_EVEN, PREC_F16ACC32, PREC_F32, PREC_I16, IsxError, load)
from .blender import Blender, FeatherBlender, MultiBandBlender, NoBlender, convert_to, dilate_and, gain_apply  # noqa: F401
from .imgio import imread, imwrite  # noqa: F401
from .seam import DpSeamFinder, seam_estimate  # noqa: F401
from .warper import CylindricalWarper, RotationWarper, SphericalWarper, remap  # noqa: F401

__all__ = ["Blender", "MultiBandBlender", "FeatherBlender", "NoBlender", "convert_to", "dilate_and", "gain_apply", "imread", "imwrite", "seam_estimate", "DpSeamFinder", "remap", "CylindricalWarper", "SphericalWarper", "RotationWarper", "IsxError", "load"]
