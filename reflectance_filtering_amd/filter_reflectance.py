#!/usr/bin/env python
"""Reflectance filtering: joint-bilateral / guided filter of a reflectance estimate.

Same operator API and command line as the reference's ``filter_reflectance.py``
(/root/reference/filter_reflectance.py:49-139); the two OpenCV-ximgproc calls are served by
hand-written HIP kernels on the MI355X (``reflectance_filtering_amd.ximgproc``).

    python filter_reflectance.py --filter_type=bilateral --sigma_color=20 --sigma_spatial=22 \
        --filename_in=x-r.png --guidance_in=x.png --path_out=out/
"""
from __future__ import division, print_function

import argparse
import os
import sys

from . import image_utils as iu
from . import ops, ximgproc

FILTER_TYPES = ("bilateral", "guided")

PARAMETER_HINTS = (
    # CNN prediction filtered with itself as guidance
    "--filter_type=bilateral --sigma_color=20 --sigma_spatial=22",
    "--filter_type=guided --sigma_color=7 --sigma_spatial=52",
    # CNN prediction filtered with an L1-flattened image ('flat') as guidance
    "--filter_type=guided --sigma_color=3 --sigma_spatial=45",
)


def _check_params(filter_type, sigma_color, sigma_spatial):
    """Validation order of /root/reference/filter_reflectance.py:56-57,71-72: sigmas first."""
    if sigma_color <= 0 or sigma_spatial <= 0:
        raise ValueError("Parameters are expected to be positive.")
    if filter_type not in FILTER_TYPES:
        raise ValueError("filter_type must be 'bilateral' or 'guided'.")


def apply_filter(filter_type, image, joint, sigma_color, sigma_spatial):
    """Filter ``image`` (uint8 HxWx3) guided by ``joint``.

    'bilateral': jointBilateralFilter(joint, image, d=-1, sigmaColor=sigma_color,
                 sigmaSpace=sigma_spatial);
    'guided':    guidedFilter(guide=joint, src=image, radius=int(sigma_spatial), eps=sigma_color)
    -- the parameter mapping of /root/reference/filter_reflectance.py:58-70.
    """
    _check_params(filter_type, sigma_color, sigma_spatial)
    if filter_type == "bilateral":
        return ximgproc.jointBilateralFilter(joint, image, d=-1, sigmaColor=sigma_color,
                                             sigmaSpace=sigma_spatial)
    return ximgproc.guidedFilter(guide=joint, src=image, radius=int(sigma_spatial),
                                 eps=sigma_color)


def apply_filter_batch(filter_type, images, joints, sigma_color, sigma_spatial, iterations=1):
    """Device-resident batch form: ``images``/``joints`` are CUDA uint8 tensors [N,H,W,C];
    returns a CUDA uint8 tensor.  ``iterations`` > 1 chains the filter with the same guidance
    (the uint8 result of one pass is the input of the next), e.g. the reference's 3x GF."""
    _check_params(filter_type, sigma_color, sigma_spatial)
    if iterations < 1:
        raise ValueError("iterations must be >= 1")
    if filter_type == "guided":
        return ops.guided_filter_u8(joints, images, int(sigma_spatial), sigma_color,
                                    iterations=iterations)
    out = images
    for _ in range(iterations):
        out = ops.joint_bilateral_u8(joints, out, -1, sigma_color, sigma_spatial)
    return out


def output_filename(filename_in, path_out, filter_type, sigma_color, sigma_spatial):
    """<path_out>/<basename>_<type>_c<sigma_color>s<sigma_spatial>.png with the parameters
    formatted by str.format (floats keep their '.0'): /root/reference/filter_reflectance.py:81-93."""
    basename = os.path.splitext(os.path.basename(filename_in))[0]
    suffix = "_{}_c{}s{}".format(filter_type, sigma_color, sigma_spatial)
    return os.path.join(path_out, basename + suffix + ".png")


def read_filter_write(filter_type, filename_in, guidance_in, sigma_color, sigma_spatial,
                      path_out):
    """Read the image and its guidance, filter, write the PNG, return the filtered image."""
    image = iu.imread(filename_in)
    joint = iu.imread(guidance_in)
    filtered = apply_filter(filter_type, image, joint, sigma_color, sigma_spatial)
    iu.imwrite(output_filename(filename_in, path_out, filter_type, sigma_color, sigma_spatial),
               filtered)
    return filtered


def build_parser():
    parser = argparse.ArgumentParser(
        description="Filter a reflectance prediction with a joint bilateral or guided filter "
                    "to strengthen the piecewise-constant reflectance prior (MI355X build).")
    parser.add_argument("--filename_in", help="image to be filtered")
    parser.add_argument("--guidance_in", help="guidance (joint) image steering the filter")
    parser.add_argument("--path_out", help="existing folder that receives the result")
    parser.add_argument("--sigma_color", type=float, help="color parameter")
    parser.add_argument("--sigma_spatial", type=float, help="spatial parameter")
    parser.add_argument("--filter_type",
                        help="'guided' (guided filter) or 'bilateral' (joint bilateral filter)")
    return parser


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    parser = build_parser()
    args = parser.parse_args(argv)
    if len(argv) > 0:
        read_filter_write(args.filter_type, args.filename_in, args.guidance_in,
                          args.sigma_color, args.sigma_spatial, args.path_out)
    else:
        parser.print_help()
        print("If you do not have any idea what parameters to choose, "
              "try one of the following combinations:")
        for hint in PARAMETER_HINTS:
            print(hint)
    return 0


if __name__ == "__main__":
    sys.exit(main())
