#!/usr/bin/env python
"""Intrinsic decomposition with the shipped 1x1 reflectance CNN, on the MI355X.

Same functions, outputs and command line as the reference's ``decompose_with_trained_CNN.py``
(/root/reference/decompose_with_trained_CNN.py:57-148).  The Caffe net is replaced by
``ReflectanceNet``, a small object with pycaffe's surface (``blobs[name].reshape/.data``,
``forward()``) whose forward pass is the HIP kernel behind ``rf_cnn_reflectance_u8``.

Image convention on the blob side: linear RGB, channels x height x width, range 0..1.
"""
from __future__ import division, print_function

import argparse
import sys
import os

import numpy as np

from . import image_utils as iu
from . import ops, _ffi
from . import weights as _weights


def imgCV2_to_caffeBlob(img):
    """uint8 BGR HxWx3 -> float64 blob [1,3,H,W] in linear RGB
    (/root/reference/decompose_with_trained_CNN.py:57-69)."""
    rgb01 = (img / 255.0)[:, :, ::-1]
    linear = iu.srgb_to_rgb(rgb01)
    return np.transpose(linear, (2, 0, 1))[np.newaxis, :, :, :]


def caffeBlob_to_imgGrayLinear(blob):
    """[1,1,H,W] blob -> HxW image; anything else is an error
    (/root/reference/decompose_with_trained_CNN.py:72-79)."""
    b, c = blob.shape[:2]
    if b != 1 or c != 1:
        raise ValueError("Expecting to get 1 image in mini-batch having 1 channel, "
                         "but got batch size of {} and {} channels".format(b, c))
    return blob[0, 0, :, :]


class _Blob(object):
    def __init__(self, shape):
        self.data = np.zeros(shape, dtype=np.float32)

    def reshape(self, *shape):
        self.data = np.zeros(shape, dtype=np.float32)


class ReflectanceNet(object):
    """pycaffe-shaped wrapper of the HIP forward pass.

    ``blobs['images'].data`` holds the float32 linear-RGB blob exactly as the reference fills it;
    since every blob value is ``float32(srgb_to_rgb(v/255))`` for a byte v, forward() recovers
    the bytes through the same 256-entry table and hands uint8 BGR to the kernel (3 B/pixel
    over PCIe instead of 12).  Blobs that were not produced from 8-bit pixels are rejected.
    """

    def __init__(self, caffemodel=None):
        self.weights = _weights.load_weights(caffemodel)
        self.blobs = {"images": _Blob((1, 3, 256, 256)),
                      "reflectance_intensity": _Blob((1, 1, 256, 256))}
        self._lut = iu.srgb_byte_lut()

    def _blob_to_bytes(self):
        x = self.blobs["images"].data
        if x.ndim != 4 or x.shape[1] != 3:
            raise ValueError("blob 'images' must be [N,3,H,W]")
        idx = np.searchsorted(self._lut, x)
        idx = np.clip(idx, 0, 255)
        if not np.array_equal(self._lut[idx], x):
            raise ValueError("blob 'images' does not hold sRGB byte levels; "
                             "use get_reflectance_batch() on uint8 images instead")
        # blob is RGB, channel-first -> BGR, channel-last
        return np.ascontiguousarray(idx.astype(np.uint8).transpose(0, 2, 3, 1)[:, :, :, ::-1])

    def forward(self):
        torch = _ffi.require_gpu()
        bgr = torch.from_numpy(self._blob_to_bytes()).cuda()
        r, _ = ops.cnn_reflectance_u8(bgr, weights=self.weights, want_u8=False)
        self.blobs["reflectance_intensity"].data = r.cpu().numpy()[:, np.newaxis, :, :]
        return {"reflectance_intensity": self.blobs["reflectance_intensity"].data}


def get_reflectance_caffe(net, image):
    """Run one uint8 BGR image through ``net`` and return the HxW float32 reflectance
    intensity (/root/reference/decompose_with_trained_CNN.py:82-95)."""
    height, width = image.shape[:2]
    net.blobs["images"].reshape(1, 3, height, width)
    net.blobs["images"].data[...] = imgCV2_to_caffeBlob(image)
    net.forward()
    return caffeBlob_to_imgGrayLinear(net.blobs["reflectance_intensity"].data)


def get_reflectance_batch(images, weights=None):
    """Device-resident batch form: CUDA uint8 BGR [N,H,W,3] -> (r float32 [N,H,W],
    r_u8 uint8 [N,H,W] = the `-r.png` bytes)."""
    return ops.cnn_reflectance_u8(images, weights=weights)


def decompose_and_filter_batch(images, sigma_color=20.0, sigma_spatial=22.0, weights=None):
    """BF(CNN, CNN) on the device, the paper's headline pipeline (README.md:56 of the reference):
    CUDA uint8 BGR [N,H,W,3] -> (r_u8 [N,H,W], filtered [N,H,W]) with `filtered` bit-identical to
    what the two CLIs produce through `<base>-r.png` (grey PNG re-read as 3 equal channels,
    filtered with itself as guidance, any channel of the result)."""
    _, r8 = ops.cnn_reflectance_u8(images, weights=weights, want_float=False)
    r1 = r8.unsqueeze(-1)
    out = ops.joint_bilateral_u8(r1, r1, -1, sigma_color, sigma_spatial, grey_as_bgr=True)
    return r8, out.squeeze(-1)


def decompose_batch(images, weights=None):
    """Everything decompose_image writes, for a device-resident batch: CUDA uint8 BGR
    [N,H,W,3] -> (r float32 [N,H,W], r_u8 [N,H,W] = `-r.png`, reflectance bytes [N,H,W,3] =
    `-r_colorized.png`, shading bytes [N,H,W] = `-s_colorized.png`), no host arithmetic
    (/root/reference/decompose_with_trained_CNN.py:113-128)."""
    r, r8 = ops.cnn_reflectance_u8(images, weights=weights)
    refl, shad = ops.colorize_srgb_u8(images, r)
    return r, r8, refl, shad


def decompose_image(filename_in, path_out, caffemodel=None):
    """Predict reflectance intensity for one image file and write `<base>-r.png`,
    `<base>-r_colorized.png`, `<base>-s_colorized.png`
    (/root/reference/decompose_with_trained_CNN.py:98-130)."""
    net = ReflectanceNet(caffemodel)
    image = iu.imread(filename_in)
    basename = os.path.splitext(os.path.basename(filename_in))[0]
    reflectance_gray = get_reflectance_caffe(net, image)
    iu.imwrite(os.path.join(path_out, basename + "-r.png"), reflectance_gray)
    reflectance, shading = iu.colorize(reflectance_gray, image)
    iu.imwrite(os.path.join(path_out, basename + "-r_colorized.png"), reflectance, sRGB=True)
    iu.imwrite(os.path.join(path_out, basename + "-s_colorized.png"), shading, sRGB=True)
    return reflectance_gray


def build_parser():
    parser = argparse.ArgumentParser(
        description="Decompose an image with the direct reflectance prediction CNN "
                    "(MI355X build).")
    parser.add_argument("--filename_in", help="image that should be decomposed")
    parser.add_argument("--path_out", help="existing folder that receives the decomposition")
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(sys.argv[1:] if argv is None else argv)
    if args.filename_in and args.path_out:
        decompose_image(args.filename_in, args.path_out)
    else:
        parser.print_help()
    return 0


if __name__ == "__main__":
    sys.exit(main())
