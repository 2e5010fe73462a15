"""Weights of the shipped 1x1 reflectance CNN.

The reference loads ``learned_weights.caffemodel`` through pycaffe
(/root/reference/decompose_with_trained_CNN.py:100-106); Caffe matches blobs to
``network_definition.prototxt`` **by layer name**, so only the six convolution
layers ``conv0..conv4`` and ``fuse_skip_layers`` matter
(/root/reference/network_definition.prototxt:18-157).  This module reads that
binary ``NetParameter`` directly from the protobuf wire format (no schema, no
protobuf package) and flattens the six layers into the 4,513-float vector the
HIP kernel consumes:

    W0[32][3] b0[32] | W1[32][32] b1[32] | W2 b2 | W3 b3 | W4 b4 | wf[160] bf[1]

A pre-decoded copy (``data/cnn_weights_f32.npy``, produced by
``tests/golden/make_golden.py`` from the reference's caffemodel) ships with the
package so the CLI works without the caffemodel file.
"""
import os
import struct

import numpy as np

N_PARAMS = 4513
CONV_LAYERS = ("conv0", "conv1", "conv2", "conv3", "conv4", "fuse_skip_layers")
_EXPECTED_SHAPES = {
    "conv0": ((32, 3, 1, 1), (32,)),
    "conv1": ((32, 32, 1, 1), (32,)),
    "conv2": ((32, 32, 1, 1), (32,)),
    "conv3": ((32, 32, 1, 1), (32,)),
    "conv4": ((32, 32, 1, 1), (32,)),
    "fuse_skip_layers": ((1, 160, 1, 1), (1,)),
}
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cnn_weights_f32.npy")


def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) over one protobuf message body."""
    pos = 0
    end = len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        field, wtype = key >> 3, key & 7
        if wtype == 0:
            val, pos = _varint(buf, pos)
        elif wtype == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wtype == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wtype == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wtype)
        yield field, wtype, val


def _blob(buf):
    """BlobProto: data = field 5 (packed float or repeated fixed32), shape = field 7
    {dim = field 1}, legacy num/channels/height/width = fields 1..4."""
    data = []
    dims = []
    legacy = {}
    for field, wtype, val in _fields(buf):
        if field == 5 and wtype == 2:
            data.append(np.frombuffer(val, dtype="<f4"))
        elif field == 5 and wtype == 5:
            data.append(np.array(struct.unpack("<f", val), dtype=np.float32))
        elif field == 7 and wtype == 2:
            for f2, w2, v2 in _fields(val):
                if f2 == 1 and w2 == 2:
                    p = 0
                    while p < len(v2):
                        d, p = _varint(v2, p)
                        dims.append(d)
                elif f2 == 1 and w2 == 0:
                    dims.append(v2)
        elif field in (1, 2, 3, 4) and wtype == 0:
            legacy[field] = val
    if not dims and legacy:
        dims = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
    arr = np.concatenate(data) if data else np.zeros(0, np.float32)
    return arr.astype(np.float32).reshape(dims if dims else (-1,))


def read_caffemodel(path):
    """Return {layer_name: [blob ndarray, ...]} for every layer that carries blobs.
    NetParameter.layer = field 100; LayerParameter.name = 1, .blobs = 7."""
    with open(path, "rb") as fh:
        buf = fh.read()
    layers = {}
    for field, wtype, val in _fields(buf):
        if field != 100 or wtype != 2:
            continue
        name = None
        blobs = []
        for f2, w2, v2 in _fields(val):
            if f2 == 1 and w2 == 2:
                name = bytes(v2).decode("utf-8")
            elif f2 == 7 and w2 == 2:
                blobs.append(_blob(v2))
        if name is not None and blobs:
            layers[name] = blobs
    return layers


def flatten(layers):
    """Six conv layers -> the 4,513-float vector (layout in the module docstring)."""
    parts = []
    for name in CONV_LAYERS:
        if name not in layers or len(layers[name]) != 2:
            raise ValueError("caffemodel lacks weights+bias for layer %r" % name)
        wgt, bias = layers[name]
        want_w, want_b = _EXPECTED_SHAPES[name]
        if tuple(wgt.shape) != want_w or tuple(bias.shape) != want_b:
            raise ValueError("layer %r has shapes %s/%s, expected %s/%s"
                             % (name, wgt.shape, bias.shape, want_w, want_b))
        parts.append(wgt.reshape(-1))
        parts.append(bias.reshape(-1))
    flat = np.concatenate(parts).astype(np.float32)
    assert flat.size == N_PARAMS
    return flat


def load_weights(caffemodel=None):
    """4,513 float32 parameters: from ``caffemodel`` if given, else the shipped copy."""
    if caffemodel is not None:
        return flatten(read_caffemodel(caffemodel))
    flat = np.load(_DATA)
    if flat.dtype != np.float32 or flat.shape != (N_PARAMS,):
        raise ValueError("corrupt weight file %s" % _DATA)
    return flat
