"""T0: float64 *definitions* of the three operators (numpy only) -- TEST INFRASTRUCTURE.

These state the mathematics the reference asks OpenCV/Caffe for
(/root/reference/filter_reflectance.py:60-70, decompose_with_trained_CNN.py:82-95),
without any claim about rounding order.  The C restatement (rf_oracle.c, "T1")
must agree with them to within bounds justified by float32 rounding; the HIP
kernels are then compared bit-for-bit with T1.
"""
import numpy as np


def _pad(img, r, mode):
    pads = ((r, r), (r, r)) + ((0, 0),) * (img.ndim - 2)
    # np.pad 'reflect' == BORDER_REFLECT_101, 'symmetric' == BORDER_REFLECT; both
    # handle pads larger than the image by repeated reflection like borderInterpolate.
    return np.pad(img, pads, mode=mode)


def joint_bilateral_f64(joint, src, sigma_color, sigma_space, d=-1):
    """dst = sum_k s_k c(alpha_k) src_k / sum_k s_k c(alpha_k), alpha = L1 colour distance,
    taps with hypot(i,j) <= radius, radius = round_half_even(1.5 sigma_space)."""
    joint = np.atleast_3d(joint).astype(np.float64)
    srcf = np.atleast_3d(src).astype(np.float64)
    radius = int(np.round(sigma_space * 1.5)) if d <= 0 else d // 2
    radius = max(radius, 1)
    h, w = srcf.shape[:2]
    jp = _pad(joint, radius, "reflect")
    sp = _pad(srcf, radius, "reflect")
    num = np.zeros_like(srcf)
    den = np.zeros((h, w, 1))
    gc = -0.5 / (sigma_color * sigma_color)
    gs = -0.5 / (sigma_space * sigma_space)
    ntaps = 0
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            rr = np.sqrt(float(i * i + j * j))
            if rr > radius:
                continue
            ntaps += 1
            jt = jp[radius + i:radius + i + h, radius + j:radius + j + w]
            st = sp[radius + i:radius + i + h, radius + j:radius + j + w]
            alpha = np.abs(joint - jt).sum(axis=2, keepdims=True)
            wgt = np.exp(rr * rr * gs) * np.exp(alpha * alpha * gc)
            num += wgt * st
            den += wgt
    out = num / den
    return (out if np.ndim(src) == 3 else out[:, :, 0]), ntaps


def box_mean_f64(x, r):
    """Normalised (2r+1)^2 box mean with BORDER_REFLECT, in float64."""
    x = np.asarray(x, dtype=np.float64)
    p = np.pad(x, ((r, r), (r, r)), mode="symmetric")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    k = 2 * r + 1
    s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return s / float(k * k)


def guided_filter_f64(guide, src, radius, eps):
    """He et al. colour guided filter, 3-channel guide, eps added un-squared to the
    covariance diagonal of 0..255-scaled data (what the reference's call does)."""
    I = np.asarray(guide, dtype=np.float64)
    P = np.atleast_3d(src).astype(np.float64)
    h, w = I.shape[:2]
    mI = np.stack([box_mean_f64(I[:, :, c], radius) for c in range(3)], axis=2)
    cov = np.empty((h, w, 3, 3))
    for a in range(3):
        for b in range(3):
            cov[:, :, a, b] = box_mean_f64(I[:, :, a] * I[:, :, b], radius) - mI[:, :, a] * mI[:, :, b]
    cov += eps * np.eye(3)
    inv = np.linalg.inv(cov)
    out = np.empty_like(P)
    for s in range(P.shape[2]):
        p = P[:, :, s]
        mp = box_mean_f64(p, radius)
        cp = np.stack([box_mean_f64(p * I[:, :, g], radius) - mp * mI[:, :, g] for g in range(3)],
                      axis=2)
        a = np.einsum("hwgk,hwk->hwg", inv, cp)
        b = mp - (a * mI).sum(axis=2)
        ma = np.stack([box_mean_f64(a[:, :, g], radius) for g in range(3)], axis=2)
        mb = box_mean_f64(b, radius)
        out[:, :, s] = mb + (ma * I).sum(axis=2)
    return out if np.ndim(src) == 3 else out[:, :, 0]


def srgb_to_linear_f64(v):
    v = np.asarray(v, dtype=np.float64)
    return np.where(v <= 0.04045, v / 12.92, np.power((v + 0.055) / 1.055, 2.4))


def cnn_reflectance_f64(bgr_u8, weights):
    """Float64 forward of the shipped 1x1 net on a uint8 BGR image -> r in (0,1), [H,W]."""
    wts = np.asarray(weights, dtype=np.float64).ravel()
    x = srgb_to_linear_f64(np.asarray(bgr_u8)[:, :, ::-1] / 255.0).astype(np.float32).astype(np.float64)
    h, w = x.shape[:2]
    x = x.reshape(-1, 3)
    W0, b0 = wts[:96].reshape(32, 3), wts[96:128]
    cur = np.maximum(x @ W0.T + b0, 0)
    cat = [cur]
    q = 128
    for _ in range(4):
        W, b = wts[q:q + 1024].reshape(32, 32), wts[q + 1024:q + 1056]
        q += 1056
        cur = np.maximum(cur @ W.T + b, 0)
        cat.append(cur)
    wf, bf = wts[q:q + 160], wts[q + 160]
    z = np.concatenate(cat, axis=1) @ wf + bf
    return (1.0 / (1.0 + np.exp(-z))).reshape(h, w)
