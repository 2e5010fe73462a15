"""T1': the C restatement's operation order, re-stated a second time in numpy -- TEST INFRASTRUCTURE.

`oracle/rf_oracle.c` claims a specific float32/float64 operation order.  This module spells
the same order out independently with numpy scalars-as-arrays (every `np.float32` multiply and
add is one correctly rounded IEEE operation, `np.float64` likewise), vectorised over pixels and
sequential over taps / window positions.  tests/test_oracle.py requires the two to agree bit
for bit on small inputs, so a slip in the C code (an accidental FMA, a double where a float
belongs, a wrong tap order) cannot hide behind the float64 tolerance checks.
"""
import numpy as np

F = np.float32


def _border(p, n, delta):
    """borderInterpolate for REFLECT (delta=0) / REFLECT_101 (delta=1)."""
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p - 1 + delta if p < 0 else n - 1 - (p - n) - delta
    return p


def _pad_idx(n, r, delta):
    return np.array([_border(i - r, n, delta) for i in range(n + 2 * r)])


def joint_bilateral_f32seq(joint, src, sigma_color, sigma_space, d=-1, true_division=False):
    joint = np.atleast_3d(joint).astype(np.int32)
    s3 = np.atleast_3d(src)
    h, w, scn = s3.shape
    radius = int(np.round(sigma_space * 1.5)) if d <= 0 else d // 2
    radius = max(radius, 1)
    gc = -0.5 / (sigma_color * sigma_color)
    gs = -0.5 / (sigma_space * sigma_space)
    lut = np.exp((np.arange(256 * joint.shape[2], dtype=np.int64) ** 2).astype(np.float64)
                 * gc).astype(F)
    yi, xi = _pad_idx(h, radius, 1), _pad_idx(w, radius, 1)
    jp = joint[yi][:, xi]
    sp = s3[yi][:, xi].astype(F)
    acc = np.zeros((h, w, scn), F)
    wsum = np.zeros((h, w), F)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            rr = np.sqrt(float(i) * i + float(j) * j)
            if rr > radius:
                continue
            sw = F(np.exp(rr * rr * gs))
            jt = jp[radius + i:radius + i + h, radius + j:radius + j + w]
            st = sp[radius + i:radius + i + h, radius + j:radius + j + w]
            alpha = np.abs(joint - jt).sum(axis=2)
            wgt = (sw * lut[alpha]).astype(F)                   # one float32 multiply
            acc = (acc + (wgt[:, :, None] * st).astype(F)).astype(F)   # multiply, then add
            wsum = (wsum + wgt).astype(F)
    if true_division:
        q = (acc / wsum[:, :, None]).astype(F)
    else:
        q = (acc * (F(1.0) / wsum)[:, :, None]).astype(F)
    out = np.clip(np.rint(q), 0, 255).astype(np.uint8)          # rint = round half to even
    return out if np.ndim(src) == 3 else out[:, :, 0]


def box_mean_seq(plane, r):
    """boxFilter(CV_32F, normalize, BORDER_REFLECT) with double running sums, row then column."""
    p = np.asarray(plane, F)
    h, w = p.shape
    ks = 2 * r + 1
    scale = 1.0 / float(ks * ks)
    ext = p[:, _pad_idx(w, r, 0)].astype(np.float64)            # [h, w + 2r]
    rows = np.empty((h, w), np.float64)
    s = np.zeros(h, np.float64)
    for i in range(ks):
        s = s + ext[:, i]
    rows[:, 0] = s
    for i in range(w - 1):
        s = s + (ext[:, i + ks] - ext[:, i])
        rows[:, i + 1] = s
    yidx = _pad_idx(h, r, 0)                                    # index into rows for y - r .. y + r
    total = np.zeros(w, np.float64)
    for t in range(ks - 1):
        total = total + rows[yidx[t]]
    out = np.empty((h, w), F)
    for y in range(h):
        s0 = total + rows[yidx[y + ks - 1]]
        out[y] = (s0 * scale).astype(F)
        total = s0 - rows[yidx[y]]
    return out


def guided_filter_f32seq(guide, src, radius, eps):
    I = [np.asarray(guide)[:, :, c].astype(F) for c in range(3)]
    P = np.atleast_3d(src)
    box = lambda x: box_mean_seq(x, radius)  # noqa: E731
    mI = [box(I[c]) for c in range(3)]
    cov = {}
    for a in range(3):
        for b in range(a, 3):
            c = box((I[a] * I[b]).astype(F))
            prod = (mI[a] * mI[b]).astype(F)
            if a == b:
                prod = (prod + F(-F(eps))).astype(F)
            cov[(a, b)] = cov[(b, a)] = (c - prod).astype(F)
    inv = {}
    for k in range(3):
        for l in range(k + 1):
            a00, a01 = cov[((k + 1) % 3, (l + 1) % 3)], cov[((k + 1) % 3, (l + 2) % 3)]
            a10, a11 = cov[((k + 2) % 3, (l + 1) % 3)], cov[((k + 2) % 3, (l + 2) % 3)]
            v = (a00 * a11).astype(F)
            inv[(k, l)] = inv[(l, k)] = (v - (a01 * a10).astype(F)).astype(F)
    det = (cov[(0, 0)] * inv[(0, 0)]).astype(F)
    det = (det + (cov[(1, 0)] * inv[(1, 0)]).astype(F)).astype(F)
    det = (det + (cov[(2, 0)] * inv[(2, 0)]).astype(F)).astype(F)
    if eps < 1e-2:
        det = np.where(np.abs(det) < F(1e-6), F(1.0), det)
    for key in list(inv):
        if key[0] >= key[1]:
            inv[key] = inv[(key[1], key[0])] = (inv[key] / det).astype(F)
    out = np.empty(P.shape, F)
    for s in range(P.shape[2]):
        p = P[:, :, s].astype(F)
        mp = box(p)
        cp = [(box((p * I[g]).astype(F)) - (mp * mI[g]).astype(F)).astype(F) for g in range(3)]
        al = []
        for g in range(3):
            a = (inv[(g, 0)] * cp[0]).astype(F)
            a = (a + (inv[(g, 1)] * cp[1]).astype(F)).astype(F)
            a = (a + (inv[(g, 2)] * cp[2]).astype(F)).astype(F)
            al.append(a)
        beta = mp
        for g in range(3):
            beta = (beta - (al[g] * mI[g]).astype(F)).astype(F)
        q = box(beta)
        for g in range(3):
            q = (q + (box(al[g]) * I[g]).astype(F)).astype(F)
        out[:, :, s] = q
    u8 = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return (u8, out) if np.ndim(src) == 3 else (u8[:, :, 0], out[:, :, 0])
