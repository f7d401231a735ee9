"""CPU oracle (NumPy form) for the sea_ice_drift pattern-matching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sea_ice_drift_amd/`` may import this
module: it exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` can check the HIP path.  The product path
fails loudly when the HIP library is missing; it never falls back to this code.

It restates, function by function, the per-grid-point operator of
``/root/reference/sea_ice_drift/pmlib.py`` (reference @ v0.7.1):

====================  =====================  ===================================
here                  reference              what
====================  =====================  ===================================
rotation_terms        pmlib.py:105-110       tc, cos/sin matrix, tc.dot(transform)
get_template          pmlib.py:89-115        rotated nearest-neighbour template
get_template_order1   pmlib.py:89-115        the same with rot_order=1 (bilinear)
match_template        pmlib.py:156 (cv2)     TM_CCOEFF_NORMED, see below
get_hessian           pmlib.py:36-59         float32 2nd-derivative magnitude
rotate_and_match      pmlib.py:117-174       angle sweep, peak pick, Hessian
use_mcc               pmlib.py:176-212       window slicing, displacement
pm_batch              pmlib.py:436-448,462   the Pool.map seam -> (N,5) array
====================  =====================  ===================================

Parity status
-------------
* Everything except ``match_template`` is pinned against the reference's own
  Python, imported in the build container (``oracle/ref_harness.py``) and frozen
  as fixtures under ``tests/golden/`` by ``tests/golden/make_golden.py``.
* ``match_template`` restates ``cv2.matchTemplate(..., cv2.TM_CCOEFF_NORMED)``.
  OpenCV (un-vendored, version unpinned: reference README.md:32, .travis.yml:13)
  is absent from this image and the reference's tests assert no number at that
  call (tests.py:332-346), so **parity is unpinned at the cv2 boundary**.  The
  restatement follows OpenCV's published algorithm (imgproc/templmatch.cpp,
  ``common_matchTemplate``) with the raw correlation taken as exact integer sums
  instead of OpenCV's float32 DFT, so it is deterministic and at least as
  accurate; it is cross-checked against an independent float64 brute-force NCC.

NCC specification (shared by this file, oracle/pm_oracle.c and the HIP kernel)
------------------------------------------------------------------------------
For a window placement with N = s*s pixels, exact integers::

    S_I = sum W      S_II = sum W*W      S_T = sum T      S_TT = sum T*T
    S_IT = sum W*T
    numer = N*S_IT - S_I*S_T     dI = N*S_II - S_I**2     dT = N*S_TT - S_T**2

then in IEEE double, one rounding per operation, no fused multiply-add::

    rI = 1.0 / sqrt(double(dI))          (per placement)
    rT = 1.0 / sqrt(double(dT))          (per template)
    q  = (double(numer) * rI) * rT
    R  = q if |q| < 1 ; copysign(1, q) if |q| < 1.125 ; else 0      (OpenCV's clamp)
    R  = 0 where the window variance is negligible:
         2*dI <= N  and  dI * 2**23 <= 10 * N * S_II
         (OpenCV: diff2 <= min(0.5, 10*FLT_EPSILON*wndSum2), diff2 = dI/N)
    R  = 1 everywhere when dT == 0 (OpenCV: constant template)
    result = float32(R)
"""
from __future__ import annotations

import numpy as np

FLAG_HES_NORM = 1
FLAG_HES_SMTH = 2
FLAG_MCC_NORM = 4
FLAG_ROT_ORDER1 = 8          # rot_order=1 (pmlib.py:89): templates sampled bilinearly


def flag_rot_order(order):
    """rot_order 0..5 as flag bits 3..5 (include/sid_pm.h SID_PM_ROT_ORDER): order 1 = FLAG_ROT_ORDER1."""
    return (int(order) & 7) << 3


def rot_order_of(flags):
    return (int(flags) >> 3) & 7


# --------------------------------------------------------------------------- a1
def rotation_terms(angle_deg, img_size):
    """(cos a, sin a, tcT0, tcT1) exactly as pmlib.py:105-110 builds them.

    tc = int(s/2.)+1 for both axes; transform = [[cos,-sin],[sin,cos]];
    tcT = tc.dot(transform) (kept as NumPy's own dot so rounding is NumPy's).
    """
    tc = int(img_size / 2.) + 1
    tc = np.array([tc, tc])
    a = np.radians(angle_deg)
    transform = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    tct = tc.dot(transform)
    return float(transform[0, 0]), float(transform[1, 0]), float(tct[0]), float(tct[1])


def get_template(img, c, r, a, s):
    """Rotated s*s uint8 template around (c, r) on ``img`` (pmlib.py:89-115,
    rot_order=0).  Closed form of scipy's NI_GeometricTransform at order 0 with
    mode='constant', cval=0:  coordinate = (0.0 + i*M00) + j*M01 + off, sampled at
    floor(coordinate + 0.5); a coordinate < 0 or > dim-1 yields 0.
    """
    cosa, sina, tct0, tct1 = rotation_terms(a, s)
    off0 = np.float64(r) - tct0
    off1 = np.float64(c) - tct1
    ii = np.arange(s, dtype=np.float64)[:, None]
    jj = np.arange(s, dtype=np.float64)[None, :]
    # matrix passed to scipy is transform.T = [[cos, sin], [-sin, cos]]
    rr = ((0.0 + ii * cosa) + jj * sina) + off0
    cc = ((0.0 + ii * (-sina)) + jj * cosa) + off1
    rows, cols = img.shape
    inside = (rr >= 0) & (rr <= rows - 1) & (cc >= 0) & (cc <= cols - 1)
    ri = np.floor(rr + 0.5).astype(np.int64)
    ci = np.floor(cc + 0.5).astype(np.int64)
    ri = np.clip(ri, 0, rows - 1)
    ci = np.clip(ci, 0, cols - 1)
    out = np.where(inside, img[ri, ci], 0).astype(np.uint8)
    return out


def get_template_order1(img, c, r, a, s):
    """The same template with ``rot_order=1`` (pmlib.py:89,112-113): scipy's NI_GeometricTransform at spline order 1
    with mode='constant', cval=0, output=uint8.  Same float64 coordinates as order 0; a coordinate < 0 or > dim-1 yields
    cval = 0 for the whole sample (no blending with the border - that would be mode='grid-constant'); otherwise
    start = floor(coordinate), weights (1 - y, y) with y = coordinate - floor(coordinate), the tap behind the last sample
    mirrored (its weight is then exactly 0), and

        t = 0.0;  t += (v[r0,c0]*w0r)*w0c;  t += (v[r0,c1]*w0r)*w1c;  t += (v[r1,c0]*w1r)*w0c;  t += (v[r1,c1]*w1r)*w1c

    in float64, in that order; uint8 output:  t > 0 ? t + 0.5 : 0, clamped to 255, truncated.  Pinned against scipy
    through the reference's own get_template: fixture G1b (tests/golden/g1b_templates_order1.npz).
    """
    cosa, sina, tct0, tct1 = rotation_terms(a, s)
    off0 = np.float64(r) - tct0
    off1 = np.float64(c) - tct1
    ii = np.arange(s, dtype=np.float64)[:, None]
    jj = np.arange(s, dtype=np.float64)[None, :]
    rr = ((0.0 + ii * cosa) + jj * sina) + off0
    cc = ((0.0 + ii * (-sina)) + jj * cosa) + off1
    rows, cols = img.shape
    inside = (rr >= 0) & (rr <= rows - 1) & (cc >= 0) & (cc <= cols - 1)
    fr, fc = np.floor(rr), np.floor(cc)
    yr, yc = rr - fr, cc - fc
    r0 = np.clip(fr.astype(np.int64), 0, rows - 1)
    c0 = np.clip(fc.astype(np.int64), 0, cols - 1)

    def mirrored(i, n):                           # scipy's edge mapping of a tap index beyond the last sample
        i = np.where(i >= n, 2 * n - 2 - i, i)
        return np.clip(i, 0, n - 1)
    r1, c1 = mirrored(r0 + 1, rows), mirrored(c0 + 1, cols)
    v = img.astype(np.float64)
    w0r, w1r, w0c, w1c = 1.0 - yr, yr, 1.0 - yc, yc
    t = 0.0
    t = t + (v[r0, c0] * w0r) * w0c
    t = t + (v[r0, c1] * w0r) * w1c
    t = t + (v[r1, c0] * w1r) * w0c
    t = t + (v[r1, c1] * w1r) * w1c
    t = np.where(t > 0, t + 0.5, 0.0)
    t = np.minimum(t, 255.0)
    return np.where(inside, t, 0.0).astype(np.uint8)


# ----------------------------------------------------------------- a1, rot_order 2..5
# scipy.ndimage.affine_transform(img, ..., order=n, mode='constant', cval=0, output=uint8, prefilter=True) for n = 2..5
# (pmlib.py:112-113 with rot_order=n): the WHOLE image goes through scipy's recursive B-spline prefilter first
# (spline_filter(input, order, output=float64, mode='constant'): per axis, axis 0 first; 'constant' takes the MIRROR boundary
# formulas of ni_splines.c), then every sample is the tensor product of order + 1 spline weights per axis over the coefficient
# image.  Restated here operation for operation; pinned against scipy itself - coefficients and weights bit for bit (probed
# through spline_filter1d and map_coordinates(prefilter=False) on impulses: tests/test_oracle_golden.py) - and against the
# reference's get_template(rot_order=n) (fixture G1c).
SPLINE_POLES = {2: (-0.171572875253809902396622551581,),
                3: (-0.267949192431122706472553658494,),
                4: (-0.361341225900220177092212841325, -0.013725429297339121360331226939),
                5: (-0.430575347099973791851434783493, -0.043096288203264653822712839920)}


def spline_filter_lines(c, order):
    """scipy's apply_filter (ni_splines.c) along axis 0 of a float64 array, every column a line: gain, then per pole the causal
    initialisation (mirror), the causal recursion, the anticausal initialisation, the anticausal recursion."""
    import math
    c = np.array(c, dtype=np.float64)
    n = c.shape[0]
    if n <= 1:
        return c
    gain = 1.0
    for z in SPLINE_POLES[order]:
        gain *= (1.0 - z) * (1.0 - 1.0 / z)
    c *= gain
    for z in SPLINE_POLES[order]:
        z_n_1 = math.pow(z, n - 1)
        z_i = z
        c0 = c[0] + z_n_1 * c[n - 1]
        for i in range(1, n - 1):
            if z_i == 0.0:                                        # (the remaining terms are +-0: they change nothing)
                break
            c0 = c0 + z_i * (c[i] + z_n_1 * c[n - 1 - i])
            z_i *= z
        c0 = c0 / (1 - z_n_1 * z_n_1)
        c[0] = c0
        for i in range(1, n):
            c[i] += z * c[i - 1]
        c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1)
        for i in range(n - 2, -1, -1):
            c[i] = z * (c[i + 1] - c[i])
    return c


def spline_coefficients(img, order):
    """spline_filter(img, order, output=float64) of a 2-D image: axis 0, then axis 1."""
    c = spline_filter_lines(np.asarray(img, dtype=np.float64), order)
    return np.ascontiguousarray(spline_filter_lines(c.T, order).T)


def spline_weights(x, order):
    """get_spline_interpolation_weights (ni_splines.c) for an array of coordinates: list of order + 1 weight arrays."""
    x = np.asarray(x, dtype=np.float64)
    x = x - np.floor(x if order & 1 else x + 0.5)
    w = [None] * (order + 1)
    y = x
    z = 1.0 - x
    if order == 2:
        w[1] = 0.75 - x * x
        y = 0.5 - x
        w[0] = 0.5 * y * y
    elif order == 3:
        w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
        w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0
        w[0] = z * z * z / 6.0
    elif order == 4:
        t = x * x
        w[2] = t * (t * 0.25 - 0.625) + 115.0 / 192.0
        y = 1.0 + x
        w[1] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0
        z = 1.0 - x
        w[3] = z * (z * (z * (5.0 - z) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0
        y = 0.5 - x
        t = y * y
        w[0] = t * t / 24.0
    elif order == 5:
        t = y * y
        w[2] = t * (t * (0.25 - y / 12.0) - 0.5) + 0.55
        t = z * z
        w[3] = t * (t * (0.25 - z / 12.0) - 0.5) + 0.55
        y = x + 1.0
        w[1] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425
        z2 = 2.0 - x
        w[4] = z2 * (z2 * (z2 * (z2 * (z2 / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425
        z = 1.0 - x
        t = z * z
        w[0] = z * t * t / 120.0
    else:
        raise ValueError('order 2..5')
    last = np.ones_like(x)
    for i in range(order):
        last = last - w[i]
    w[order] = last
    return w


def get_template_spline(img, c, r, a, s, order, coeffs=None):
    """get_template(img, c, r, a, s, rot_order=order) for order 2..5 (pmlib.py:89-115).  ``coeffs``: the prefiltered image
    (spline_coefficients(img, order)) when the caller has it already."""
    if coeffs is None:
        coeffs = spline_coefficients(img, order)
    cosa, sina, tct0, tct1 = rotation_terms(a, s)
    off0 = np.float64(r) - tct0
    off1 = np.float64(c) - tct1
    ii = np.arange(s, dtype=np.float64)[:, None]
    jj = np.arange(s, dtype=np.float64)[None, :]
    rr = ((0.0 + ii * cosa) + jj * sina) + off0
    cc = ((0.0 + ii * (-sina)) + jj * cosa) + off1
    rows, cols = img.shape
    inside = (rr >= 0) & (rr <= rows - 1) & (cc >= 0) & (cc <= cols - 1)
    rr = np.where(inside, rr, 0.0)
    cc = np.where(inside, cc, 0.0)
    if order & 1:
        sr, sc = np.floor(rr).astype(np.int64) - order // 2, np.floor(cc).astype(np.int64) - order // 2
    else:
        sr, sc = np.floor(rr + 0.5).astype(np.int64) - order // 2, np.floor(cc + 0.5).astype(np.int64) - order // 2
    wr, wc = spline_weights(rr, order), spline_weights(cc, order)

    def mirror(i, n):                                               # scipy's edge mapping of a tap index (NI_GeometricTransform)
        if n <= 1:
            return np.zeros_like(i)
        s2 = 2 * n - 2
        neg = i < 0
        j = np.where(neg, s2 * (-i // s2) + i, i)
        j = np.where(neg, np.where(j <= 1 - n, j + s2, -j), j)
        big = ~neg & (i >= n)
        k = np.where(big, i - s2 * (i // s2), j)
        k = np.where(big & (k >= n), s2 - k, k)
        return k
    t = np.zeros((s, s), dtype=np.float64)
    for ta in range(order + 1):
        ia = mirror(sr + ta, rows)
        for tb in range(order + 1):
            ib = mirror(sc + tb, cols)
            t = t + (coeffs[ia, ib] * wr[ta]) * wc[tb]
    t = np.where(t > 0, t + 0.5, 0.0)
    t = np.minimum(t, 255.0)
    return np.where(inside, t, 0.0).astype(np.uint8)


# --------------------------------------------------------------------------- a3
def window_sums(image, s):
    """Exact S_I and S_II for every placement of an s*s box (int64)."""
    img = image.astype(np.int64)
    def box(x):
        ii = np.zeros((x.shape[0] + 1, x.shape[1] + 1), dtype=np.int64)
        ii[1:, 1:] = x.cumsum(0).cumsum(1)
        return ii[s:, s:] - ii[:-s, s:] - ii[s:, :-s] + ii[:-s, :-s]
    return box(img), box(img * img)


def raw_correlation(image, templ):
    """Exact sum W*T for every placement (int64)."""
    s0, s1 = templ.shape
    rh, rw = image.shape[0] - s0 + 1, image.shape[1] - s1 + 1
    img = image.astype(np.int64)
    acc = np.zeros((rh, rw), dtype=np.int64)
    for i in range(s0):
        for j in range(s1):
            t = int(templ[i, j])
            if t:
                acc += t * img[i:i + rh, j:j + rw]
    return acc


def match_template(image, templ, mtype=None):
    """Restated cv2.matchTemplate(image, templ, cv2.TM_CCOEFF_NORMED) (pmlib.py:156).

    Signature matches the reference's ``template_matcher`` plug point
    (pmlib.py:119-120).  See the module docstring for the arithmetic.
    """
    s = templ.shape[0]
    assert templ.shape[0] == templ.shape[1], 'square templates only on this path'
    n = s * s
    s_i, s_ii = window_sums(image, s)
    s_it = raw_correlation(image, templ)
    t64 = templ.astype(np.int64)
    s_t = int(t64.sum())
    s_tt = int((t64 * t64).sum())
    d_t = n * s_tt - s_t * s_t
    if d_t == 0:
        return np.ones(s_it.shape, dtype=np.float32)
    numer = n * s_it - s_i * s_t
    d_i = n * s_ii - s_i * s_i
    small = 2 * d_i <= n                      # guards the shift below against overflow
    lowvar = small & (np.where(small, d_i, 0) * (1 << 23) <= 10 * n * s_ii)
    with np.errstate(divide='ignore', invalid='ignore'):
        r_i = 1.0 / np.sqrt(d_i.astype(np.float64))
        r_t = 1.0 / np.sqrt(np.float64(d_t))
        q = (numer.astype(np.float64) * r_i) * r_t
    aq = np.abs(q)
    res = np.where(aq < 1.0, q, np.where(aq < 1.125, np.copysign(1.0, q), 0.0))
    res = np.where(lowvar, 0.0, res)
    return res.astype(np.float32)


def ncc_bruteforce_f64(image, templ):
    """Independent zero-mean NCC in float64 (definition, not OpenCV's algebra);
    used only to cross-check match_template in tests."""
    s = templ.shape[0]
    rh, rw = image.shape[0] - s + 1, image.shape[1] - s + 1
    t = templ.astype(np.float64)
    t = t - t.mean()
    tn = np.sqrt((t * t).sum())
    out = np.zeros((rh, rw))
    for y in range(rh):
        for x in range(rw):
            w = image[y:y + s, x:x + s].astype(np.float64)
            w = w - w.mean()
            d = np.sqrt((w * w).sum()) * tn
            out[y, x] = (w * t).sum() / d if d > 0 else 0.0
    return out


# --------------------------------------------------------------------------- a5
def _gradient_f32(f, axis):
    """1-D second-order central difference with one-sided edges, float32, the
    arithmetic np.gradient applies with unit spacing (pmlib.py:51-54)."""
    f = np.asarray(f, dtype=np.float32)
    f = np.moveaxis(f, axis, 0)
    g = np.empty_like(f)
    g[1:-1] = (f[2:] - f[:-2]) / np.float32(2.0)
    g[0] = f[1] - f[0]
    g[-1] = f[-1] - f[-2]
    return np.moveaxis(g, 0, axis)


def raw_hessian(ccm):
    """hypot(d2/dx2, d2/dy2) in float32 (pmlib.py:51-55)."""
    ccm = np.asarray(ccm, dtype=np.float32)
    d2x = _gradient_f32(_gradient_f32(ccm, 1), 1)
    d2y = _gradient_f32(_gradient_f32(ccm, 0), 0)
    return np.hypot(d2x, d2y)


def get_hessian(ccm, hes_norm=True, hes_smth=False):
    """pmlib.py:36-59."""
    ccm = np.asarray(ccm, dtype=np.float32)
    if hes_smth:
        from scipy import ndimage as nd
        ccm = nd.gaussian_filter(ccm, 1)
    hes = raw_hessian(ccm)
    if hes_norm:
        hes = (hes - np.median(hes)) / np.std(hes)
    return hes


# ----------------------------------------------------------------------- a2,a4,a6
def rotate_and_match(img1, c1, r1, img_size, image2, alpha0, angles=(-3, 0, 3),
                     mcc_norm=False, hes_norm=True, hes_smth=False, full=False, rot_order=0, coeffs=None):
    """pmlib.py:117-174 (rot_order: 0 or 1, forwarded to get_template as pmlib.py:151 does).  Returns (dc, dr, best_a, best_r, best_h) and, with
    full=True, also (best_ij, best_angle_index, best_result, best_template)."""
    nan = np.nan
    best_r = -np.inf
    best = None
    for k, angle in enumerate(angles):
        if rot_order >= 2:                                         # (coeffs: spline_coefficients(img1, rot_order), when the caller has them)
            template = get_template_spline(img1, c1, r1, angle - alpha0, img_size, rot_order, coeffs=coeffs)
        else:
            template = (get_template_order1 if rot_order == 1 else get_template)(img1, c1, r1, angle - alpha0, img_size)
        if template.min() == 0:                                   # pmlib.py:152-154
            if full:
                return (nan, nan, nan, nan, nan), ((-1, -1), -1, None, None)
            return nan, nan, nan, nan, nan
        result = match_template(image2, template)
        ij = np.unravel_index(np.argmax(result), result.shape)   # first max, row-major
        if result.max() > best_r:                                 # strict: first angle wins ties
            best_r = result.max()
            best = (angle, result, template, ij, k)
    best_a, best_result, best_template, best_ij, best_k = best
    best_h = get_hessian(best_result, hes_norm=hes_norm, hes_smth=hes_smth)[best_ij]
    dr = best_ij[0] - (image2.shape[0] - img_size) / 2.
    dc = best_ij[1] - (image2.shape[1] - img_size) / 2.
    if mcc_norm:
        best_r = (best_r - np.median(best_result)) / np.std(best_result)
    out = (dc, dr, best_a, best_r, best_h)
    if full:
        return out, ((int(best_ij[0]), int(best_ij[1])), best_k, best_result, best_template)
    return out


def window_bounds(c2fg, r2fg, border, img_size):
    """Row/col slice limits of the search window (pmlib.py:200-202)."""
    hws = int(img_size / 2.)
    return (int(r2fg - hws - border), int(r2fg + hws + border + 1),
            int(c2fg - hws - border), int(c2fg + hws + border + 1))


def use_mcc(c1, r1, c2fg, r2fg, border, img1, img2, img_size, alpha0, full=False, **kw):
    """pmlib.py:176-212.  A window that is not wholly inside img2, or that leaves fewer
    than 2 placements per axis (np.gradient needs 2), is never produced by the reference's
    validity mask (pmlib.py:417-426); it yields NaN*5 here and in the HIP kernel."""
    r0, r1_, c0, c1_ = window_bounds(c2fg, r2fg, border, img_size)
    ok = (0 <= r0 and 0 <= c0 and r1_ <= img2.shape[0] and c1_ <= img2.shape[1]
          and r1_ - r0 >= img_size + 1 and c1_ - c0 >= img_size + 1)
    if not ok:
        nan = np.nan
        return ((nan,) * 5, ((-1, -1), -1, None, None)) if full else (nan,) * 5
    image = img2[r0:r1_, c0:c1_]
    res = rotate_and_match(img1, c1, r1, img_size, image, alpha0, full=full, **kw)
    (dc, dr, a, r, h) = res[0] if full else res
    out = (c2fg + dc, r2fg + dr, a, r, h)
    return (out, res[1]) if full else out


# --------------------------------------------------------------------------- a7
def pm_batch(img1, img2, c1, r1, c2fg, r2fg, border, img_size, alpha0, angles,
             flags=FLAG_HES_NORM):
    """The batch seam of pmlib.py:436-448 + :462 : (N,5) float64 and (N,3) int32
    [best row, best col, best angle index] (-1 for NaN points)."""
    n = len(c1)
    out = np.full((n, 5), np.nan)
    ij = np.full((n, 3), -1, dtype=np.int32)
    order = rot_order_of(flags)
    kw = dict(angles=list(angles), hes_norm=bool(flags & FLAG_HES_NORM),
              hes_smth=bool(flags & FLAG_HES_SMTH), mcc_norm=bool(flags & FLAG_MCC_NORM),
              rot_order=order, coeffs=spline_coefficients(img1, order) if order >= 2 else None)
    for i in range(n):
        res, (bij, bk, _, _) = use_mcc(c1[i], r1[i], c2fg[i], r2fg[i], border[i], img1, img2,
                                       img_size, alpha0, full=True, **kw)
        out[i] = res
        ij[i] = (bij[0], bij[1], bk)
    return out, ij
