"""ORACLE (test infrastructure only - never imported by the product): NumPy restatement of the reference's
``get_uint8_image`` (``/root/reference/sea_ice_drift/lib.py:27-59``), pinned by tests/golden/g6_uint8_image.npz
(outputs of the reference's own function under numpy 2.2).

``nanpercentile`` is restated step by step (numpy/lib/_function_base_impl.py, method 'linear') because the
product computes it from exact order statistics: for a float32 image NumPy divides the percentile by
float32(100), forms the virtual index (n - 1) * q in float32 and interpolates in float32
(``_lerp``: a + (b - a) * t, or b - (b - a) * (1 - t) when t >= 0.5).
"""
import numpy as np


def percentile_from_sorted(sorted_valid, p, dtype=np.float32):
    """np.nanpercentile(image, p) given the sorted non-NaN pixels (float32 image)."""
    n = sorted_valid.size
    if n == 0:
        return dtype(np.nan)
    lo, hi, t = percentile_ranks(n, p, dtype)
    a, b = sorted_valid[lo], sorted_valid[hi]
    return lerp(a, b, t)


def percentile_ranks(n, p, dtype=np.float32):
    q = np.true_divide(p, dtype(100))                 # nanpercentile: q / a.dtype.type(100) for float input
    vi = (n - 1) * q                                  # virtual index, float32
    prev = np.floor(vi)
    t = vi - prev
    lo = int(prev)
    lo = min(max(lo, 0), n - 1)
    hi = min(lo + 1, n - 1)
    return lo, hi, t


def lerp(a, b, t):
    with np.errstate(all='ignore'):
        d = np.subtract(b, a)
        r = np.add(a, d * t)
        if t >= 0.5:
            r = np.subtract(b, d * (1 - t))
    return r


def nanpercentile(image, p):
    v = image[~np.isnan(image)]
    return percentile_from_sorted(np.sort(v), p, image.dtype.type)


def get_uint8_image(image, vmin, vmax, pmin, pmax):
    """lib.py:27-59 with the percentiles from the restatement above."""
    if vmin is None:
        vmin = nanpercentile(image, pmin)
    if vmax is None:
        vmax = nanpercentile(image, pmax)
    with np.errstate(all='ignore'):
        u = 1 + 254 * (image - vmin) / (vmax - vmin)
        u[u < 1] = 1
        u[u > 255] = 255
        u[~np.isfinite(image)] = 0
        u[np.isnan(u)] = 0                            # NaN -> uint8 is undefined in C; the build defines 0
        return u.astype('uint8'), vmin, vmax
