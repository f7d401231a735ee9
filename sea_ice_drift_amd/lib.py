"""Host-side helpers the PM path needs from the reference's lib.py.

Only what ``pattern_matching`` touches: the two first-guess interpolators
(reference lib.py:139-177 and :179-201), the grid scatter (lib.py:408-412) and a
stand-in for ``nansat.NSR``.  Image staging from files (lib.py:27-59, 256-340) and the
geo/Haversine helpers are outside the hot path.
"""
import numpy as np
from scipy.interpolate import griddata

try:                                      # a real nansat passes through untouched
    from nansat import NSR                # pragma: no cover - not installed in this image
except Exception:                         # noqa: BLE001
    class NSR(object):
        """Placeholder for nansat.NSR: carries the srs string to duck-typed domains."""
        def __init__(self, srs=None):
            self.srs = srs
            self.wkt = srs

        def __repr__(self):
            return 'NSR(%r)' % (self.srs,)


def _design_matrix(x, y, order):
    cols = [np.ones(len(x)), x, y]
    if order > 1:
        cols += [x ** 2, y ** 2, x * y]
    if order > 2:
        cols += [x ** 3, y ** 3, x ** 2 * y, y ** 2 * x]
    return np.vstack(cols).T


def interpolation_poly(x1, y1, x2, y2, x1grd, y1grd, order=1, **kwargs):
    """Least-squares polynomial map (x1,y1)->(x2,y2) evaluated on the grid points
    (reference lib.py:139-177; term order 1, x, y, x^2, y^2, xy, x^3, y^3, x^2 y, y^2 x)."""
    A = _design_matrix(x1, y1, order)
    bx = np.linalg.lstsq(A, x2, rcond=-1)[0]
    by = np.linalg.lstsq(A, y2, rcond=-1)[0]
    xf, yf = x1grd.flatten(), y1grd.flatten()
    G = _design_matrix(xf, yf, order)
    return np.dot(G, bx).reshape(x1grd.shape), np.dot(G, by).reshape(x1grd.shape)


def interpolation_near(x1, y1, x2, y2, x1grd, y1grd, method='linear', **kwargs):
    """scipy griddata of x2/y2 from the keypoints onto the grid points; NaN outside the
    convex hull (reference lib.py:179-201; note the (row, col) point order)."""
    src = np.array([y1, x1]).T
    dst = np.array([y1grd, x1grd]).T
    return griddata(src, x2, dst, method=method).T, griddata(src, y2, dst, method=method).T


def _fill_gpi(shape, gpi, data):
    """Scatter the values of the valid points into a NaN grid (reference lib.py:408-412)."""
    y = np.zeros(shape).flatten() + np.nan
    y[gpi] = data
    return y.reshape(shape)
