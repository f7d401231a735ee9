"""A georeferenced uint8 array with the small part of the Nansat interface the PM path uses.

nansat / GDAL are not installed on the build or GPU machines, and the benchmark runs on
synthetic arrays, so ``pattern_matching`` accepts any object with these members (SURVEY.md
Appendix A; call sites in the reference: pmlib.py:81-83, 279-282, 394, 398, 410-411,
421-426, 473-481; seaicedrift.py:85-86):

    n[1]                                  band 1 as a 2-D uint8 array
    n.shape()                             (rows, cols)
    n.transform_points(x, y, 0)           pixel (col, row) -> (lon, lat)
    n.transform_points(lon, lat, 1)       (lon, lat) -> pixel (col, row), float
    n.transform_points(c, r, 0, nsr)      pixel -> coordinates in the destination SRS
    n.get_corners()                       (lons, lats) of UL, LL, UR, LR

A real ``nansat.Nansat`` satisfies the same contract and can be passed instead.
``ArrayNansat`` maps pixels to "lon/lat" with a plain affine transform - the file-less
recipe of the reference's examples/drift_from_arrays.ipynb uses a fake Mercator domain for
the same purpose.
"""
import numpy as np


class ArrayNansat(object):
    """uint8 image + affine georeference  [lon, lat] = origin + A @ [col, row]."""

    def __init__(self, image, origin=(0.0, 0.0), matrix=((1.0, 0.0), (0.0, 1.0)), dst_scale=None):
        image = np.asarray(image)
        if image.ndim != 2 or image.dtype != np.uint8:
            raise TypeError('image must be a 2-D uint8 array (0 = invalid, reference lib.py:52-57)')
        self.image = image
        self.origin = np.asarray(origin, dtype=np.float64)
        self.matrix = np.asarray(matrix, dtype=np.float64).reshape(2, 2)
        self.inverse = np.linalg.inv(self.matrix)
        # destination-SRS stand-in: coordinates = lon/lat * dst_scale (None -> lon/lat themselves)
        self.dst_scale = dst_scale

    @classmethod
    def rotated(cls, image, angle_deg=0.0, scale=1.0, origin=(0.0, 0.0), **kw):
        """Georeference rotated by angle_deg about the origin (gives a non-zero alpha0)."""
        a = np.radians(angle_deg)
        m = scale * np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return cls(image, origin=origin, matrix=m, **kw)

    def __getitem__(self, band):
        if band != 1:
            raise KeyError('ArrayNansat holds one band (1)')
        return self.image

    def shape(self):
        return self.image.shape

    def transform_points(self, x, y, DstToSrc=0, dst_srs=None):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        if DstToSrc:
            dx, dy = x - self.origin[0], y - self.origin[1]
            col = self.inverse[0, 0] * dx + self.inverse[0, 1] * dy
            row = self.inverse[1, 0] * dx + self.inverse[1, 1] * dy
            return col, row
        lon = self.origin[0] + self.matrix[0, 0] * x + self.matrix[0, 1] * y
        lat = self.origin[1] + self.matrix[1, 0] * x + self.matrix[1, 1] * y
        if dst_srs is not None and self.dst_scale is not None:
            return lon * self.dst_scale, lat * self.dst_scale
        return lon, lat

    def get_corners(self):
        rows, cols = self.image.shape
        c = np.array([0.0, 0.0, cols, cols])
        r = np.array([0.0, rows, 0.0, rows])
        return self.transform_points(c, r, 0)
