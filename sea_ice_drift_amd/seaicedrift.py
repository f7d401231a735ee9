"""``SeaIceDrift`` with the reference's public methods (reference seaicedrift.py:23-88).

Only ``get_drift_PM`` is backed by this package (it is the hot path).  The constructor
takes two Nansat-like objects (``sea_ice_drift_amd.domain.ArrayNansat`` or real
``nansat.Nansat``); opening Sentinel-1 files (reference lib.get_n, lib.py:256-340) needs
nansat/GDAL and is outside the scope of this package, as is feature tracking
(``get_drift_FT``: OpenCV ORB + BFMatcher, reference ftlib.py) - both raise with a pointer
to what to pass instead.
"""
from sea_ice_drift_amd.pmlib import pattern_matching


class SeaIceDrift(object):
    """Retrieve sea ice drift with pattern matching on an MI355X."""

    def __init__(self, n1, n2, **kwargs):
        for n in (n1, n2):
            if isinstance(n, str):
                raise NotImplementedError(
                    'file staging (reference lib.get_n) is out of scope here: open the files with '
                    'sea_ice_drift.lib.get_n / nansat yourself and pass the two Nansat objects, '
                    'or wrap uint8 arrays in sea_ice_drift_amd.domain.ArrayNansat')
        self.n1 = n1
        self.n2 = n2

    def get_drift_FT(self, **kwargs):
        raise NotImplementedError(
            'feature tracking (ORB + BFMatcher, reference ftlib.py) is not part of the PM hot path; '
            'run sea_ice_drift.ftlib.feature_tracking (OpenCV) and pass its keypoints to get_drift_PM')

    def get_drift_PM(self, lons, lats, lon1, lat1, lon2, lat2, **kwargs):
        """Same arguments and returns as the reference (seaicedrift.py:62-88):
        u, v, a, r, h, lon2_dst, lat2_dst on the (lons, lats) grid."""
        x1, y1 = self.n1.transform_points(lon1, lat1, 1)
        x2, y2 = self.n2.transform_points(lon2, lat2, 1)
        return pattern_matching(lons, lats, self.n1, x1, y1, self.n2, x2, y2, **kwargs)
