"""``SeaIceDrift`` with the reference's public methods (reference seaicedrift.py:23-88).

``get_drift_PM`` is the hot path; ``get_drift_FT`` runs the key-point detector (``sea_ice_drift_amd.orb``, an
ORB-family detector behind the reference's interface - OpenCV is not used) and the Hamming matcher on the GPU and
the reference's filters on the host; a ``find_key_points=`` callable replaces the detector.  The
constructor takes two Nansat-like objects (``sea_ice_drift_amd.domain.ArrayNansat`` or real
``nansat.Nansat``); opening Sentinel-1 files (reference lib.get_n, lib.py:256-340) needs nansat/GDAL and
is outside the scope of this package - passing file names raises with a pointer to what to pass instead.
"""
from sea_ice_drift_amd.ftlib import feature_tracking
from sea_ice_drift_amd.lib import get_drift_vectors, prefetch_triangulation
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
        """Same returns as the reference (seaicedrift.py:42-60): u, v, lon1, lat1, lon2, lat2 of the matched
        key points.  Detection and matching run on the GPU (``sea_ice_drift_amd.orb``, ``include/sid_ft.h``), the
        filters on the host; ``find_key_points=`` takes another detector (see ``ftlib.feature_tracking``).  ``nsr=``
        selects the units of u, v as in the reference (``lib.get_drift_vectors``)."""
        x1, y1, x2, y2 = feature_tracking(self.n1, self.n2, **kwargs)
        kwargs.pop('find_key_points', None)
        out = get_drift_vectors(self.n1, x1, y1, self.n2, x2, y2, **kwargs)
        self._prefetch_first_guess(out[2], out[3])
        return out

    def _prefetch_first_guess(self, lon1, lat1):
        """The usual next call is ``get_drift_PM(lons, lats, lon1, lat1, lon2, lat2)`` with these very vectors; its prelude
        triangulates the key points of image 1 carried into the pixel space of image 2 (pmlib.py:280-288 through
        lib.interpolation_near) - the longest step of the call.  The same arithmetic on the same numbers, started now on a
        worker thread: whatever the caller does before ``get_drift_PM`` runs beside the triangulation, and the call picks
        the result up when its points are bit-identical (otherwise it triangulates as before).  Results are unchanged."""
        try:
            import os
            import numpy as np
            if len(lon1) < 4 or os.environ.get('SID_NO_TRI_PREFETCH'):   # (the switch: A/B runs)
                return
            x1, y1 = self.n1.transform_points(lon1, lat1, 1)             # get_drift_PM (seaicedrift.py:85)
            lon, lat = self.n1.transform_points(x1, y1)                  # prepare_first_guess (pmlib.py:280-282)
            c1n2, r1n2 = self.n2.transform_points(lon, lat, 1)
            prefetch_triangulation(np.array([r1n2, c1n2]).T)             # interpolation_near's point order (lib.py:195)
        except Exception:                                                # noqa: BLE001 - a convenience, never an error
            pass

    def get_drift_PM(self, lons, lats, lon1, lat1, lon2, lat2, **kwargs):
        """Same arguments and returns as the reference (seaicedrift.py:62-88):
        u, v, a, r, h, lon2_dst, lat2_dst on the (lons, lats) grid."""
        x1, y1 = self.n1.transform_points(lon1, lat1, 1)
        x2, y2 = self.n2.transform_points(lon2, lat2, 1)
        return pattern_matching(lons, lats, self.n1, x1, y1, self.n2, x2, y2, **kwargs)
