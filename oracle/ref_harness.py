"""Import the reference's own pmlib in the BUILD CONTAINER (never on the GPU box).

TEST INFRASTRUCTURE ONLY.  /root/reference is read-only, public and untrusted; it
is imported from where it lies, nothing is copied.  cv2, osgeo (GDAL) and nansat
are not installed in this image, so stub modules are registered first (SURVEY.md
Appendix B).  The stub ``cv2.matchTemplate`` raises if it is ever called: callers
must pass ``template_matcher=oracle.pm_oracle.match_template`` through the
reference's own plug point (pmlib.py:119-120) - that is the one piece of the path
whose arithmetic is OpenCV's and therefore restated (parity unpinned there).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('SID_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'sea_ice_drift'))


class _NSR(object):
    """Stand-in for nansat.NSR: remembers the srs string only."""
    def __init__(self, srs=None):
        self.srs = srs
        self.wkt = srs


def _install_stubs():
    if 'cv2' not in sys.modules:
        cv2 = types.ModuleType('cv2')
        cv2.TM_CCOEFF_NORMED = 5
        cv2.NORM_HAMMING = 6
        cv2.__version__ = '4.0.0-stub'

        def _absent(*a, **k):
            raise RuntimeError('cv2 is a stub in this image: inject template_matcher=')
        cv2.matchTemplate = _absent
        cv2.BFMatcher = _absent
        cv2.ORB_create = _absent
        sys.modules['cv2'] = cv2
    if 'osgeo' not in sys.modules:
        osgeo = types.ModuleType('osgeo')
        gdal = types.ModuleType('osgeo.gdal')
        osgeo.gdal = gdal
        sys.modules['osgeo'] = osgeo
        sys.modules['osgeo.gdal'] = gdal
    if 'nansat' not in sys.modules:
        nansat = types.ModuleType('nansat')
        nansat.NSR = _NSR
        nansat.Nansat = type('Nansat', (object,), {})
        nansat.Domain = type('Domain', (object,), {})
        sys.modules['nansat'] = nansat


def load():
    """Return the reference's (pmlib, lib) modules."""
    if not available():
        raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
    import matplotlib
    matplotlib.use('Agg')
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    pmlib = importlib.import_module('sea_ice_drift.pmlib')
    lib = importlib.import_module('sea_ice_drift.lib')
    return pmlib, lib


def load_ftlib():
    """The reference's ftlib module (ORB itself needs the real cv2: inject key points, see make_golden g7)."""
    load()
    import importlib
    return importlib.import_module('sea_ice_drift.ftlib')
