"""Feature tracking (the first guess of pattern matching) with the GPU matcher and detector.

Public surface = what a user of the reference's ``sea_ice_drift.ftlib`` calls (ftlib.py:26-285):
``find_key_points``, ``get_match_coords``, ``domain_filter``, ``max_drift_filter``, ``lstsq_filter`` and
``feature_tracking``, with the same arguments and the same returns.  Underneath, the module is organised
differently from the reference: every stage is a pure function from arrays to a boolean *keep mask*
(``mask_inside_domain``, ``mask_lowe_ratio``, ``mask_drift_limit``, ``mask_model_residual``) and
``track`` composes them over one ``Matches`` record; the public names are thin adaptors over those.

* Hamming k=2 nearest neighbours (``cv2.BFMatcher(NORM_HAMMING).knnMatch``, ftlib.py:92-99) run on the GPU
  behind ``sid_ft_knn2`` (include/sid_ft.h).
* Key points: ``find_key_points`` uses the package's own ORB-interface detector on the GPU
  (``sea_ice_drift_amd.orb``, include/sid_orb.h) - OpenCV is not part of this package.  Any callable
  ``f(image, **kw) -> (key points, uint8 [N,32] descriptors)`` can be passed as ``find_key_points=``.
  Key points are ``(N, 2)`` arrays of ``(x, y)`` or sequences of objects with a ``.pt`` (``cv2.KeyPoint``).

There is no CPU fallback: without the HIP library the matcher and the detector raise.
"""
import collections

import numpy as np

from . import _capi
from .lib import fit_polynomial_map, great_circle_km

Matches = collections.namedtuple('Matches', 'x1 y1 x2 y2')
_EMPTY = Matches(*(np.array([]),) * 4)


# --------------------------------------------------------------------------- key-point containers
def _xy(key_points):
    """(N, 2) float64 array of (x, y) from an array or from cv2.KeyPoint-like objects."""
    if isinstance(key_points, np.ndarray):
        return key_points.astype(np.float64, copy=False).reshape(-1, 2)
    return np.array([k.pt for k in key_points], dtype=np.float64).reshape(-1, 2)


def _take(key_points, keep):
    """Subset of a key-point collection, preserving its kind (array stays array, objects stay a list)."""
    if isinstance(key_points, np.ndarray):
        return _xy(key_points)[keep]
    return [k for k, ok in zip(key_points, keep) if ok]


def _say(verbose, fmt, *args):
    if verbose:
        print(fmt % args)


# --------------------------------------------------------------------------- stages as keep masks
def mask_inside_domain(n, xy, domain, margin=0):
    """Key points of image ``n`` whose location falls on ``domain``'s raster, ``margin`` pixels in from its
    edges on every side (both bounds inclusive, as ftlib.py:134-137 has them)."""
    lon, lat = n.transform_points(xy[:, 0], xy[:, 1], 0)
    col, row = domain.transform_points(lon, lat, 1)
    rows, cols = domain.shape()[0], domain.shape()[1]
    keep = (col >= 0 + margin) & (row >= 0 + margin)
    keep &= (col <= cols - margin) & (row <= rows - margin)
    return keep


def mask_lowe_ratio(dist, ratio):
    """Lowe's test on the two nearest Hamming distances of every query (ftlib.py:104-106): the best match
    must be closer than ``ratio`` times the runner-up."""
    d = np.asarray(dist, dtype=np.float64)
    return d[:, 0] < float(ratio) * d[:, 1]


def _has_time(n):
    try:
        n.time_coverage_start
    except (ValueError, AttributeError):
        return False
    return True


def mask_drift_limit(n1, n2, m, max_speed=0.5, max_drift=None):
    """Vectors no faster than ``max_speed`` m/s when both images carry a time stamp, else no longer than
    ``max_drift`` metres; neither available is an error (ftlib.py:171-198)."""
    km = great_circle_km(n1, m.x1, m.y1, n2, m.x2, m.y2)
    if _has_time(n1) and _has_time(n2):
        seconds = (n2.time_coverage_start - n1.time_coverage_start).total_seconds()
        return 1000. * km / abs(seconds) <= max_speed
    if max_drift is None:
        raise ValueError('the images carry no time stamp and max_drift is not set: give max_drift, the largest '
                         'plausible displacement between the two images in metres')
    return 1000. * km <= max_drift


def mask_model_residual(m, psi=200, order=2):
    """Vectors whose end point lies closer than ``psi`` pixels to a least-squares polynomial map
    start -> end fitted to all of them (ftlib.py:220-230)."""
    fx, fy = fit_polynomial_map(m.x1, m.y1, m.x2, m.y2, order)(m.x1, m.y1)
    return np.hypot(m.x2 - fx, m.y2 - fy) < psi


def _subset(m, keep):
    return Matches(m.x1[keep], m.y1[keep], m.x2[keep], m.y2[keep])


# --------------------------------------------------------------------------- matcher
def _get_matches(descriptors1, descriptors2, device=0, verbose=False):
    """For every descriptor of image 1 the two Hamming-nearest descriptors of image 2, as index and distance
    arrays [n1, 2] (the GPU form of ftlib.py:92-99's list of DMatch pairs)."""
    if np.asarray(descriptors2).reshape(-1, 32).shape[0] < 2:
        # k = 2 needs two candidates; the reference fails at this point too (``for m, n in matches``)
        raise ValueError('need at least 2 train descriptors for k=2 matching')
    return _capi.ft_knn2(descriptors1, descriptors2, device=device)


def match(kp1, descr1, kp2, descr2, ratio_test=0.7, device=0, verbose=False):
    """Ratio-tested correspondences between two key-point sets -> Matches (query order)."""
    idx, dist = _get_matches(descr1, descr2, device=device, verbose=verbose)
    keep = mask_lowe_ratio(dist, ratio_test)
    _say(verbose, 'Ratio test %f found %d keypoints', ratio_test, int(keep.sum()))
    a, b = _xy(kp1)[keep], _xy(kp2)[idx[keep, 0]]
    return Matches(a[:, 0], a[:, 1], b[:, 0], b[:, 1])


# --------------------------------------------------------------------------- the pipeline
def track(n1, n2, detector, domainMargin=0, ratio_test=0.7, max_speed=0.5, max_drift=None, psi=200, order=2,
          device=0, verbose=False, concurrent_detection=False, **detector_kwargs):
    """Detector -> domain masks -> matcher + ratio mask -> drift mask -> model mask (ftlib.py:259-281).
    Fewer than two key points on either side at any stage ends with four empty arrays.
    ``concurrent_detection``: the two images on two host threads (the package's own detector only: its per-level host
    work - candidate selection, sorting - then runs beside the other image's kernels; the results do not depend on it)."""
    def detect(n):
        kp, descr = detector(n[1], **detector_kwargs)
        return kp, np.asarray(descr)
    if concurrent_detection:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=2) as pool:
            sets = list(pool.map(detect, (n1, n2)))
    else:
        sets = [detect(n1)]
        if len(sets[0][0]) >= 2:
            sets.append(detect(n2))
    if any(len(kp) < 2 for kp, _ in sets):
        return _EMPTY
    for this, other, (kp, descr) in ((0, n2, sets[0]), (1, n1, sets[1])):
        keep = mask_inside_domain((n1, n2)[this], _xy(kp), other, domainMargin)
        _say(verbose, 'Domain filter: %d -> %d', len(keep), int(keep.sum()))
        if keep.sum() < 2:
            return _EMPTY
        sets[this] = (_take(kp, keep), descr[keep])
    m = match(sets[0][0], sets[0][1], sets[1][0], sets[1][1], ratio_test=ratio_test, device=device, verbose=verbose)
    keep = mask_drift_limit(n1, n2, m, max_speed=max_speed, max_drift=max_drift)
    _say(verbose, 'MaxDrift filter: %d -> %d', len(keep), int(keep.sum()))
    m = _subset(m, keep)
    if len(m.x1) == 0:
        return _EMPTY
    keep = mask_model_residual(m, psi=psi, order=order)
    _say(verbose, 'LSTSQ filter: %d -> %d', len(keep), int(keep.sum()))
    return _subset(m, keep)


# --------------------------------------------------------------------------- the reference's public names
def find_key_points(image, edgeThreshold=34, nFeatures=100000, nLevels=7, patchSize=34, verbose=False, device=0, **kwargs):
    """Key points and 256-bit descriptors of a uint8 image - the interface of ftlib.py:26-61
    (``cv2.ORB_create`` with ``edgeThreshold=34, nFeatures=100000, nLevels=7, patchSize=34``) served by the
    package's own detector on the GPU: FAST-9 corners, Harris ranking, 7-level pyramid, intensity-centroid
    orientation, steered BRIEF-256 (``sea_ice_drift_amd.orb``).  Returns ``((N, 2) float64 (x, y), uint8 [N, 32])``.
    OpenCV's exact key points are not reproduced (cv2 is not available to pin them); the output feeds the
    same matcher and filters."""
    from . import orb
    xy, descr = orb.detect_and_compute(image, edge_threshold=edgeThreshold, n_features=nFeatures, n_levels=nLevels,
                                       patch_size=patchSize, device=device)
    _say(verbose, 'Key points found: %d', len(xy))
    return xy, descr


def get_match_coords(keyPoints1, descriptors1, keyPoints2, descriptors2, matcher=None, norm=None,
                     ratio_test=0.7, verbose=False, device=0, **kwargs):
    """Signature of ftlib.py:64-90 -> x1, y1, x2, y2.  Only the reference's default matcher (brute force,
    Hamming) exists on the device: passing ``matcher`` / ``norm`` raises ``NotImplementedError``."""
    if matcher is not None or norm is not None:
        raise NotImplementedError('only the brute-force Hamming matcher (the reference default) is implemented')
    return tuple(match(keyPoints1, descriptors1, keyPoints2, descriptors2, ratio_test=ratio_test, device=device,
                       verbose=verbose))


def domain_filter(n, keyPoints, descr, domain, domainMargin=0, verbose=False, **kwargs):
    """ftlib.py:118-142 -> (key points, descriptors) that fall on ``domain``."""
    keep = mask_inside_domain(n, _xy(keyPoints), domain, domainMargin)
    _say(verbose, 'Domain filter: %d -> %d', len(keep), int(keep.sum()))
    return _take(keyPoints, keep), np.asarray(descr)[keep]


def max_drift_filter(n1, x1, y1, n2, x2, y2, max_speed=0.5, max_drift=None, verbose=False, **kwargs):
    """ftlib.py:144-206 -> x1, y1, x2, y2 within the speed / displacement limit."""
    m = Matches(*(np.asarray(v) for v in (x1, y1, x2, y2)))
    keep = mask_drift_limit(n1, n2, m, max_speed=max_speed, max_drift=max_drift)
    _say(verbose, 'MaxDrift filter: %d -> %d', len(keep), int(keep.sum()))
    return tuple(_subset(m, keep))


def lstsq_filter(x1, y1, x2, y2, psi=200, order=2, verbose=False, **kwargs):
    """ftlib.py:208-238 -> x1, y1, x2, y2 consistent with a polynomial displacement model."""
    m = Matches(*(np.asarray(v) for v in (x1, y1, x2, y2)))
    if len(m.x1) == 0:
        return tuple(_EMPTY)
    keep = mask_model_residual(m, psi=psi, order=order)
    _say(verbose, 'LSTSQ filter: %d -> %d', len(keep), int(keep.sum()))
    return tuple(_subset(m, keep))


def feature_tracking(n1, n2, find_key_points=find_key_points, **kwargs):
    """ftlib.py:241-285: x1, y1, x2, y2 (pixels) of matched and filtered key points of two images.
    ``kwargs`` are shared by the detector and the filters, as in the reference."""
    own = ('domainMargin', 'ratio_test', 'max_speed', 'max_drift', 'psi', 'order', 'device', 'verbose')
    stage_kw = {k: kwargs[k] for k in own if k in kwargs}
    det_kw = {k: v for k, v in kwargs.items() if k not in own or k in ('verbose', 'device')}
    if find_key_points is not globals()['find_key_points']:
        det_kw = dict(kwargs)                       # a user detector sees everything, like the reference's call
    return tuple(track(n1, n2, lambda image, **kw: find_key_points(image, **det_kw),
                       concurrent_detection=find_key_points is globals()['find_key_points'], **stage_kw))
