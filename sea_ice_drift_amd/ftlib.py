"""Feature-tracking matcher, GPU-backed: host mirror of the reference's ``ftlib.get_match_coords``
(``/root/reference/sea_ice_drift/ftlib.py:64-116``).

``cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(descriptors1, descriptors2, k=2)`` (ftlib.py:92-99) runs as a HIP
kernel behind ``sid_ft_knn2`` (include/sid_ft.h); the Lowe ratio filter (ftlib.py:101-116) stays on the host.
Key-point detection (ORB, ftlib.py:26-61) is OpenCV's and is not part of this package: key points come in as
``cv2.KeyPoint``-like objects (anything with ``.pt``) or as an ``(N, 2)`` array of ``(x, y)``.
There is no CPU fallback: without the HIP library the call raises.
"""
import numpy as np

from . import _capi
from .lib import get_displacement_km, get_speed_ms, interpolation_poly


def _points(key_points):
    if isinstance(key_points, np.ndarray):
        return np.asarray(key_points, dtype=np.float64).reshape(-1, 2)
    return np.array([kp.pt for kp in key_points], dtype=np.float64).reshape(-1, 2)


def _get_matches(descriptors1, descriptors2, device=0, verbose=False):
    """ftlib.py:92-99: for every descriptor of image 1 the two nearest (Hamming) descriptors of image 2.
    Returns (idx [n1,2], dist [n1,2]) instead of a list of DMatch pairs."""
    d2 = np.asarray(descriptors2)
    if d2.reshape(-1, 32).shape[0] < 2:
        # the reference unpacks ``for m, n in matches`` (ftlib.py:104) and fails the same way
        raise ValueError('need at least 2 train descriptors for k=2 matching')
    return _capi.ft_knn2(descriptors1, descriptors2, device=device)


def _filter_matches(matches, ratio_test, keyPoints1, keyPoints2, verbose=False):
    """ftlib.py:101-116: Lowe's ratio test ``m.distance < ratio_test * n.distance`` and the coordinates of
    the surviving pairs, in query order."""
    idx, dist = matches
    good = dist[:, 0].astype(np.float64) < float(ratio_test) * dist[:, 1].astype(np.float64)
    if verbose:
        print('Ratio test %f found %d keypoints' % (ratio_test, int(good.sum())))
    p1, p2 = _points(keyPoints1), _points(keyPoints2)
    q = np.nonzero(good)[0]
    t = idx[q, 0]
    return p1[q, 0], p1[q, 1], p2[t, 0], p2[t, 1]


def get_match_coords(keyPoints1, descriptors1, keyPoints2, descriptors2, matcher=None, norm=None,
                     ratio_test=0.7, verbose=False, device=0, **kwargs):
    """Signature of ftlib.py:64-90.  ``matcher`` / ``norm`` are accepted for compatibility; only the default
    (brute force, Hamming) exists on the device and anything else raises ``NotImplementedError``."""
    if matcher is not None or norm is not None:
        raise NotImplementedError('only the brute-force Hamming matcher (the reference default) is implemented')
    matches = _get_matches(descriptors1, descriptors2, device=device, verbose=verbose)
    return _filter_matches(matches, ratio_test, keyPoints1, keyPoints2, verbose)


def find_key_points(image, edgeThreshold=34, nFeatures=100000, nLevels=7, patchSize=34, verbose=False, **kwargs):
    """ORB key points and descriptors (reference ftlib.py:26-61).  ORB is OpenCV's: this needs ``cv2``
    (not part of this package, absent from the build image) - or pass ``find_key_points=`` to
    ``feature_tracking`` with your own detector returning (key points, uint8 [N, 32] descriptors)."""
    try:
        import cv2
    except ImportError:
        raise NotImplementedError('ORB detection needs OpenCV (cv2); pass find_key_points= to feature_tracking '
                                  'or match precomputed descriptors with get_match_coords')
    detector = cv2.ORB_create()                                           # pragma: no cover
    detector.setEdgeThreshold(edgeThreshold)                              # pragma: no cover
    detector.setMaxFeatures(nFeatures)                                    # pragma: no cover
    detector.setNLevels(nLevels)                                          # pragma: no cover
    detector.setPatchSize(patchSize)                                      # pragma: no cover
    keyPoints, descriptors = detector.detectAndCompute(image, None)       # pragma: no cover
    if verbose:                                                           # pragma: no cover
        print('Key points found: %d' % len(keyPoints))
    return keyPoints, descriptors                                         # pragma: no cover


def domain_filter(n, keyPoints, descr, domain, domainMargin=0, verbose=False, **kwargs):
    """Key points of ``n`` that fall inside ``domain`` (reference ftlib.py:118-142)."""
    pts = _points(keyPoints)
    lon, lat = n.transform_points(pts[:, 0], pts[:, 1], 0)
    colsD, rowsD = domain.transform_points(lon, lat, 1)
    gpi = ((colsD >= 0 + domainMargin) *
           (rowsD >= 0 + domainMargin) *
           (colsD <= domain.shape()[1] - domainMargin) *
           (rowsD <= domain.shape()[0] - domainMargin))
    if verbose:
        print('Domain filter: %d -> %d' % (len(pts), len(gpi[gpi])))
    kept = pts[gpi] if isinstance(keyPoints, np.ndarray) else list(np.array(keyPoints, dtype=object)[gpi])
    return kept, np.asarray(descr)[gpi]


def max_drift_filter(n1, x1, y1, n2, x2, y2, max_speed=0.5, max_drift=None, verbose=False, **kwargs):
    """Drop vectors faster than ``max_speed`` m/s (images with time stamps) or longer than ``max_drift``
    metres (reference ftlib.py:144-206)."""
    try:
        n1.time_coverage_start
        n2.time_coverage_start
    except (ValueError, AttributeError):
        data_has_timestamp = False
    else:
        data_has_timestamp = True
    if data_has_timestamp:
        gpi = get_speed_ms(n1, x1, y1, n2, x2, y2) <= max_speed
    elif max_drift is not None:
        gpi = 1000. * get_displacement_km(n1, x1, y1, n2, x2, y2) <= max_drift
    else:
        raise ValueError('Input data does not have time stamp, and <max_drift> is not set: provide max_drift, '
                         'the maximum allowed ice displacement between the images in metres')
    if verbose:
        print('MaxDrift filter: %d -> %d' % (len(x1), len(gpi[gpi])))
    return x1[gpi], y1[gpi], x2[gpi], y2[gpi]


def lstsq_filter(x1, y1, x2, y2, psi=200, order=2, verbose=False, **kwargs):
    """Drop vectors further than ``psi`` pixels from a least-squares polynomial model (reference ftlib.py:208-238)."""
    if len(x1) == 0:
        return tuple(map(np.array, [[], [], [], []]))
    x2sim, y2sim = interpolation_poly(x1, y1, x2, y2, x1, y1, order=order)
    err = np.hypot(x2 - x2sim, y2 - y2sim)
    gpi = err < psi
    if verbose:
        print('LSTSQ filter: %d -> %d' % (len(x1), len(gpi[gpi])))
    return x1[gpi], y1[gpi], x2[gpi], y2[gpi]


def feature_tracking(n1, n2, find_key_points=find_key_points, **kwargs):
    """The reference's feature-tracking driver (ftlib.py:241-285): key points -> domain filter -> Hamming
    matching on the GPU + Lowe filter -> drift and least-squares filters; returns x1, y1, x2, y2 in pixels.
    ``find_key_points(image, **kwargs) -> (key points, descriptors)`` defaults to OpenCV's ORB."""
    kp1, descr1 = find_key_points(n1[1], **kwargs)
    kp2, descr2 = find_key_points(n2[1], **kwargs)
    if len(kp1) < 2 or len(kp2) < 2:
        return (np.array([]),) * 4
    kp1, descr1 = domain_filter(n1, kp1, descr1, n2, **kwargs)
    if len(kp1) < 2:
        return (np.array([]),) * 4
    kp2, descr2 = domain_filter(n2, kp2, descr2, n1, **kwargs)
    if len(kp2) < 2:
        return (np.array([]),) * 4
    x1, y1, x2, y2 = get_match_coords(kp1, descr1, kp2, descr2, **kwargs)
    x1, y1, x2, y2 = max_drift_filter(n1, x1, y1, n2, x2, y2, **kwargs)
    x1, y1, x2, y2 = lstsq_filter(x1, y1, x2, y2, **kwargs)
    return x1, y1, x2, y2
