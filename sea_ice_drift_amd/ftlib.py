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
