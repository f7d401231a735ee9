"""Key points and 256-bit descriptors on the GPU: the detector behind ``ftlib.find_key_points``.

The reference calls OpenCV's ORB (ftlib.py:26-61: ``edgeThreshold=34, nFeatures=100000, nLevels=7, patchSize=34``).
OpenCV is not part of this package, so this is the package's own ORB-family detector with an all-integer
specification (include/sid_orb.h; restated in NumPy by oracle/orb_oracle.py): pyramid, FAST-9 corners with 3x3
non-maximum suppression, Harris ranking, intensity-centroid orientation quantised to 32 directions, 256 intensity
comparisons on a smoothed image through a pre-rotated pair pattern.  Same interface as the reference's function -
uint8 image in, key points + uint8 [N, 32] descriptors out - but not OpenCV's key points (parity unpinned).
The tables below (directions, comparison pattern) are part of the specification and are shared with the oracle.
"""
import ctypes as C
import functools

import numpy as np

N_DIRS = 32
PATTERN_RADIUS = 13
PATTERN_SEED = 20240531


@functools.lru_cache(maxsize=None)
def _direction_table():
    th = 2.0 * np.pi * np.arange(N_DIRS) / N_DIRS
    out = np.stack([np.rint(16384.0 * np.cos(th)), np.rint(16384.0 * np.sin(th))], axis=1).astype(np.int32)
    out.setflags(write=False)
    return out


def direction_table():
    """[32, 2] int32: round(2^14 cos), round(2^14 sin) of the directions 2 pi b / 32 (built once, read-only)."""
    return _direction_table()


def base_pattern():
    """[256, 4] int: (ax, ay, bx, by) - seeded isotropic Gaussian point pairs inside the disc of radius 13."""
    rng = np.random.Generator(np.random.PCG64(PATTERN_SEED))
    pts = []
    while len(pts) < 512:
        p = np.rint(rng.normal(0.0, PATTERN_RADIUS / 2.0, 2)).astype(np.int64)
        if p[0] * p[0] + p[1] * p[1] <= (PATTERN_RADIUS - 1) ** 2:
            pts.append(p)
    pts = np.array(pts).reshape(256, 4)
    same = (pts[:, 0] == pts[:, 2]) & (pts[:, 1] == pts[:, 3])
    pts[same, 2] = -pts[same, 2] - 1                          # a comparison of a point with itself carries no bit
    return pts


@functools.lru_cache(maxsize=None)
def _rotated_pattern():
    base = base_pattern().astype(np.float64)
    th = 2.0 * np.pi * np.arange(N_DIRS) / N_DIRS
    out = np.empty((N_DIRS, 256, 4), dtype=np.int8)
    for b, t in enumerate(th):
        c, s = np.cos(t), np.sin(t)
        for k in (0, 2):
            x, y = base[:, k], base[:, k + 1]
            out[b, :, k] = np.rint(x * c - y * s).astype(np.int8)
            out[b, :, k + 1] = np.rint(x * s + y * c).astype(np.int8)
    out.setflags(write=False)
    return out


def rotated_pattern():
    """[32, 256, 4] int8: the base pattern rotated to every direction, rounded to pixels (built once, read-only)."""
    return _rotated_pattern()


class OrbParams(C.Structure):
    _fields_ = [('edge_threshold', C.c_int32), ('n_features', C.c_int32), ('n_levels', C.c_int32),
                ('patch_size', C.c_int32), ('fast_threshold', C.c_int32), ('scale_factor', C.c_float)]


def detect_and_compute(image, edge_threshold=34, n_features=100000, n_levels=7, patch_size=34, fast_threshold=20,
                       scale_factor=1.2, device=0, full=False):
    """uint8 image -> ((N, 2) float64 key points (x, y), uint8 [N, 32] descriptors); with ``full`` also the
    int32 [N, 4] (level x, level y, level, direction) and the int64 Harris responses."""
    from . import _capi
    img = np.asarray(image)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise TypeError('a 2-D uint8 image is expected (reference contract: lib.py:27-59)')
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    L = _capi.lib()
    n_max = int(max(n_features, 0))
    xy = np.empty((n_max, 2), dtype=np.float32)
    meta = np.empty((n_max, 4), dtype=np.int32)
    resp = np.empty(n_max, dtype=np.int64)
    desc = np.empty((n_max, 32), dtype=np.uint8)
    n = C.c_int64(0)
    p = OrbParams(int(edge_threshold), int(n_features), int(n_levels), int(patch_size), int(fast_threshold), float(scale_factor))
    pat = np.ascontiguousarray(rotated_pattern())
    dirs = np.ascontiguousarray(direction_table())
    rc = L.sid_orb_detect(int(device), img.ctypes.data_as(C.POINTER(C.c_uint8)), img.shape[0], img.shape[1], img.strides[0],
                          C.byref(p), pat.ctypes.data_as(C.POINTER(C.c_int8)), dirs.ctypes.data_as(C.POINTER(C.c_int32)),
                          xy.ctypes.data_as(C.POINTER(C.c_float)), meta.ctypes.data_as(C.POINTER(C.c_int32)),
                          resp.ctypes.data_as(C.POINTER(C.c_int64)), desc.ctypes.data_as(C.POINTER(C.c_uint8)), n_max, C.byref(n))
    if rc != 0:
        raise _capi.SidPmError(rc, L.sid_orb_last_error().decode())
    k = int(n.value)
    if full:
        return xy[:k].astype(np.float64), desc[:k].copy(), meta[:k].copy(), resp[:k].copy()
    return xy[:k].astype(np.float64), desc[:k].copy()
