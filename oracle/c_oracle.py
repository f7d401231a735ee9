"""ctypes binding of oracle/pm_oracle.c (TEST INFRASTRUCTURE ONLY - see that file)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libsid_pm_oracle.so')
_lib = None

_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def build(force=False):
    """Compile the C oracle (gcc via oracle/Makefile)."""
    src = os.path.join(_HERE, 'pm_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.sid_oracle_get_template.restype = C.c_int
        L.sid_oracle_get_template.argtypes = [_u8p, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                                              C.c_double, _f64p, C.c_int, _u8p]
        L.sid_oracle_get_template1.restype = C.c_int
        L.sid_oracle_get_template1.argtypes = L.sid_oracle_get_template.argtypes
        L.sid_oracle_match_template.restype = C.c_int
        L.sid_oracle_match_template.argtypes = [_u8p, C.c_int, C.c_int, C.c_int64, _u8p, C.c_int, _f32p]
        L.sid_oracle_hessian.restype = C.c_int
        L.sid_oracle_hessian.argtypes = [_f32p, C.c_int, C.c_int, C.c_uint, _f32p]
        L.sid_oracle_pm_batch.restype = C.c_int
        L.sid_oracle_pm_batch.argtypes = (
            [_u8p, C.c_int64, C.c_int64, C.c_int64] * 2 + [_f64p] * 5 +
            [C.c_int64, C.c_int, C.c_double, _f64p, _f64p, C.c_int, C.c_uint, C.c_int, _f64p, _i32p])
        L.sid_oracle_pm_batch_gap.restype = C.c_int
        L.sid_oracle_pm_batch_gap.argtypes = L.sid_oracle_pm_batch.argtypes + [_f32p]
        L.sid_oracle_max_threads.restype = C.c_int
        L.sid_oracle_spline_coefficients.restype = C.c_int
        L.sid_oracle_spline_coefficients.argtypes = [_u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int, _f64p]
        L.sid_oracle_get_template_spline.restype = C.c_int
        L.sid_oracle_get_template_spline.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_double, C.c_double, _f64p, C.c_int, C.c_int, _u8p]
        L.sid_oracle_rotate_and_match.restype = C.c_int
        L.sid_oracle_rotate_and_match.argtypes = ([_u8p, C.c_int64, C.c_int64, C.c_int64] * 2 +
                                                  [C.c_double, C.c_double, C.c_int, _f64p, _f64p, C.c_int, C.c_uint, _f64p, _i32p, _f32p, _u8p])
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_f64p)


def spline_coefficients(img, order):
    """scipy's spline_filter(img, order, output=float64) restated (pm_oracle.c sid_oracle_spline_coefficients)."""
    img, pi = _u8(img)
    coef = np.empty(img.shape, dtype=np.float64)
    if lib().sid_oracle_spline_coefficients(pi, img.shape[0], img.shape[1], img.strides[0], int(order), coef.ctypes.data_as(_f64p)):
        raise ValueError('order 2..5')
    return coef


def get_template(img, c, r, rot, s, rot_order=0, coeffs=None):
    img, pi = _u8(img)
    rot, pr = _f64(rot)
    out = np.empty((s, s), dtype=np.uint8)
    if rot_order >= 2:
        coeffs = spline_coefficients(img, rot_order) if coeffs is None else np.ascontiguousarray(coeffs, dtype=np.float64)
        lib().sid_oracle_get_template_spline(coeffs.ctypes.data_as(_f64p), img.shape[0], img.shape[1], float(c), float(r), pr, int(s),
                                             int(rot_order), out.ctypes.data_as(_u8p))
        return out
    (lib().sid_oracle_get_template1 if rot_order == 1 else lib().sid_oracle_get_template)(pi, img.shape[0], img.shape[1], img.strides[0], float(c), float(r),
                                  pr, int(s), out.ctypes.data_as(_u8p))
    return out


def match_template(image, templ, mtype=None):
    image, pi = _u8(image)
    templ, pt = _u8(templ)
    s = templ.shape[0]
    out = np.empty((image.shape[0] - s + 1, image.shape[1] - s + 1), dtype=np.float32)
    rc = lib().sid_oracle_match_template(pi, image.shape[0], image.shape[1], image.strides[0], pt, s,
                                         out.ctypes.data_as(_f32p))
    if rc:
        raise ValueError('window smaller than template')
    return out


def hessian(ccm, flags=1):
    ccm = np.ascontiguousarray(ccm, dtype=np.float32)
    out = np.empty_like(ccm)
    lib().sid_oracle_hessian(ccm.ctypes.data_as(_f32p), ccm.shape[0], ccm.shape[1], int(flags),
                             out.ctypes.data_as(_f32p))
    return out


def pm_batch(img1, img2, c1, r1, c2fg, r2fg, border, img_size, alpha0, angles, rot=None, flags=1,
             nthreads=1, want_gap=False):
    """(N,5) float64 + (N,3) int32, same contract as oracle.pm_oracle.pm_batch.  want_gap: also the float32 [N]
    distance between the peak and the second-largest NCC value of every point (cv2-flip exposure)."""
    img1, p1 = _u8(img1)
    img2, p2 = _u8(img2)
    vecs = [_f64(v) for v in (c1, r1, c2fg, r2fg, border)]
    n = len(vecs[0][0])
    angles, pa = _f64(angles)
    if rot is not None:
        rot, prot = _f64(rot)
    else:
        prot = None
    out = np.empty((n, 5), dtype=np.float64)
    ij = np.empty((n, 3), dtype=np.int32)
    gap = np.empty(n, dtype=np.float32) if want_gap else None
    rc = lib().sid_oracle_pm_batch_gap(p1, img1.shape[0], img1.shape[1], img1.strides[0],
                                       p2, img2.shape[0], img2.shape[1], img2.strides[0],
                                       *[v[1] for v in vecs], n, int(img_size), float(alpha0), pa, prot,
                                       len(angles), int(flags), int(nthreads),
                                       out.ctypes.data_as(_f64p), ij.ctypes.data_as(_i32p),
                                       gap.ctypes.data_as(_f32p) if want_gap else None)
    if rc:
        raise ValueError('sid_oracle_pm_batch failed: %d' % rc)
    return (out, ij, gap) if want_gap else (out, ij)


def rotate_and_match(img1, c1, r1, img_size, image2, alpha0, angles, rot, flags=1):
    """pmlib.py:117-174 with the whole of ``image2`` (any rectangular shape) as the search window.  ``rot`` = [K,4] rotation
    terms of angle - alpha0 (NumPy's, as the reference computes them).  -> dict(out = dc, dr, a, r, h; ij = row, col, angle
    index; ccm; template), NaN / -1 / None for a point the reference answers with NaN x 7."""
    img1, p1 = _u8(img1)
    image2, p2 = _u8(image2)
    angles, pa = _f64(angles)
    rot, prot = _f64(rot)
    s = int(img_size)
    rh, rw = image2.shape[0] - s + 1, image2.shape[1] - s + 1
    out5 = np.empty(5, dtype=np.float64)
    ij3 = np.empty(3, dtype=np.int32)
    ccm = np.empty((max(rh, 0), max(rw, 0)), dtype=np.float32)
    tmpl = np.empty((s, s), dtype=np.uint8)
    rc = lib().sid_oracle_rotate_and_match(p1, img1.shape[0], img1.shape[1], img1.strides[0], p2, image2.shape[0], image2.shape[1],
                                           image2.strides[0], float(c1), float(r1), s, pa, prot, len(angles), int(flags),
                                           out5.ctypes.data_as(_f64p), ij3.ctypes.data_as(_i32p), ccm.ctypes.data_as(_f32p),
                                           tmpl.ctypes.data_as(_u8p))
    if rc < 0:
        raise ValueError('sid_oracle_rotate_and_match: bad shape')
    return dict(out=out5, ij=ij3, ccm=ccm if rc == 0 else None, template=tmpl if rc == 0 else None)


def max_threads():
    return lib().sid_oracle_max_threads()
