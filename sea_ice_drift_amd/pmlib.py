"""Pattern matching on MI355X with the reference's call shape.

Drop-in for ``sea_ice_drift.pmlib.pattern_matching`` (reference pmlib.py:326-497):

    u, v, a, r, h, lon2, lat2 = pattern_matching(lon_pm1, lat_pm1, n1, c1, r1, n2, c2, r2,
                                                 margin=0, img_size=35, threads=5, srs=..., **kwargs)

The structure is  prelude (host, NumPy)  ->  dispatch (GPU)  ->  postlude (host, NumPy):

* ``pm_prelude``   builds the five per-point vectors, the validity mask and the scene
                   rotation exactly as pmlib.py:393-428 / :249-324 do;
* the dispatch replaces the ``multiprocessing.Pool.map`` of pmlib.py:436-448 with the HIP
  kernel behind the C ABI (``_capi``).  ``threads`` is accepted for signature compatibility
  and ignored.  There is no CPU fallback: without the HIP library / a gfx950 device this
  raises.  Options the kernels do not implement (an ``mtype`` other
  than TM_CCOEFF_NORMED, a user ``template_matcher``, ``img_size`` outside 2..255) raise
  ``NotImplementedError``;
* ``pm_postlude``  turns the (N,5) result block into the seven output grids as
                   pmlib.py:451-497 does.

The per-point functions of the reference are here with their own signatures as well - ``rotate_and_match`` (pmlib.py:117-174,
any rectangular ``image2`` up to a whole image: the large-window pipeline of csrc/pm_large.hip), ``use_mcc`` (:176-212),
``get_template`` (:89-115), ``get_hessian`` (:36-59), ``get_distance_to_nearest_keypoint`` (:61-77) - each a call into the
HIP library.
"""
from __future__ import absolute_import, print_function

import atexit
import threading
import time

import numpy as np
from scipy.spatial import cKDTree

from sea_ice_drift_amd import _capi
from sea_ice_drift_amd.lib import NSR, _fill_gpi, interpolation_near, interpolation_poly

DEFAULT_SRS = '+proj=latlong +datum=WGS84 +ellps=WGS84 +no_defs'
TM_CCOEFF_NORMED = 5                  # cv2.TM_CCOEFF_NORMED, the reference's default mtype (pmlib.py:119)


# ------------------------------------------------------------------ small pieces
rotation_terms = _capi.rotation_terms       # (cos a, sin a, tcT0, tcT1) with NumPy, as pmlib.py:105-110 computes them
rotation_table = _capi.rotation_table       # [K,4] of rotation_terms(angle - alpha0) (pmlib.py:151)


def get_initial_rotation(n1, n2):
    """Angle of n2's left edge seen in n1 pixel space, degrees (reference pmlib.py:79-87)."""
    lons, lats = n2.get_corners()
    x0, y0 = n1.transform_points([lons[0]], [lats[0]], 1)
    x1, y1 = n1.transform_points([lons[1]], [lats[1]], 1)
    return np.degrees(np.arctan2(x1 - x0, y1 - y0)[0])


def _first_guess_device(kwargs):
    """Where the first guess is evaluated: kwarg ``first_guess_on`` = 'device' | 'host' | 'auto' (default: the GPU
    when one is visible).  Not a reference kwarg; the reference's own code path is 'host'."""
    mode = kwargs.get('first_guess_on', 'auto')
    if mode == 'host':
        return None
    if mode == 'device':
        return int(kwargs.get('device', 0))
    if mode != 'auto':
        raise ValueError("first_guess_on must be 'auto', 'device' or 'host'")
    try:
        import torch
        return int(kwargs.get('device', 0)) if torch.cuda.is_available() else None
    except ImportError:
        return None


def nearest_keypoint_distance(x_kp, y_kp, rows_q, cols_q, shape=None, device=None):
    """Distance from integer pixels (rows_q, cols_q) to the nearest keypoint pixel.

    Same numbers as sampling the reference's full-image Euclidean distance transform
    (pmlib.py:61-77: seed[uint16(y), uint16(x)] = True; distance_transform_edt) at those
    pixels, but evaluated only where needed with a KD-tree: at 10000x10000 the EDT costs
    ~0.8 GB and seconds of CPU.  Both are sqrt of an exact integer
    squared distance in float64.  ``shape`` = (rows, cols) of image 2: a key point whose uint16-cast pixel lies
    outside it cannot seed the reference's EDT image - its fancy-index assignment raises IndexError, and so
    does this.
    """
    seed_r, seed_c = np.uint16(y_kp), np.uint16(x_kp)
    if shape is not None and seed_r.size and (seed_r.max() >= shape[0] or seed_c.max() >= shape[1]):
        bad = int(np.argmax((seed_r >= shape[0]) | (seed_c >= shape[1])))
        raise IndexError('key point %d at pixel (row %d, col %d) is out of bounds for image 2 of shape %s '
                         '(reference pmlib.py:73)' % (bad, int(seed_r[bad]), int(seed_c[bad]), tuple(shape[:2])))
    seeds = np.stack([seed_r.astype(np.float64), seed_c.astype(np.float64)], axis=1)
    q = np.stack([np.asarray(rows_q, dtype=np.float64), np.asarray(cols_q, dtype=np.float64)], axis=1)
    if device is not None:                                      # brute force on the GPU: the same exact distances
        return _capi.fg_nearest_dist(seeds, q, device=device)
    tree = cKDTree(seeds)
    d, _ = tree.query(q, k=1)
    return d


def prepare_first_guess(c2pm1, r2pm1, n1, c1, r1, n2, c2, r2, img_size,
                        min_fg_pts=5, min_border=20, max_border=50, old_border=True, **kwargs):
    """First-guess position and search border per grid point (reference pmlib.py:249-324)."""
    n2_shape = n2.shape()
    fg_dev = _first_guess_device(kwargs)
    lon1, lat1 = n1.transform_points(c1, r1)
    c1n2, r1n2 = n2.transform_points(lon1, lat1, 1)

    c2p, r2p = np.round(interpolation_poly(c1n2, r1n2, c2, r2, c2pm1, r2pm1, **kwargs))
    c2fg, r2fg = np.round(interpolation_near(c1n2, r1n2, c2, r2, c2pm1, r2pm1, first_guess_device=fg_dev, **kwargs))

    if old_border:
        border = np.zeros(c2pm1.size) + max_border
        inside = ((c2pm1 >= 0) * (c2pm1 < n2_shape[1]) * (r2pm1 >= 0) * (r2pm1 < n2_shape[0]))
        # the reference indexes the EDT image with int16-cast rounded coordinates (:304-305)
        rq = np.round(r2pm1[inside]).astype(np.int16)
        cq = np.round(c2pm1[inside]).astype(np.int16)
        # negative int16 values index from the end, as NumPy fancy indexing would
        rq = np.where(rq < 0, rq + n2_shape[0], rq)
        cq = np.where(cq < 0, cq + n2_shape[1], cq)
        border[inside] = nearest_keypoint_distance(c2, r2, rq, cq, shape=n2_shape, device=fg_dev)
    else:
        c2t, r2t = interpolation_poly(c1n2, r1n2, c2, r2, c1n2, r1n2, **kwargs)
        # (these values are not rounded but go through hypot and floor: SciPy's own evaluation, so that no last bit differs)
        c2d, r2d = interpolation_near(c1n2, r1n2, c2 - c2t, r2 - r2t, c2pm1, r2pm1, first_guess_device=None, **kwargs)
        border = np.hypot(c2d, r2d)

    border[border < min_border] = min_border
    border[border > max_border] = max_border
    border[np.isnan(c2fg)] = max_border
    border = np.floor(border)

    nofg = np.isnan(c2fg)
    c2fg[nofg] = c2p[nofg]
    nofg = np.isnan(r2fg)
    r2fg[nofg] = r2p[nofg]
    return c2fg, r2fg, border


# ------------------------------------------------------------------ prelude / postlude
def pm_prelude(lon_pm1, lat_pm1, n1, c1, r1, n2, c2, r2, margin=0, img_size=35, **kwargs):
    """Host work before the per-point sweep (reference pmlib.py:394-428)."""
    dst_shape = lon_pm1.shape
    c2pm1, r2pm1 = n2.transform_points(lon_pm1.flatten(), lat_pm1.flatten(), 1)
    c2pm1i, r2pm1i = np.round([c2pm1, r2pm1])                       # half-to-even, as np.round
    lon1i, lat1i = n2.transform_points(c2pm1i, r2pm1i)
    c1pm1i, r1pm1i = n1.transform_points(lon1i, lat1i, 1)

    c2fg, r2fg, brd2 = prepare_first_guess(c2pm1i, r2pm1i, n1, c1, r1, n2, c2, r2, img_size, **kwargs)

    hws = round(img_size / 2) + 1                                   # Python-3 round (pmlib.py:417)
    hyp = np.hypot(hws, hws)
    rows2, cols2 = n2.shape()[0], n2.shape()[1]
    rows1, cols1 = n1.shape()[0], n1.shape()[1]
    gpi = ((c2fg - brd2 - hws - margin > 0) * (r2fg - brd2 - hws - margin > 0) *
           (c2fg + brd2 + hws + margin < cols2) * (r2fg + brd2 + hws + margin < rows2) *
           (c1pm1i - hyp - margin > 0) * (r1pm1i - hyp - margin > 0) *
           (c1pm1i + hyp + margin < cols1) * (r1pm1i + hyp + margin < rows1))
    alpha0 = get_initial_rotation(n1, n2)
    return dict(dst_shape=dst_shape, c2pm1=c2pm1, r2pm1=r2pm1, c2pm1i=c2pm1i, r2pm1i=r2pm1i,
                c1pm1i=c1pm1i, r1pm1i=r1pm1i, c2fg=c2fg, r2fg=r2fg, brd2=brd2, gpi=gpi, alpha0=alpha0)


def pm_postlude(pre, results, n2, srs=DEFAULT_SRS):
    """(N,5) results -> u, v, a, r, h, lon2, lat2 grids (reference pmlib.py:451-497)."""
    dst_shape, gpi = pre['dst_shape'], pre['gpi']
    if len(results) == 0:
        return tuple(np.zeros(dst_shape) + np.nan for _ in range(7))
    results = np.asarray(results, dtype=np.float64)
    # integer start coordinates were matched; add the sub-pixel part back (:469-470)
    dci, dri = pre['c2pm1'] - pre['c2pm1i'], pre['r2pm1'] - pre['r2pm1i']
    c2pm2, r2pm2 = results[:, 0] + dci[gpi], results[:, 1] + dri[gpi]

    xpm1, ypm1 = n2.transform_points(pre['c2pm1'], pre['r2pm1'], 0, NSR(srs))
    xpm2, ypm2 = n2.transform_points(c2pm2, r2pm2, 0, NSR(srs))
    lon_pm2, lat_pm2 = n2.transform_points(c2pm2, r2pm2, 0)

    u = _fill_gpi(dst_shape, gpi, xpm2) - xpm1.reshape(dst_shape)
    v = _fill_gpi(dst_shape, gpi, ypm2) - ypm1.reshape(dst_shape)
    a = _fill_gpi(dst_shape, gpi, results[:, 2])
    r = _fill_gpi(dst_shape, gpi, results[:, 3])
    h = _fill_gpi(dst_shape, gpi, results[:, 4])
    return u, v, a, r, h, _fill_gpi(dst_shape, gpi, lon_pm2), _fill_gpi(dst_shape, gpi, lat_pm2)


# ------------------------------------------------------------------ dispatch
def _sweep_options(kwargs):
    """Pick the kernel's options out of the reference's catch-all kwargs (pmlib.py:358-375)."""
    if kwargs.get('template_matcher') is not None:
        raise NotImplementedError('template_matcher= is a CPU plug point of the reference (pmlib.py:120); '
                                  'the device kernel implements TM_CCOEFF_NORMED only')
    # mtype (pmlib.py:119,156) is handed to the matcher; the device matcher IS cv2.TM_CCOEFF_NORMED (= 5 in every OpenCV
    # release: imgproc.hpp TemplateMatchModes).  Anything else would be answered with the wrong correlation: refuse it.
    mtype = kwargs.get('mtype', None)
    if mtype is not None and (isinstance(mtype, bool) or mtype != TM_CCOEFF_NORMED):
        raise NotImplementedError('mtype=%r: the device matcher implements cv2.TM_CCOEFF_NORMED (%d) only (pmlib.py:119,156)'
                                  % (mtype, TM_CCOEFF_NORMED))
    # rot_order (pmlib.py:89,112-113): scipy's spline order of the template rotation, 0..5, all on the device: 0 (nearest, the
    # default), 1 (bilinear), 2..5 (the WHOLE image 1 through scipy's recursive B-spline prefilter once per pair, then n + 1
    # weights per axis per sample) - scipy's float64 arithmetic and uint8 rounding, operation for operation.  Anything else is
    # what scipy itself refuses ('spline order not supported').
    rot_order = kwargs.get('rot_order', 0)
    if isinstance(rot_order, bool) or rot_order not in (0, 1, 2, 3, 4, 5):
        raise RuntimeError('rot_order=%r: spline order not supported (scipy.ndimage.affine_transform takes 0..5)' % (rot_order,))
    angles = list(kwargs.get('angles', [-3, 0, 3]))
    flags = _capi.flags_from_kwargs(hes_norm=kwargs.get('hes_norm', True), hes_smth=kwargs.get('hes_smth', False),
                                    mcc_norm=kwargs.get('mcc_norm', False), rot_order=int(rot_order))
    return angles, flags


# One device handle per GPU, created on first use and kept (streams, device buffers, the two pair slots).  A handle
# carries the state of ONE call at a time (current pair, resident points, results), so every use of a shared handle
# holds that device's lock from the upload to the fetch: concurrent pattern_matching / pm_dispatch calls on one GPU
# queue up instead of interleaving (the reference's module globals, pmlib.py:33-34, make it
# non-re-entrant; here the C ABI is re-entrant per handle and the Python mirror serialises per device).  A caller
# who wants two calls in flight on one GPU passes its own ``context=`` to each.
_CONTEXTS = {}
_CONTEXT_LOCKS = {}
_REGISTRY_LOCK = threading.Lock()


def _shared_context(device):
    """(handle, lock) of ``device``.  The lock object of a device is created once and never replaced."""
    with _REGISTRY_LOCK:
        lock = _CONTEXT_LOCKS.get(device)
        if lock is None:
            lock = _CONTEXT_LOCKS[device] = threading.RLock()
        ctx = _CONTEXTS.get(device)
        if ctx is None:
            ctx = _CONTEXTS[device] = _capi.PMContext(device)
        return ctx, lock


def release_contexts(timeout=5.0):
    """Destroy the per-device handles (each keeps two image-pair slots resident in HBM) and hand back the cached device
    memory of the detector, the matcher and the first-guess evaluation.  Registered with atexit; the next call creates a
    fresh handle.  Every handle is closed under its own device's lock - the one its users hold from upload to fetch - so a
    call in flight on another thread finishes first; a lock that stays held for ``timeout`` seconds (a worker stuck at
    interpreter exit) is given up on and its handle left to the process teardown."""
    with _REGISTRY_LOCK:
        items = [(device, ctx, _CONTEXT_LOCKS[device]) for device, ctx in _CONTEXTS.items()]
        _CONTEXTS.clear()                                             # (the locks stay: a device's lock is never replaced)
    for device, ctx, lock in items:
        if lock.acquire(timeout=timeout):
            try:
                ctx.close()
            finally:
                lock.release()
    _capi.release_workspaces(-1)                                      # (a no-op unless this process loaded the library)


atexit.register(release_contexts)


class _NoLock(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def pm_dispatch(img1, img2, c1, r1, c2fg, r2fg, border, img_size, alpha0, device=0, context=None, **kwargs):
    """The batch seam (reference pmlib.py:436-448): N points -> (N,5) float64 on the GPU.
    ``context``: a PMContext to run on; by default one handle per device is created on first use and reused, so
    that repeated calls pay neither for streams and events nor for device buffers again."""
    angles, flags = _sweep_options(kwargs)
    rot = rotation_table(angles, alpha0, img_size)
    ctx, lock = _shared_context(device) if context is None else (context, _NoLock())
    with lock:
        if img1 is not None:                # (None: the caller uploaded the pair to ``context`` already)
            ctx.upload_pair(img1, img2)     # slot 0, and selected: the handle may have had another pair current
        try:
            ctx.set_points(c1, r1, c2fg, r2fg, border, img_size, alpha0, angles, rot=rot, flags=flags)
        except _capi.SidPmError as e:
            if e.code == -4:
                raise NotImplementedError(str(e))
            raise
        ctx.run()
        return ctx.fetch(want_ij=False)


def get_template(img, c, r, a, s, rot_order=0, **kwargs):
    """Rotated and shifted square template: same signature and return as the reference's get_template (pmlib.py:89-115) -
    the (s, s) uint8 array scipy's affine_transform would give for ``rot_order`` 0 or 1, sampled on the GPU."""
    if isinstance(rot_order, bool) or rot_order not in (0, 1, 2, 3, 4, 5):
        raise RuntimeError('rot_order=%r: spline order not supported (scipy.ndimage.affine_transform takes 0..5)' % (rot_order,))
    return _capi.get_template(img, c, r, rotation_terms(a, s), s, rot_order=int(rot_order), device=int(kwargs.get('device', 0)))


def get_hessian(ccm, hes_norm=True, hes_smth=False, **kwargs):
    """Hessian of a cross-correlation matrix: same signature and return as the reference's get_hessian (pmlib.py:36-59) for
    the float32 matrices cv2.matchTemplate produces.  The raw magnitudes and the median are exact; np.std comes from float64
    sums, so a normalised value agrees with NumPy's to ~1e-7 relative (north_star's bar on h is 1e-5)."""
    ccm = np.asarray(ccm)
    if ccm.dtype != np.float32:
        raise NotImplementedError('get_hessian on the device takes float32 matrices (what cv2.matchTemplate returns, pmlib.py:156); '
                                  'got %s' % ccm.dtype)
    flags = (_capi.HES_NORM if hes_norm else 0) | (_capi.HES_SMTH if hes_smth else 0)
    return _capi.get_hessian(ccm, flags=flags, device=int(kwargs.get('device', 0)))


def get_distance_to_nearest_keypoint(x1, y1, shape, device=0):
    """Full-resolution matrix of the distance to the nearest key point in pixels: same signature and return as the reference's
    get_distance_to_nearest_keypoint (pmlib.py:61-77: seed[uint16(y1), uint16(x1)] = True, then scipy's exact Euclidean distance
    transform) - evaluated per pixel on the GPU through buckets of seeds; both are the square root, in float64, of an exact
    integer squared distance."""
    seed_r, seed_c = np.atleast_1d(np.uint16(y1)), np.atleast_1d(np.uint16(x1))
    if seed_r.size == 0:
        raise ValueError('get_distance_to_nearest_keypoint needs at least one key point')
    if seed_r.max() >= shape[0] or seed_c.max() >= shape[1]:
        bad = int(np.argmax((seed_r >= shape[0]) | (seed_c >= shape[1])))
        raise IndexError('key point %d at pixel (row %d, col %d) is out of bounds for an image of shape %s (reference pmlib.py:73)'
                         % (bad, int(seed_r[bad]), int(seed_c[bad]), tuple(shape[:2])))
    seeds = np.unique(np.stack([seed_r.astype(np.float64), seed_c.astype(np.float64)], axis=1), axis=0)
    return _capi.fg_distance_image(seeds, shape[0], shape[1], device=device)


def rotate_and_match(img1, c1, r1, img_size, image2, alpha0, angles=[-3, 0, 3], mtype=TM_CCOEFF_NORMED, template_matcher=None,
                     mcc_norm=False, **kwargs):
    """Rotate the template in a range of angles and run MCC for each: same signature and return as the reference's
    rotate_and_match (pmlib.py:117-174) -

        dc, dr, best_a, best_r, best_h, best_result, best_template

    with ``best_result`` the float32 cross-correlation matrix and ``best_template`` the uint8 template of the winning angle, or
    seven NaNs when a rotated template touches a 0 pixel (pmlib.py:152-154).  ``image2`` is the search window - any rectangular
    uint8 array, up to a whole image (tests.py:336-337); its placements are tiled over the whole GPU (csrc/pm_large.hip).
    ``kwargs``: ``rot_order`` (0 / 1), ``hes_norm``, ``hes_smth`` as in the reference; ``device`` / ``context`` as in pm_dispatch.
    A window with fewer than two placements along an axis raises ValueError (np.gradient does, in the reference)."""
    kw = dict(kwargs, angles=angles, mtype=mtype, template_matcher=template_matcher, mcc_norm=mcc_norm)
    context = kw.pop('context', None)
    device = kw.pop('device', 0)
    ang, flags = _sweep_options(kw)
    if len(ang) == 0:
        raise UnboundLocalError("rotate_and_match with an empty list of angles (the reference's loop, pmlib.py:150, leaves best_result undefined)")
    image2 = np.asarray(image2)
    if image2.ndim != 2 or image2.shape[0] - img_size + 1 < 2 or image2.shape[1] - img_size + 1 < 2:
        raise ValueError('image2 of shape %s leaves fewer than two placements of a %d px template along an axis '
                         '(np.gradient needs two: pmlib.py:51)' % (image2.shape, img_size))
    ctx, lock = _shared_context(device) if context is None else (context, _NoLock())
    with lock:
        ctx.upload_pair(img1, image2)
        try:
            d = ctx.rotate_and_match(c1, r1, img_size, alpha0, ang, flags=flags, window=(0, 0, image2.shape[0], image2.shape[1]))
        except _capi.SidPmError as e:
            if e.code == -4:
                raise NotImplementedError(str(e))
            raise
    if d['ij'][2] < 0:
        return np.nan, np.nan, np.nan, np.nan, np.nan, np.nan, np.nan
    dc, dr, _, r, h = d['out']
    return dc, dr, ang[int(d['ij'][2])], np.float32(r), np.float32(h), d['ccm'], d['template']


def use_mcc(c1, r1, c2fg, r2fg, border, img1, img2, img_size, alpha0, **kwargs):
    """One point, same signature and return as the reference's use_mcc (pmlib.py:176-212)."""
    out = pm_dispatch(img1, img2, [c1], [r1], [c2fg], [r2fg], [border], img_size, alpha0, **kwargs)
    c2, r2, a, r, h = out[0]
    return c2, r2, a, np.float32(r), np.float32(h)


def pattern_matching(lon_pm1, lat_pm1, n1, c1, r1, n2, c2, r2,
                     margin=0, img_size=35, threads=5, srs=DEFAULT_SRS, **kwargs):
    """Run pattern matching on two images; same arguments and returns as the reference
    (pmlib.py:326-392): u, v, a, r, h, lon2_dst, lat2_dst, each shaped like lon_pm1.

    Keyword arguments that are NOT the reference's:

    ``first_guess_on`` = 'auto' (default) | 'device' | 'host' - where the first guess of the prelude is evaluated
        (``prepare_first_guess``, reference pmlib.py:249-324).  'host' is the reference's own code path: SciPy's
        ``griddata`` for the displacement field and a KD-tree for the distance to the nearest key point.  'device'
        evaluates both on the GPU (include/sid_fg.h): point location and barycentric interpolation in the SAME SciPy
        triangulation, exact nearest-key-point distances.  'auto' means 'device' whenever a GPU is visible, so this
        keyword changes the code path with the machine - on purpose, and without changing the result: for the default
        ``old_border=True`` the first guess that reaches the kernel is the ROUNDED interpolation (pmlib.py:285-288), and
        the device path flags every query whose rounded value could depend on SciPy's choice of simplex or on the last bit
        of its arithmetic and evaluates exactly those with SciPy itself (lib.interpolation_near; fixture G4 pins the
        prelude with both paths, tests/test_gpu_first_guess.py the flagging).  A triangulation with a degenerate simplex
        and ``old_border=False`` (unrounded values) are evaluated by SciPy alone.
    ``device`` (GPU index, default 0) and ``context`` (a ``_capi.PMContext`` of the caller's, for two calls in flight on
        one GPU)."""
    t0 = time.time()
    img1, img2 = n1[1], n2[1]
    _sweep_options(kwargs)                                            # unsupported options fail before any work
    # The image pair goes to the device while the host works on the first guess (its Delaunay triangulation is the
    # longest step of the prelude): the upload is a C call that does not hold the interpreter lock.
    ctx = kwargs.pop('context', None)
    ctx, lock = _shared_context(kwargs.get('device', 0)) if ctx is None else (ctx, _NoLock())
    with lock:                                                        # the handle is this call's from upload to fetch
        upload = ctx.upload_pair_background(img1, img2)
        try:
            pre = pm_prelude(lon_pm1, lat_pm1, n1, c1, r1, n2, c2, r2, margin=margin, img_size=img_size, **kwargs)
        finally:
            upload.wait()                                             # (re-raises what the upload raised)
        gpi = pre['gpi']
        if gpi.any():
            results = pm_dispatch(None, None, pre['c1pm1i'][gpi], pre['r1pm1i'][gpi], pre['c2fg'][gpi],
                                  pre['r2fg'][gpi], pre['brd2'][gpi], img_size, pre['alpha0'], context=ctx, **kwargs)
        else:
            results = np.zeros((0, 5))
    print('\n', 'Pattern matching - OK! (%3.0f sec)' % (time.time() - t0))
    return pm_postlude(pre, results, n2, srs=srs)
