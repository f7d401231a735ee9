"""Host-side helpers the PM path needs from the reference's lib.py.

What ``pattern_matching`` touches: the two first-guess interpolators (reference lib.py:139-177 and
:179-201), the grid scatter (lib.py:408-412) and a stand-in for ``nansat.NSR``; plus the uint8 staging step
``get_uint8_image`` (lib.py:27-59), whose two full-image passes run on the GPU (include/sid_stage.h).
Reading files (lib.py:256-340) and the geo/Haversine helpers are outside the hot path.
"""
import numpy as np
from scipy.interpolate import griddata

try:                                      # a real nansat passes through untouched
    from nansat import NSR                # pragma: no cover - not installed in this image
except Exception:                         # noqa: BLE001
    class NSR(object):
        """Placeholder for nansat.NSR: carries the srs string to duck-typed domains."""
        def __init__(self, srs=None):
            self.srs = srs
            self.wkt = srs

        def __repr__(self):
            return 'NSR(%r)' % (self.srs,)


EARTH_RADIUS_KM = 6371  # mean radius the reference's Haversine uses (lib.py:25)
AVG_EARTH_RADIUS = EARTH_RADIUS_KM  # the reference's name for it


def _haversine_km(lon_a, lat_a, lon_b, lat_b):
    """Great-circle distance in km between (lon, lat) pairs given in degrees."""
    phi_a, phi_b = np.radians(lat_a), np.radians(lat_b)
    half_dphi = (phi_b - phi_a) * 0.5
    half_dlam = (np.radians(lon_b) - np.radians(lon_a)) * 0.5
    hav = np.sin(half_dphi) ** 2 + np.cos(phi_a) * np.cos(phi_b) * np.sin(half_dlam) ** 2
    return 2 * EARTH_RADIUS_KM * np.arcsin(np.sqrt(hav))


def great_circle_km(n1, x1, y1, n2, x2, y2):
    """Distance in km between pixel (x1, y1) of image n1 and pixel (x2, y2) of image n2 (what the
    reference's get_displacement_km returns, lib.py:61-85)."""
    return _haversine_km(*(n1.transform_points(x1, y1) + n2.transform_points(x2, y2)))


def get_displacement_km(n1, x1, y1, n2, x2, y2):
    """Reference name (lib.py:61) of ``great_circle_km``."""
    return great_circle_km(n1, x1, y1, n2, x2, y2)


def get_displacement_pix(n1, x1, y1, n2, x2, y2):
    """Displacement in pixels of the first image (reference lib.py:103-121): the key points of image 2 are carried
    through lon/lat into the pixel space of image 1; returns (dx, dy) = their offsets from (x1, y1)."""
    on_n1 = n1.transform_points(*n2.transform_points(x2, y2), 1)
    return on_n1[0] - x1, on_n1[1] - y1


def get_speed_ms(n1, x1, y1, n2, x2, y2):
    """Drift speed in m/s between two time-stamped images (reference lib.py:87-102)."""
    elapsed = abs((n2.time_coverage_start - n1.time_coverage_start).total_seconds())
    return 1000. * great_circle_km(n1, x1, y1, n2, x2, y2) / elapsed


def _is_projected(nsr):
    """True when ``nsr`` names a projected spatial reference.  Understood: an ``osr.SpatialReference`` (what nansat's NSR
    is: ``IsGeographic()``), the local placeholder (``.srs``), anything else that carries its definition as ``.wkt`` -
    a definition that reads as geographic ('+proj=longlat', 'GEOGCS[', 'GEOGCRS[') is not projected."""
    if nsr is None:
        return False
    is_geographic = getattr(nsr, 'IsGeographic', None)
    if callable(is_geographic):
        try:
            return not bool(is_geographic())
        except Exception:                        # noqa: BLE001
            pass
    text = getattr(nsr, 'srs', None)
    if text in (None, ''):
        text = getattr(nsr, 'wkt', None)
    if text in (None, ''):
        return False
    text = str(text).lstrip()
    return not ('+proj=longlat' in text or '+proj=latlong' in text or text.upper().startswith(('GEOGCS', 'GEOGCRS')))


def _nansat_domain():
    """nansat's ``Domain`` class when nansat is installed, else None."""
    try:
        from nansat import Domain                # pragma: no cover - not installed in this image
    except Exception:                            # noqa: BLE001
        return None
    return Domain                                # pragma: no cover


def get_drift_vectors(n1, x1, y1, n2, x2, y2, nsr=None, **kwargs):
    """u, v, lon1, lat1, lon2, lat2 of matched points (reference lib.py:375-406).

    The reference projects lon/lat through ``nansat.Domain(nsr, '-te -10 -10 10 10 -tr 1 1')`` - a grid with origin
    (-10, 10) and unit pixels in the units of ``nsr`` - and returns pixel differences: with (X, Y) the coordinates of
    a point in ``nsr``, its pixel is (X + 10, 10 - Y), so u = (X2 + 10) - (X1 + 10) and v = (10 - Y1) - (10 - Y2).
    That arithmetic is restated here.  Where (X, Y) come from:

    * nansat installed: the reference's own Domain call, for every ``nsr``;
    * no nansat, default or geographic ``nsr`` (lon/lat WGS84): X = lon, Y = lat;
    * no nansat, projected ``nsr`` (``_is_projected``: an ``osr.SpatialReference`` that is not geographic, the local
      placeholder's ``.srs``, or a ``.wkt`` that does not read as geographic): the images' own ``transform_points(x, y, 0, nsr)`` (pixel -> coordinates
      in the destination SRS, the call pm_postlude makes for ``srs=``, pmlib.py:473-478; Appendix A of SURVEY.md) -
      any Nansat-like object that implements it serves; one that does not raises NotImplementedError."""
    lon1, lat1 = n1.transform_points(x1, y1)
    lon2, lat2 = n2.transform_points(x2, y2)
    Domain = _nansat_domain()
    if nsr is not None and Domain is not None:
        # nansat is installed: the reference's own call for every nsr, geographic or projected (lib.py:394-399)
        d = Domain(nsr, '-te -10 -10 10 10 -tr 1 1')
        px1, py1 = d.transform_points(lon1, lat1, 1)
        px2, py2 = d.transform_points(lon2, lat2, 1)
        return px2 - px1, py1 - py2, lon1, lat1, lon2, lat2
    if _is_projected(nsr):
        try:
            X1, Y1 = n1.transform_points(x1, y1, 0, nsr)
            X2, Y2 = n2.transform_points(x2, y2, 0, nsr)
        except TypeError:
            raise NotImplementedError('a projected nsr needs nansat, or image objects whose transform_points accepts a '
                                      'destination SRS (transform_points(x, y, 0, nsr))')
    else:
        X1, Y1, X2, Y2 = lon1, lat1, lon2, lat2
    px1, py1 = X1 - (-10.0), 10.0 - Y1                                        # Domain('-te -10 -10 10 10 -tr 1 1')
    px2, py2 = X2 - (-10.0), 10.0 - Y2
    return px2 - px1, py1 - py2, lon1, lat1, lon2, lat2


def _poly_terms(x, y, order):
    """Columns of the design matrix in the reference's term order (lib.py:160-168):
    1, x, y | x^2, y^2, xy | x^3, y^3, x^2 y, y^2 x."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    terms = [np.ones(len(x)), x, y]
    if order > 1:
        terms.extend((x ** 2, y ** 2, x * y))
    if order > 2:
        terms.extend((x ** 3, y ** 3, x ** 2 * y, y ** 2 * x))
    return np.vstack(terms).T


def fit_polynomial_map(x1, y1, x2, y2, order=1):
    """Least-squares polynomial map (x1, y1) -> (x2, y2); returns ``f(x, y) -> (x', y')`` for flat arrays.
    The solve is NumPy's ``lstsq`` with ``rcond=-1`` on the design matrix above, one right-hand side at a
    time, and the evaluation is a matrix-vector product - the arithmetic of lib.py:139-177."""
    design = _poly_terms(x1, y1, order)
    coef_x = np.linalg.lstsq(design, x2, rcond=-1)[0]
    coef_y = np.linalg.lstsq(design, y2, rcond=-1)[0]

    def evaluate(x, y):
        g = _poly_terms(x, y, order)
        return np.dot(g, coef_x), np.dot(g, coef_y)
    evaluate.coefficients = (coef_x, coef_y)
    return evaluate


def interpolation_poly(x1, y1, x2, y2, x1grd, y1grd, order=1, **kwargs):
    """Reference name and signature (lib.py:139-177): the polynomial map evaluated on the grid points,
    shaped like ``x1grd``."""
    fx, fy = fit_polynomial_map(x1, y1, x2, y2, order)(np.ravel(x1grd), np.ravel(y1grd))
    return fx.reshape(np.shape(x1grd)), fy.reshape(np.shape(x1grd))


# Triangulations started ahead of their use (``prefetch_triangulation``): key = digest of the point array, value = a
# one-element future.  SeaIceDrift.get_drift_FT starts the triangulation of its matched key points on a worker thread the
# moment the filters return - the Delaunay triangulation (Qhull, 30-85 ms for 2-3 x 10^4 points) is the longest step of the
# pattern-matching prelude, and whatever the caller does between the two calls now runs beside it.
_TRI_CACHE = {}
_TRI_LOCK = None


def _tri_key(src):
    import hashlib
    a = np.ascontiguousarray(src, dtype=np.float64)
    return hashlib.blake2b(a.tobytes(), digest_size=16).digest() + str(a.shape).encode()


def prefetch_triangulation(src):
    """Start ``scipy.spatial.Delaunay(src)`` on a worker thread; ``_triangulation(src)`` with the SAME points (bit for bit)
    picks the result up.  At most two are kept."""
    import threading
    global _TRI_LOCK
    if _TRI_LOCK is None:
        _TRI_LOCK = threading.Lock()
    src = np.array(src, dtype=np.float64)
    if src.ndim != 2 or src.shape[0] < 4 or src.shape[1] != 2:
        return None
    key = _tri_key(src)
    box = {}

    def work():
        from scipy.spatial import Delaunay
        try:
            box['tri'] = Delaunay(src)
        except Exception as e:                   # noqa: BLE001 - the consumer triangulates itself and raises what it raises
            box['err'] = e
    t = threading.Thread(target=work, name='sid-delaunay', daemon=True)
    with _TRI_LOCK:
        while len(_TRI_CACHE) >= 2:
            _TRI_CACHE.pop(next(iter(_TRI_CACHE)))
        _TRI_CACHE[key] = (t, box)
    t.start()
    return key


def _triangulation(src):
    """``scipy.spatial.Delaunay(src)`` - the one started by ``prefetch_triangulation`` for these very points, if any."""
    from scipy.spatial import Delaunay
    if _TRI_CACHE:
        key = _tri_key(src)
        with _TRI_LOCK:
            hit = _TRI_CACHE.pop(key, None)
        if hit is not None:
            t, box = hit
            t.join()
            tri = box.get('tri')
            if tri is not None and tri.points.shape == np.shape(src) and np.array_equal(tri.points, src):
                return tri
    return Delaunay(src)


def interpolation_near(x1, y1, x2, y2, x1grd, y1grd, method='linear', first_guess_device=None, **kwargs):
    """scipy griddata of x2/y2 from the keypoints onto the grid points; NaN outside the
    convex hull (reference lib.py:179-201; note the (row, col) point order).

    ``first_guess_device`` (not a reference argument): GPU index on which the grid points are located in the
    triangulation and interpolated (include/sid_fg.h) - the triangulation is SciPy's Delaunay either way.  ``None``
    evaluates with SciPy on the host."""
    src = np.array([y1, x1]).T
    dst = np.array([y1grd, x1grd]).T
    if method == 'linear' and src.shape[1] == 2:
        # The reference calls griddata twice, i.e. triangulates the same keypoints twice - the slowest
        # step of the whole prelude.  One Delaunay triangulation serves both components; every component
        # is the same barycentric sum, so the values are bit-identical to the two separate calls.
        tri = _triangulation(src)
        vals = np.array([x2, y2], dtype=np.float64).T
        if first_guess_device is not None and _has_degenerate_simplex(tri):
            # SciPy gives a (nearly) flat simplex a NaN barycentric transform and then accepts queries in its neighbours
            # with a much wider tolerance (sqrt(eps)) towards it; the device's flags do not model that,
            # so such a triangulation - collinear or duplicated key points - is evaluated by SciPy alone
            first_guess_device = None
        if first_guess_device is not None:
            # Point location and barycentric evaluation on the GPU.  Queries whose result could depend on SciPy's own choice
            # of simplex (on an edge, a vertex or the hull) or on the last bit of its arithmetic (value next to a
            # half-integer, which the caller rounds) come back flagged and are evaluated by SciPy itself, so that the
            # ROUNDED first guess is the reference's in every case (include/sid_fg.h).
            from . import _capi
            flat = dst.reshape(-1, 2)
            both, _, doubt = _capi.fg_interp_linear(tri.points, tri.simplices, vals, flat, device=first_guess_device, details=True)
            # (SciPy rejects a query outside the bounding box of the points before it looks at any simplex: queries on the
            # box, where that absolute test and the barycentric one could part, go to SciPy as well)
            lo, hi = tri.points.min(axis=0), tri.points.max(axis=0)
            doubt |= (np.abs(flat - lo) < 1e-9).any(axis=1) | (np.abs(flat - hi) < 1e-9).any(axis=1)
            if doubt.any():
                from scipy.interpolate import LinearNDInterpolator
                both[doubt] = LinearNDInterpolator(tri, vals)(flat[doubt])
            both = both.reshape(dst.shape[:-1] + (2,))
        else:
            from scipy.interpolate import LinearNDInterpolator
            both = LinearNDInterpolator(tri, vals)(dst)
        return both[..., 0].T, both[..., 1].T
    return griddata(src, x2, dst, method=method).T, griddata(src, y2, dst, method=method).T


def _has_degenerate_simplex(tri, rcond_limit=1e-10):
    """True when some simplex of the 2-D triangulation is so flat that SciPy may treat it as degenerate: SciPy's test is
    LAPACK's reciprocal condition estimate of the 2 x 2 edge matrix below 1000 eps (qhull.pyx, _get_barycentric_transforms);
    here the exact 1-norm condition number with a limit three orders of magnitude more careful."""
    p = tri.points[tri.simplices]                                     # [ns, 3, 2]
    a, b = p[:, 0] - p[:, 2], p[:, 1] - p[:, 2]                       # columns of the matrix SciPy factorises
    det = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    n1 = np.maximum(np.abs(a).sum(axis=1), np.abs(b).sum(axis=1))     # ||M||_1; the adjugate has the same column sums, permuted
    nadj = np.maximum(np.abs(b[:, 1]) + np.abs(a[:, 1]), np.abs(b[:, 0]) + np.abs(a[:, 0]))
    with np.errstate(divide='ignore', invalid='ignore'):
        rcond = np.abs(det) / (n1 * nadj)
    return bool((~np.isfinite(rcond) | (rcond < rcond_limit)).any())


def _fill_gpi(shape, gpi, data):
    """Grid of ``shape`` holding ``data`` at the flat positions where ``gpi`` is set and NaN elsewhere
    (reference lib.py:408-412)."""
    grid = np.full(int(np.prod(shape)), np.nan)
    grid[np.asarray(gpi, dtype=bool).ravel()] = data
    return grid.reshape(shape)


def _percentile_ranks(n, p, ftype):
    """Neighbour ranks and weight of np.nanpercentile's 'linear' interpolation for a float32 image:
    q = p / float32(100), virtual index (n - 1) * q in float32 (numpy/lib/_function_base_impl.py)."""
    q = np.true_divide(p, ftype(100))
    vi = (n - 1) * q
    prev = np.floor(vi)
    lo = min(max(int(prev), 0), n - 1)
    return lo, min(lo + 1, n - 1), vi - prev


def _lerp(a, b, t):
    with np.errstate(all='ignore'):
        d = np.subtract(b, a)
        r = np.add(a, d * t)
        if t >= 0.5:
            r = np.subtract(b, d * (1 - t))
    return r


_STAGE_WS = {}                     # one staging workspace per device, created on first use


def get_uint8_image(image, vmin, vmax, pmin, pmax, device=0):
    """Scale a float32 image to uint8 on the GPU: signature and results of the reference's
    ``get_uint8_image`` (lib.py:27-59): ``1 + 254 * (image - vmin) / (vmax - vmin)`` clipped to [1, 255],
    0 for pixels that are not finite; ``vmin`` / ``vmax`` default to ``np.nanpercentile(image, pmin / pmax)``.

    ``image``: 2-D float32 NumPy array (uploaded) or a CUDA/HIP torch tensor (used in place); the result is of
    the same kind.  The device selects the order statistics and maps the pixels (include/sid_stage.h); the
    three-operation float32 interpolation between the two neighbouring order statistics is NumPy's own
    arithmetic, so the percentiles - and the image - equal the reference's bit for bit.  No CPU fallback."""
    import torch
    from . import _capi
    is_tensor = isinstance(image, torch.Tensor)
    if is_tensor:
        t = image
        if not t.is_cuda:
            raise ValueError('a torch input must live on the GPU (pass NumPy arrays for host data)')
    else:
        a = np.asarray(image)
        if a.dtype != np.float32:
            raise NotImplementedError('get_uint8_image on the device takes float32 images (got %s)' % a.dtype)
        t = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device('cuda', device))
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise NotImplementedError('2-D float32 image with unit inner stride expected')
    rows, cols, stride = int(t.shape[0]), int(t.shape[1]), int(t.stride(0))
    stream = torch.cuda.current_stream(t.device).cuda_stream
    ftype = np.float32
    if vmin is None or vmax is None:
        dev_index = t.device.index or 0
        ws = _STAGE_WS.get(dev_index)
        if ws is None:
            ws = _STAGE_WS[dev_index] = _capi.StageWorkspace(dev_index)
        want = [p for p, v in ((pmin, vmin), (pmax, vmax)) if v is None]
        # (the percentiles as fractions: the first pass then already counts inside sampled key ranges around them)
        n = ws.begin(t.data_ptr(), rows, cols, stride, stream, fractions=[float(p) / 100.0 for p in want])
        if n == 0:
            vals = {p: ftype(np.nan) for p in want}
        else:
            rk = {p: _percentile_ranks(n, p, ftype) for p in want}
            ranks = sorted({r for p in want for r in rk[p][:2]})
            stat = dict(zip(ranks, ws.order_stats(ranks)))
            vals = {p: _lerp(stat[rk[p][0]], stat[rk[p][1]], rk[p][2]) for p in want}
        if vmin is None:
            vmin = vals[pmin]
            print('VMIN: ', vmin)
        if vmax is None:
            vmax = vals[pmax]
            print('VMAX: ', vmax)
    with np.errstate(all='ignore'):
        denom = ftype(vmax - vmin)                    # NumPy scalar rules: float32 when either is a percentile
        vmin32 = ftype(vmin)
    out = torch.empty((rows, cols), dtype=torch.uint8, device=t.device)
    _capi.stage_scale_u8(t.data_ptr(), rows, cols, stride, vmin32, denom, out.data_ptr(), int(out.stride(0)), stream)
    return out if is_tensor else out.cpu().numpy()
