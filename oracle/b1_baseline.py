"""CPU baseline B1: the reference-SHAPED cost structure of pattern matching.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as the rest of ``oracle/``): imported by ``bench.py``'s
``cpu_baseline`` leg and by ``tests/``; nothing under ``sea_ice_drift_amd/`` may import it.

What it is: one Python task per grid point under ``multiprocessing.Pool`` - the structure of
``pmlib.py:430-448`` (fork-inherited read-only state, ``Pool.map`` over point indices) - where each task does
what ``use_mcc`` / ``rotate_and_match`` do (pmlib.py:117-212): ``scipy.ndimage.affine_transform`` (order 0)
for every rotated template, an FFT-based float32 correlation standing in for OpenCV's DFT path inside
``cv2.matchTemplate`` (cv2 is not installed; OpenCV correlates 8-bit images through a float32 DFT and
normalises in double), ``np.argmax``, and the NumPy Hessian of pmlib.py:36-59.

What it is not: the reference itself (nothing of ``/root/reference`` runs on the GPU box) and not bit-exact -
the FFT correlation carries ~1e-6 noise like OpenCV's.  It is timed next to B2 (``oracle/pm_oracle.c``, exact
direct sums in C with OpenMP), which is the stronger baseline; neither is a target, both are labelled.
"""
import multiprocessing

import numpy as np
from scipy import ndimage as nd
from scipy import signal

_state = {}


def _template(img, c, r, angle_deg, s):
    """pmlib.py:89-115 with the same scipy call (order 0, output uint8)."""
    tc = np.array([int(s / 2.) + 1] * 2)
    a = np.radians(angle_deg)
    t = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    offset = np.array([r, c]) - tc.dot(t)
    return nd.affine_transform(img, t.T, order=0, offset=offset, output_shape=(s, s), cval=0.0, output=np.uint8)


def _ccoeff_normed_fft(window, templ):
    """TM_CCOEFF_NORMED with the raw correlation through a float32 FFT (OpenCV's route for 8-bit input)."""
    s = templ.shape[0]
    n = float(s * s)
    w32 = window.astype(np.float32)
    corr = signal.fftconvolve(w32, templ[::-1, ::-1].astype(np.float32), mode='valid').astype(np.float64)
    w64 = window.astype(np.float64)

    def box(x):
        ii = np.zeros((x.shape[0] + 1, x.shape[1] + 1))
        ii[1:, 1:] = x.cumsum(0).cumsum(1)
        return ii[s:, s:] - ii[:-s, s:] - ii[s:, :-s] + ii[:-s, :-s]
    wsum, wsq = box(w64), box(w64 * w64)
    t64 = templ.astype(np.float64)
    tmean = t64.sum() / n
    tnorm2 = (t64 * t64).sum() - n * tmean * tmean
    if tnorm2 <= 0:
        return np.ones(corr.shape, dtype=np.float32)
    num = corr - wsum * tmean
    den = np.sqrt(np.maximum(wsq - wsum * wsum / n, 0.0)) * np.sqrt(tnorm2)
    with np.errstate(divide='ignore', invalid='ignore'):
        q = num / den
    q = np.where(np.abs(num) < den, q, np.where(np.abs(num) < den * 1.125, np.sign(num), 0.0))
    return q.astype(np.float32)


def _hessian(ccm):
    dy, dx = np.gradient(ccm)
    d2x = np.gradient(dx)[1]
    d2y = np.gradient(dy)[0]
    h = np.hypot(d2x, d2y)
    return (h - np.median(h)) / np.std(h)


def _one_point(i):
    st = _state
    s, img1, img2 = st['s'], st['img1'], st['img2']
    c1, r1, c2fg, r2fg, b = (st[k][i] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border'))
    hws = int(s / 2.)
    window = img2[int(r2fg - hws - b):int(r2fg + hws + b + 1), int(c2fg - hws - b):int(c2fg + hws + b + 1)]
    best = None
    for angle in st['angles']:
        templ = _template(img1, c1, r1, angle - st['alpha0'], s)
        if templ.min() == 0:
            return (np.nan,) * 5
        res = _ccoeff_normed_fft(window, templ)
        ij = np.unravel_index(np.argmax(res), res.shape)
        if best is None or res.max() > best[0]:
            best = (res.max(), angle, res, ij)
    r, a, res, ij = best
    h = _hessian(res)[ij]
    return (c2fg + ij[1] - (window.shape[1] - s) / 2., r2fg + ij[0] - (window.shape[0] - s) / 2., a, r, h)


def _init(state):
    _state.update(state)


def run(img1, img2, c1, r1, c2fg, r2fg, border, img_size, alpha0, angles, processes=1):
    """(N,5) float64, one task per point over a fork pool of ``processes`` workers (pmlib.py:436-448)."""
    state = dict(img1=img1, img2=img2, c1=np.asarray(c1), r1=np.asarray(r1), c2fg=np.asarray(c2fg),
                 r2fg=np.asarray(r2fg), border=np.asarray(border), s=int(img_size), alpha0=float(alpha0),
                 angles=list(angles))
    n = len(state['c1'])
    if processes <= 1:
        _init(state)
        return np.array([_one_point(i) for i in range(n)], dtype=np.float64).reshape(n, 5)
    ctx = multiprocessing.get_context('fork')
    with ctx.Pool(processes, initializer=_init, initargs=(state,)) as pool:
        res = pool.map(_one_point, range(n))
    return np.array(res, dtype=np.float64).reshape(n, 5)
