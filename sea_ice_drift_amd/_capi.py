"""ctypes binding of libsid_pm.so (the C ABI of include/sid_pm.h).

This is the only way the Python host code reaches the GPU.  There is no CPU fallback:
if the HIP library is missing or no gfx950 device is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SID_PM_LIB') or os.path.join(_HERE, 'libsid_pm.so')   # SID_PM_LIB: A/B builds

HES_NORM = 1
HES_SMTH = 2
MCC_NORM = 4
ROT_ORDER1 = 8          # rot_order=1: bilinear template sampling (include/sid_pm.h)

ABI_VERSION = 4

# every symbol include/sid_pm.h declares
SYMBOLS = (
    'sid_pm_abi_version', 'sid_pm_strerror', 'sid_pm_last_error', 'sid_pm_device_count',
    'sid_pm_batch', 'sid_pm_create', 'sid_pm_destroy', 'sid_pm_set_stream', 'sid_pm_upload_pair',
    'sid_pm_select_pair', 'sid_pm_bind_pair', 'sid_pm_set_points', 'sid_pm_bind_results', 'sid_pm_run', 'sid_pm_sync', 'sid_pm_check', 'sid_pm_unpermute',
    'sid_pm_fetch', 'sid_pm_device_results', 'sid_pm_work_info', 'sid_pm_debug_point', 'sid_pm_debug_ncc_selftest',
    'sid_pm_debug_rsqrt', 'sid_pm_debug_hypot_selftest', 'sid_pm_estimate_cost', 'sid_pm_estimate_residency',
    'sid_pm_rotate_and_match', 'sid_pm_get_template', 'sid_pm_get_hessian', 'sid_pm_estimate_run_time',
)

# every symbol include/sid_ft.h declares (feature-tracking matcher, same library)
FT_SYMBOLS = ('sid_ft_knn2', 'sid_ft_knn2_device', 'sid_ft_workspace_bytes', 'sid_ft_last_error', 'sid_ft_release')

# every symbol include/sid_stage.h declares (uint8 staging, same library)
STAGE_SYMBOLS = ('sid_stage_create', 'sid_stage_destroy', 'sid_stage_begin', 'sid_stage_begin_hint', 'sid_stage_order_stats_ws',
                 'sid_stage_count_valid', 'sid_stage_order_stats', 'sid_stage_scale_u8', 'sid_stage_last_error')

# every symbol include/sid_orb.h declares (key-point detector, same library)
ORB_SYMBOLS = ('sid_orb_detect', 'sid_orb_last_error', 'sid_orb_release')

# every symbol include/sid_fg.h declares (first-guess evaluation, same library)
FG_SYMBOLS = ('sid_fg_interp_linear', 'sid_fg_nearest_dist', 'sid_fg_distance_image', 'sid_fg_last_error', 'sid_fg_release')

_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_lib = None


class SidPmError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'sid_pm error %d: %s' % (code, msg))
        self.code = code


def lib():
    """Load libsid_pm.so once; raise with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            '%s not found: build it with `make -C %s` (hipcc --offload-arch=gfx950) or '
            '`python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback.'
            % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    try:                                      # torch first: its wheel carries its own HIP runtime (same soname), and the
        import torch                          # process must end up with one copy - torch cannot enumerate the GPU after
        if torch.cuda.is_available():         # the system runtime has initialised it ("No HIP GPUs are available")
            torch.cuda.init()
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    img = [_u8p, C.c_int64, C.c_int64, C.c_int64]
    ptr = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    L.sid_pm_abi_version.restype = C.c_int
    L.sid_pm_strerror.restype = C.c_char_p
    L.sid_pm_strerror.argtypes = [C.c_int]
    L.sid_pm_last_error.restype = C.c_char_p
    L.sid_pm_device_count.argtypes = [C.POINTER(C.c_int)]
    L.sid_pm_batch.argtypes = img + img + [_f64p] * 5 + [C.c_int64, C.c_int, C.c_double, _f64p, _f64p,
                                                        C.c_int, C.c_uint32, _f64p, _i32p]
    L.sid_pm_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.sid_pm_destroy.argtypes = [C.c_void_p]
    L.sid_pm_destroy.restype = None
    L.sid_pm_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.sid_pm_upload_pair.argtypes = [C.c_void_p, C.c_int] + img + img
    L.sid_pm_select_pair.argtypes = [C.c_void_p, C.c_int]
    L.sid_pm_bind_pair.argtypes = [C.c_void_p] + ptr + ptr
    L.sid_pm_set_points.argtypes = [C.c_void_p] + [_f64p] * 5 + [C.c_int64, C.c_int, C.c_double, _f64p, _f64p,
                                                                C.c_int, C.c_uint32]
    L.sid_pm_bind_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.sid_pm_run.argtypes = [C.c_void_p]
    L.sid_pm_sync.argtypes = [C.c_void_p]
    if hasattr(L, 'sid_pm_check'):               # (absent from the libraries of earlier rounds that A/B runs load through SID_PM_LIB)
        L.sid_pm_check.argtypes = [C.c_void_p]
    L.sid_pm_fetch.argtypes = [C.c_void_p, _f64p, _i32p]
    L.sid_pm_device_results.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.sid_pm_work_info.argtypes = [C.c_void_p, _f64p]
    L.sid_pm_debug_point.argtypes = [C.c_void_p] + [C.c_double] * 5 + [C.c_int, C.c_double, _f64p, _f64p, C.c_int,
                                                                      C.c_uint32, _u8p, _f32p, _f32p, C.c_int64,
                                                                      _i32p, _f64p, _i32p, C.POINTER(C.c_int64)]
    L.sid_pm_debug_rsqrt.argtypes = [C.c_void_p, _f64p, _f64p, C.c_int64]
    L.sid_pm_debug_ncc_selftest.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int, C.POINTER(C.c_uint64)]
    if hasattr(L, 'sid_pm_unpermute'):
        L.sid_pm_unpermute.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sid_pm_estimate_cost.argtypes = [_f64p, C.c_int64, C.c_int, C.c_int, C.c_uint32, _f64p]
    L.sid_pm_estimate_residency.argtypes = [_f64p, C.c_int64, C.c_int, C.c_int, C.c_uint32, _i32p]
    L.sid_pm_debug_hypot_selftest.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_uint64)]
    if hasattr(L, 'sid_pm_estimate_run_time'):
        L.sid_pm_estimate_run_time.argtypes = [_f64p, C.c_int64, C.c_int, C.c_int, C.c_uint32, _f64p]
    if hasattr(L, 'sid_pm_rotate_and_match'):
        L.sid_pm_rotate_and_match.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                              C.c_double, _f64p, _f64p, C.c_int, C.c_uint32, _f64p, _i32p, _f32p, C.c_int64, _u8p]
        L.sid_pm_get_template.argtypes = [C.c_int, _u8p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, _f64p, C.c_int, C.c_int, _u8p]
        L.sid_pm_get_hessian.argtypes = [C.c_int, _f32p, C.c_int64, C.c_int64, C.c_uint32, _f32p]
    L.sid_ft_knn2.argtypes = [C.c_int, _u8p, C.c_int64, _u8p, C.c_int64, _i32p, _i32p]
    L.sid_ft_knn2_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sid_ft_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
    L.sid_ft_workspace_bytes.restype = C.c_int64
    L.sid_ft_last_error.restype = C.c_char_p
    L.sid_stage_count_valid.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    L.sid_stage_order_stats.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_int,
                                        _f32p, C.c_void_p]
    L.sid_stage_scale_u8.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_void_p,
                                     C.c_int64, C.c_void_p]
    L.sid_stage_last_error.restype = C.c_char_p
    L.sid_stage_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.sid_stage_destroy.argtypes = [C.c_void_p]
    L.sid_stage_destroy.restype = None
    L.sid_stage_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    L.sid_stage_order_stats_ws.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int, _f32p]
    if hasattr(L, 'sid_stage_begin_hint'):
        L.sid_stage_begin_hint.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_int,
                                           C.POINTER(C.c_int64), C.c_void_p]
    L.sid_orb_detect.argtypes = [C.c_int, _u8p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int8), _i32p,
                                 _f32p, _i32p, C.POINTER(C.c_int64), _u8p, C.c_int64, C.POINTER(C.c_int64)]
    L.sid_orb_last_error.restype = C.c_char_p
    L.sid_fg_interp_linear.argtypes = [C.c_int, _f64p, C.c_int64, _i32p, C.c_int64, _f64p, _f64p, C.c_int64, _f64p, _i32p, _i32p]
    L.sid_fg_nearest_dist.argtypes = [C.c_int, _f64p, C.c_int64, _f64p, C.c_int64, _f64p]
    if hasattr(L, 'sid_fg_distance_image'):
        L.sid_fg_distance_image.argtypes = [C.c_int, _f64p, C.c_int64, C.c_int64, C.c_int64, _f64p]
    L.sid_fg_last_error.restype = C.c_char_p
    # SID_PM_LIB (A/B runs against the library of an earlier round) may lack the entry points added since
    optional = ('sid_pm_check', 'sid_pm_unpermute', 'sid_pm_rotate_and_match', 'sid_pm_get_template', 'sid_pm_get_hessian', 'sid_pm_estimate_run_time') if os.environ.get('SID_PM_LIB') else ()
    for name in SYMBOLS:
        if name not in optional:
            getattr(L, name)                  # AttributeError here = header/library mismatch
    if L.sid_pm_abi_version() != ABI_VERSION and not os.environ.get('SID_PM_LIB'):
        raise ImportError('libsid_pm.so ABI %d != binding ABI %d' % (L.sid_pm_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        L = lib()
        msg = L.sid_pm_last_error() or L.sid_pm_strerror(rc)
        raise SidPmError(rc, msg.decode() if isinstance(msg, bytes) else str(msg))


def device_count():
    n = C.c_int(0)
    rc = lib().sid_pm_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _u8(a):
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise TypeError('images must be 2-D uint8 arrays (reference contract: lib.py:27-59)')
    if a.strides[1] != 1:
        a = np.ascontiguousarray(a)
    return a


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(t)


def release_workspaces(device=-1):
    """Hand the cached device memory of the detector, the matcher and the first-guess evaluation back (``sid_orb_release``,
    ``sid_ft_release``, ``sid_fg_release``; -1: every device).  No call on that device may be in flight.
    A process that never loaded the library has nothing cached: this returns without loading it (it is registered with
    atexit through pmlib.release_contexts, and loading means importing torch and initialising the GPU - not at interpreter
    shutdown, and not in a process that only used the host code)."""
    if _lib is None:
        return
    L = _lib
    for name in ('sid_orb_release', 'sid_ft_release', 'sid_fg_release'):
        if hasattr(L, name):
            getattr(L, name)(int(device))


def unpermute(stack_ptr, world, m, perm_ptr, n, out_ptr, ij_ptr, stream):
    """``sid_pm_unpermute`` on raw device(-visible) pointers (torch ``data_ptr()``) and a HIP stream handle."""
    _check(lib().sid_pm_unpermute(stack_ptr, int(world), int(m), perm_ptr, int(n), out_ptr, ij_ptr, stream))


def estimate_cost(border, img_size=34, n_angles=15, flags=HES_NORM):
    """Estimated nanoseconds per grid point (include/sid_pm.h sid_pm_estimate_cost): host arithmetic of the library.
    ``flags``: the flags of the run (they decide the Hessian's LDS layout, hence the class borders)."""
    b = _f64(border).ravel()
    out = np.empty(b.size, dtype=np.float64)
    _check(lib().sid_pm_estimate_cost(_p(b, _f64p), b.size, int(img_size), int(n_angles), int(flags), _p(out, _f64p)))
    return out


CLASS_PER_CU, CLASS_GS, CLASS_BIG, CLASS_LARGE = 15, 16, 32, 128          # include/sid_pm.h SID_PM_CLASS_*


def estimate_run_time(border, img_size=34, n_angles=15, flags=HES_NORM):
    """Estimated kernel time (ns) of one run over points with these borders (include/sid_pm.h sid_pm_estimate_run_time): point
    costs, launch tails, the launcher's side-by-side rule, large-window points - host arithmetic of the library."""
    b = _f64(border).ravel()
    t = C.c_double(0.0)
    _check(lib().sid_pm_estimate_run_time(_p(b, _f64p), b.size, int(img_size), int(n_angles), int(flags), C.byref(t)))
    return t.value


def estimate_residency(border, img_size=34, n_angles=15, flags=HES_NORM):
    """Launch class of each grid point (include/sid_pm.h sid_pm_estimate_residency): workgroups per CU (1 .. 4) in the low four
    bits (``& CLASS_PER_CU``), ``CLASS_GS`` for the launches that keep their per-placement sums in global memory, ``CLASS_BIG``
    for borders beyond 68 px; points of equal value share a launch."""
    b = _f64(border).ravel()
    out = np.empty(b.size, dtype=np.int32)
    _check(lib().sid_pm_estimate_residency(_p(b, _f64p), b.size, int(img_size), int(n_angles), int(flags), _p(out, _i32p)))
    return out


def rotation_terms(angle_deg, img_size):
    """(cos a, sin a, tcT0, tcT1) for one sampling angle, computed with NumPy exactly like the reference's get_template
    (pmlib.py:105-110): tc = int(s/2.)+1, a = radians(angle), transform = [[cos,-sin],[sin,cos]], tcT = [tc,tc].dot(transform).
    The C ABI requires these numbers from its caller (include/sid_pm.h: `rot`), so that the device samples the same float64
    coordinates the reference's scipy call would."""
    tc = int(img_size / 2.) + 1
    tc = np.array([tc, tc])
    a = np.radians(angle_deg)
    transform = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    tct = tc.dot(transform)
    return float(transform[0, 0]), float(transform[1, 0]), float(tct[0]), float(tct[1])


def rotation_table(angles, alpha0, img_size):
    """[K,4] table of rotation_terms(angle - alpha0) (pmlib.py:151)."""
    return np.array([rotation_terms(a - alpha0, img_size) for a in angles], dtype=np.float64).reshape(-1, 4)


def _rot_arg(rot, angles, alpha0, img_size):
    """The `rot` argument of the C ABI: the caller's table, or NumPy's own values (never the library's: sid_pm.h)."""
    rot = rotation_table(angles, alpha0, img_size) if rot is None else _f64(rot)
    if rot.shape != (len(angles), 4):
        raise ValueError('rot must be [n_angles, 4]')
    return rot


def rot_order_flag(order):
    """rot_order 0..5 as flag bits 3..5 (include/sid_pm.h SID_PM_ROT_ORDER); order 1 = ROT_ORDER1."""
    return (int(order) & 7) << 3


def flags_from_kwargs(hes_norm=True, hes_smth=False, mcc_norm=False, rot_order=0):
    if isinstance(rot_order, bool) or rot_order not in (0, 1, 2, 3, 4, 5):
        raise ValueError('rot_order=%r: spline orders 0..5 (scipy.ndimage.affine_transform)' % (rot_order,))
    return ((HES_NORM if hes_norm else 0) | (HES_SMTH if hes_smth else 0) | (MCC_NORM if mcc_norm else 0) | rot_order_flag(rot_order))


def pm_batch(img1, img2, c1, r1, c2fg, r2fg, border, img_size, alpha0, angles, rot=None, flags=HES_NORM):
    """One-shot host call == the Pool.map seam (pmlib.py:436-448).  -> (N,5) f64, (N,3) i32."""
    img1, img2 = _u8(img1), _u8(img2)
    v = [_f64(x) for x in (c1, r1, c2fg, r2fg, border)]
    n = len(v[0])
    angles = _f64(angles)
    rot = _rot_arg(rot, angles, alpha0, img_size) if len(angles) else _f64(np.zeros((0, 4)))
    rotp = _p(rot, _f64p)
    out = np.empty((n, 5), dtype=np.float64)
    ij = np.empty((n, 3), dtype=np.int32)
    _check(lib().sid_pm_batch(_p(img1, _u8p), img1.shape[0], img1.shape[1], img1.strides[0],
                              _p(img2, _u8p), img2.shape[0], img2.shape[1], img2.strides[0],
                              *[_p(x, _f64p) for x in v], n, int(img_size), float(alpha0),
                              _p(angles, _f64p), rotp, len(angles), int(flags), _p(out, _f64p), _p(ij, _i32p)))
    return out, ij


class PMContext(object):
    """Device-resident handle: images and points stay in HBM between runs."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().sid_pm_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self.n = 0
        self._keep = []

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib().sid_pm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except TypeError:                                  # interpreter shutdown: the module globals are already gone
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_handle):
        _check(lib().sid_pm_set_stream(self._h, C.c_void_p(int(stream_handle) if stream_handle else 0)))

    def upload_pair(self, img1, img2, slot=0, select=True):
        """Copy a host pair into device slot 0/1 and (select=True) make it the pair the next run matches.
        Streaming code that prefetches pair k+1 while pair k runs passes select=False and switches with
        select_pair (the C call alone only selects a slot when no other owned slot is current)."""
        img1, img2 = _u8(img1), _u8(img2)
        self._keep = (self._keep + [img1, img2])[-4:]  # async copies: keep the host buffers of both slots alive
        _check(lib().sid_pm_upload_pair(self._h, int(slot),
                                        _p(img1, _u8p), img1.shape[0], img1.shape[1], img1.strides[0],
                                        _p(img2, _u8p), img2.shape[0], img2.shape[1], img2.strides[0]))
        if select:
            self.select_pair(slot)

    def upload_pair_background(self, img1, img2, slot=0):
        """upload_pair on a worker thread, entered before this returns: the C call runs without the interpreter lock,
        so host work of the caller (however long it keeps the lock itself) overlaps the copy.  Returns an object
        whose wait() joins the thread and re-raises what the upload raised."""
        import threading
        img1, img2 = _u8(img1), _u8(img2)
        self._keep = (self._keep + [img1, img2])[-4:]
        fn, h, s = lib().sid_pm_upload_pair, self._h, int(slot)
        args = (h, s, _p(img1, _u8p), img1.shape[0], img1.shape[1], img1.strides[0],
                _p(img2, _u8p), img2.shape[0], img2.shape[1], img2.strides[0])
        box = {'rc': None, 'exc': None, 'msg': None}
        entered = threading.Event()
        last_error = lib().sid_pm_last_error

        def run():
            try:
                entered.set()                       # the next thing this thread does is the C call (lock released there)
                box['rc'] = fn(*args)
                if box['rc'] != 0:                  # the library's error text is thread-local: read it on this thread
                    box['msg'] = last_error()
            except BaseException as e:              # noqa: handed to the waiting thread
                box['exc'] = e

        t = threading.Thread(target=run, daemon=True)
        t.start()
        entered.wait()
        ctx = self

        class _Pending:
            def wait(self_inner):
                t.join()
                if box['exc'] is not None:
                    raise box['exc']
                if box['rc'] != 0:
                    msg = box['msg'] or lib().sid_pm_strerror(box['rc'])
                    raise SidPmError(box['rc'], msg.decode() if isinstance(msg, bytes) else str(msg))
                ctx.select_pair(s)
        return _Pending()

    def select_pair(self, slot):
        _check(lib().sid_pm_select_pair(self._h, int(slot)))

    def bind_pair_ptr(self, p1, rows1, cols1, stride1, p2, rows2, cols2, stride2):
        _check(lib().sid_pm_bind_pair(self._h, C.c_void_p(int(p1)), rows1, cols1, stride1,
                                      C.c_void_p(int(p2)), rows2, cols2, stride2))

    def bind_pair_tensors(self, t1, t2):
        """Borrow two CUDA/HIP uint8 torch tensors (2-D, unit inner stride)."""
        for t in (t1, t2):
            if t.dim() != 2 or t.stride(1) != 1 or str(t.dtype) != 'torch.uint8' or not t.is_cuda:
                raise TypeError('need 2-D uint8 device tensors with unit inner stride')
        self._keep = [t1, t2]
        self.bind_pair_ptr(t1.data_ptr(), t1.shape[0], t1.shape[1], t1.stride(0),
                           t2.data_ptr(), t2.shape[0], t2.shape[1], t2.stride(0))

    def set_points(self, c1, r1, c2fg, r2fg, border, img_size, alpha0, angles, rot=None, flags=HES_NORM):
        v = [_f64(x) for x in (c1, r1, c2fg, r2fg, border)]
        n = len(v[0])
        if any(len(x) != n for x in v):
            raise ValueError('point vectors differ in length')
        angles = _f64(angles)
        rot = _rot_arg(rot, angles, alpha0, img_size) if len(angles) else _f64(np.zeros((0, 4)))
        rotp = _p(rot, _f64p)
        _check(lib().sid_pm_set_points(self._h, *[_p(x, _f64p) for x in v], n, int(img_size), float(alpha0),
                                       _p(angles, _f64p), rotp, len(angles), int(flags)))
        self.n = n

    def bind_results_tensors(self, t_out, t_ij=None):
        """Write results into caller-owned device tensors: float64 [n,5] and int32 [n,3]."""
        if tuple(t_out.shape) != (self.n, 5) or str(t_out.dtype) != 'torch.float64' or not t_out.is_contiguous():
            raise TypeError('t_out must be a contiguous float64 [n,5] device tensor')
        if t_ij is not None and (tuple(t_ij.shape) != (self.n, 3) or str(t_ij.dtype) != 'torch.int32'
                                 or not t_ij.is_contiguous()):
            raise TypeError('t_ij must be a contiguous int32 [n,3] device tensor')
        self._keep_out = (t_out, t_ij)
        _check(lib().sid_pm_bind_results(self._h, C.c_void_p(t_out.data_ptr()),
                                         C.c_void_p(t_ij.data_ptr()) if t_ij is not None else None))

    def bind_results_host(self):
        """Zero-copy results: allocate pinned host tensors float64 [n,5] / int32 [n,3] and make the kernels write their 52 B per
        point straight into them (pinned memory is device-visible on ROCm).  After ``run()`` + ``sync()`` the tensors hold
        the results - no copy after the kernels, which is worth 1 % of a 40 000-point step and more of a shorter one.
        Returns (out, ij); they stay bound until the next ``set_points`` / ``bind_results_tensors``."""
        import torch
        out = torch.empty((self.n, 5), dtype=torch.float64, pin_memory=True)
        ij = torch.empty((self.n, 3), dtype=torch.int32, pin_memory=True)
        self.bind_results_tensors(out, ij)
        return out, ij

    def run(self):
        _check(lib().sid_pm_run(self._h))

    def sync(self):
        _check(lib().sid_pm_sync(self._h))

    def check(self):
        """Raise if a launch refused a point with a valid window (``sid_pm_check``); for callers that synchronise the
        stream themselves (torch) instead of through ``sync`` / ``fetch``."""
        if hasattr(lib(), 'sid_pm_check'):
            _check(lib().sid_pm_check(self._h))

    def fetch(self, want_ij=True):
        out = np.empty((self.n, 5), dtype=np.float64)
        ij = np.empty((self.n, 3), dtype=np.int32) if want_ij else None
        _check(lib().sid_pm_fetch(self._h, _p(out, _f64p), _p(ij, _i32p) if want_ij else None))
        return (out, ij) if want_ij else out

    def device_results(self):
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib().sid_pm_device_results(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def work_info(self):
        info = np.zeros(6, dtype=np.float64)
        _check(lib().sid_pm_work_info(self._h, _p(info, _f64p)))
        return dict(launches=int(info[0]), valid_points=int(info[1]), macs=float(info[2]),
                    hbm_bytes=float(info[3]), max_lds_bytes=int(info[4]))

    def debug_point(self, c1, r1, c2fg, r2fg, border, img_size, alpha0, angles, rot=None, flags=HES_NORM,
                    cap=256 * 256):
        angles = _f64(angles)
        rot = _rot_arg(rot, angles, alpha0, img_size)
        rotp = _p(rot, _f64p)
        K, s = len(angles), int(img_size)
        tm = np.zeros((K, s, s), dtype=np.uint8)
        ccm = np.zeros(cap, dtype=np.float32)
        hes = np.zeros(cap, dtype=np.float32)
        shape = np.zeros(2, dtype=np.int32)
        out5 = np.zeros(5, dtype=np.float64)
        ij3 = np.zeros(3, dtype=np.int32)
        cyc = np.zeros(32, dtype=np.int64)
        _check(lib().sid_pm_debug_point(self._h, float(c1), float(r1), float(c2fg), float(r2fg), float(border), s,
                                        float(alpha0), _p(angles, _f64p), rotp, K, int(flags), _p(tm, _u8p),
                                        _p(ccm, _f32p), _p(hes, _f32p), cap, _p(shape, _i32p), _p(out5, _f64p),
                                        _p(ij3, _i32p), cyc.ctypes.data_as(C.POINTER(C.c_int64))))
        rh, rw = int(shape[0]), int(shape[1])
        n = rh * rw
        return dict(templates=tm, ccm=ccm[:n].reshape(rh, rw), hes=hes[:n].reshape(rh, rw), out=out5, ij=ij3,
                    cycles=cyc)

    def rotate_and_match(self, c1, r1, img_size, alpha0, angles, rot=None, flags=HES_NORM, window=None, want_ccm=True,
                         want_template=True):
        """``sid_pm_rotate_and_match`` on the handle's current pair: the templates around (c1, r1) of image 1 against ``window`` =
        (row0, col0, rows, cols) of image 2 (None: the whole image).  -> dict(out = dc, dr, a, r, h; ij = peak row, peak column,
        angle index (-1: NaN point); ccm [rh, rw] float32 and template [s, s] uint8, None for a NaN point)."""
        angles = _f64(angles)
        rot = _rot_arg(rot, angles, alpha0, img_size)
        s = int(img_size)
        if window is None:
            raise ValueError('window = (row0, col0, rows, cols) of image 2')
        r0, c0, wh, ww = [int(v) for v in window]
        rh, rw = wh - s + 1, ww - s + 1
        out5 = np.zeros(5, dtype=np.float64)
        ij3 = np.zeros(3, dtype=np.int32)
        ccm = np.zeros(max(rh, 0) * max(rw, 0), dtype=np.float32) if want_ccm else None
        tm = np.zeros((s, s), dtype=np.uint8) if want_template else None
        _check(lib().sid_pm_rotate_and_match(self._h, float(c1), float(r1), s, r0, c0, wh, ww, float(alpha0), _p(angles, _f64p),
                                             _p(rot, _f64p), len(angles), int(flags), _p(out5, _f64p), _p(ij3, _i32p),
                                             _p(ccm, _f32p) if want_ccm else None, ccm.size if want_ccm else 0,
                                             _p(tm, _u8p) if want_template else None))
        ok = ij3[2] >= 0
        return dict(out=out5, ij=ij3, ccm=ccm.reshape(rh, rw) if (want_ccm and ok) else None,
                    template=tm if (want_template and ok) else None)

    def debug_rsqrt(self, x):
        x = _f64(x)
        y = np.empty_like(x)
        _check(lib().sid_pm_debug_rsqrt(self._h, _p(x, _f64p), _p(y, _f64p), x.size))
        return y

    def debug_ncc_selftest(self, evaluations, img_size=34, seed=1):
        """(evaluations, mismatches, spec-route evaluations) of the shortened NCC normalisation against the full one."""
        counts = (C.c_uint64 * 3)()
        _check(lib().sid_pm_debug_ncc_selftest(self._h, int(seed), int(evaluations), int(img_size), counts))
        return int(counts[0]), int(counts[1]), int(counts[2])

    def debug_hypot_selftest(self, evaluations, seed=1):
        """(evaluations, mismatches) of the kernel's shortened hypotf against the IEEE route, on the device."""
        counts = (C.c_uint64 * 2)()
        _check(lib().sid_pm_debug_hypot_selftest(self._h, int(seed), int(evaluations), counts))
        return int(counts[0]), int(counts[1])


def get_template(img, c, r, rot4, img_size, rot_order=0, device=0):
    """``sid_pm_get_template``: the s x s uint8 template around (c, r) of a host uint8 image (pmlib.py:89-115)."""
    img = _u8(img)
    rot4 = _f64(rot4).reshape(4)
    s = int(img_size)
    out = np.empty((s, s), dtype=np.uint8)
    _check(lib().sid_pm_get_template(int(device), _p(img, _u8p), img.shape[0], img.shape[1], img.strides[0], float(c), float(r),
                                     _p(rot4, _f64p), s, int(rot_order), _p(out, _u8p)))
    return out


def get_hessian(ccm, flags=HES_NORM, device=0):
    """``sid_pm_get_hessian``: the Hessian magnitudes of a float32 matrix (pmlib.py:36-59)."""
    ccm = np.ascontiguousarray(ccm, dtype=np.float32)
    if ccm.ndim != 2:
        raise ValueError('get_hessian needs a 2-D matrix')
    out = np.empty_like(ccm)
    _check(lib().sid_pm_get_hessian(int(device), _p(ccm, _f32p), ccm.shape[0], ccm.shape[1], int(flags), _p(out, _f32p)))
    return out


def ft_knn2(desc1, desc2, device=0):
    """Two nearest train descriptors (Hamming) of every query descriptor: include/sid_ft.h sid_ft_knn2.

    desc1 [n1,32], desc2 [n2,32] uint8 -> (idx [n1,2] int32, dist [n1,2] int32), -1 where absent."""
    d1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
    idx = np.empty((len(d1), 2), dtype=np.int32)
    dist = np.empty((len(d1), 2), dtype=np.int32)
    L = lib()
    rc = L.sid_ft_knn2(int(device), d1.ctypes.data_as(_u8p), len(d1), d2.ctypes.data_as(_u8p), len(d2),
                       idx.ctypes.data_as(_i32p), dist.ctypes.data_as(_i32p))
    if rc != 0:
        raise SidPmError(rc, L.sid_ft_last_error().decode())
    return idx, dist


def _stage_check(rc):
    if rc != 0:
        raise SidPmError(rc, lib().sid_stage_last_error().decode())


class StageWorkspace(object):
    """Persistent device + pinned host buffers of the staging step (include/sid_stage.h sid_stage_create)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _stage_check(lib().sid_stage_create(int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib().sid_stage_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except TypeError:                                  # interpreter shutdown: the module globals are already gone
            pass

    def begin(self, ptr, rows, cols, stride, stream=0, fractions=None):
        """First pass: number of non-NaN pixels; keeps what order_stats needs - the leading-digit histogram, or (``fractions``:
        where the wanted order statistics lie, as fractions of the non-NaN pixels) the histograms inside sampled key ranges
        around them, which saves order_stats one of its two passes (include/sid_stage.h sid_stage_begin_hint)."""
        n = C.c_int64(0)
        if fractions is not None and len(fractions) > 0:
            f = np.ascontiguousarray(fractions, dtype=np.float64)
            _stage_check(lib().sid_stage_begin_hint(self._h, C.c_void_p(int(ptr)), rows, cols, stride,
                                                    f.ctypes.data_as(C.POINTER(C.c_double)), len(f), C.byref(n), C.c_void_p(int(stream))))
        else:
            _stage_check(lib().sid_stage_begin(self._h, C.c_void_p(int(ptr)), rows, cols, stride, C.byref(n), C.c_void_p(int(stream))))
        return int(n.value)

    def order_stats(self, ranks):
        r = np.ascontiguousarray(ranks, dtype=np.int64)
        out = np.empty(len(r), dtype=np.float32)
        _stage_check(lib().sid_stage_order_stats_ws(self._h, r.ctypes.data_as(C.POINTER(C.c_int64)), len(r), out.ctypes.data_as(_f32p)))
        return out


def stage_count_valid(ptr, rows, cols, stride, stream=0):
    """Non-NaN pixels of a device float32 image (include/sid_stage.h)."""
    n = C.c_int64(0)
    _stage_check(lib().sid_stage_count_valid(C.c_void_p(int(ptr)), rows, cols, stride, C.byref(n), C.c_void_p(int(stream))))
    return int(n.value)


def stage_order_stats(ptr, rows, cols, stride, ranks, stream=0):
    """Exact order statistics (0-based ranks among the non-NaN pixels) of a device float32 image."""
    r = np.ascontiguousarray(ranks, dtype=np.int64)
    out = np.empty(len(r), dtype=np.float32)
    _stage_check(lib().sid_stage_order_stats(C.c_void_p(int(ptr)), rows, cols, stride, r.ctypes.data_as(C.POINTER(C.c_int64)),
                                             len(r), out.ctypes.data_as(_f32p), C.c_void_p(int(stream))))
    return out


def stage_scale_u8(ptr, rows, cols, stride, vmin, denom, out_ptr, out_stride, stream=0):
    _stage_check(lib().sid_stage_scale_u8(C.c_void_p(int(ptr)), rows, cols, stride, float(vmin), float(denom),
                                          C.c_void_p(int(out_ptr)), out_stride, C.c_void_p(int(stream))))


def fg_interp_linear(pts, simplices, values, q, device=0, details=False):
    """Piecewise-linear interpolation of two value columns at the points q in a given triangulation
    (include/sid_fg.h sid_fg_interp_linear): pts [n,2], simplices [m,3], values [n,2], q [k,2] -> [k,2] (NaN outside).
    details=True: also the simplex index per query (-1: none) and the doubt flags (queries on an edge, a vertex, the hull,
    or with a value next to a half-integer: to be evaluated with SciPy by the caller)."""
    pts, values, q = _f64(pts), _f64(values), _f64(q)
    simp = np.ascontiguousarray(simplices, dtype=np.int32)
    out = np.empty((len(q), 2), dtype=np.float64)
    sx = np.empty(len(q), dtype=np.int32)
    dbt = np.empty(len(q), dtype=np.int32)
    L = lib()
    rc = L.sid_fg_interp_linear(int(device), _p(pts, _f64p), len(pts), simp.ctypes.data_as(_i32p), len(simp), _p(values, _f64p),
                                _p(q, _f64p), len(q), _p(out, _f64p), sx.ctypes.data_as(_i32p), dbt.ctypes.data_as(_i32p))
    if rc != 0:
        raise SidPmError(rc, L.sid_fg_last_error().decode())
    return (out, sx, dbt.astype(bool)) if details else out


def fg_nearest_dist(seeds, q, device=0):
    """Distance from every point of q [k,2] to the nearest of seeds [n,2] (include/sid_fg.h sid_fg_nearest_dist)."""
    seeds, q = _f64(seeds), _f64(q)
    out = np.empty(len(q), dtype=np.float64)
    L = lib()
    rc = L.sid_fg_nearest_dist(int(device), _p(seeds, _f64p), len(seeds), _p(q, _f64p), len(q), _p(out, _f64p))
    if rc != 0:
        raise SidPmError(rc, L.sid_fg_last_error().decode())
    return out


def fg_distance_image(seeds, rows, cols, device=0):
    """Distance of every pixel of a rows x cols image to the nearest of seeds [n,2] = (row, column)
    (include/sid_fg.h sid_fg_distance_image) -> float64 [rows, cols]."""
    seeds = _f64(seeds)
    out = np.empty((int(rows), int(cols)), dtype=np.float64)
    L = lib()
    rc = L.sid_fg_distance_image(int(device), _p(seeds, _f64p), len(seeds), int(rows), int(cols), _p(out, _f64p))
    if rc != 0:
        raise SidPmError(rc, L.sid_fg_last_error().decode())
    return out
