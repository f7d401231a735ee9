"""Determinism soak (race detector) as part of the GPU suite.

Every configuration runs the same 6 400-point workload 200 times on one handle; every repetition must be bit-identical
to the first, and the first equals the C oracle on a subsample.  Covers the row-pair sweep (15 and 7 angles, 4- and
8-row bands), the classic kernel's paired-slot and 8-row-band sweeps (SID_PM_NO_RP), the table sampler and the
on-the-fly sampler (SID_PM_NO_SAMP_TABLE; both switches are read at every set_points), mixed and large borders (all
three residency classes).  History: DESIGN.md section 6b.
"""
import os

import numpy as np
import pytest

from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn

pytestmark = pytest.mark.gpu
REPS = 200


@pytest.fixture(scope='module')
def workload():
    size = 4000
    img1, img2 = syn.make_pair(size, size, seed=777)
    return img1, img2, size


@pytest.mark.parametrize('angles,img_size,border,no_table,no_rp', [
    (7, 34, 'mixed', False, False),     # 15 angles: row-pair sweep, table sampler
    (7, 34, 'mixed', True, False),      # ... on-the-fly sampler
    (3, 34, 'mixed', False, False),     # 7 angles: row-pair sweep with paired slots (8 output rows per item)
    (3, 35, 'mixed', True, False),      # ... on-the-fly sampler, template side 35 (three strip columns)
    (1, 34, 30, False, False),          # 3 angles (the reference's default), two workgroups per CU
    (3, 34, 'mixed', False, True),      # ... the classic kernel's paired-slot sweep (SID_PM_NO_RP)
    (3, 34, 'mixed', True, True),
    (7, 35, 44, False, False),          # reference default template side, large windows (one workgroup per CU)
    (7, 34, 28, False, False),          # two workgroups per CU: 8-row band
    (7, 34, 30, False, True),           # ... of the classic kernel
])
def test_repeated_runs_are_bit_identical(workload, c_oracle, angles, img_size, border, no_table, no_rp):
    img1, img2, size = workload
    g = syn.make_grid(size, size, 80, border=border)
    ang = list(range(-angles, angles + 1))
    rot = my.rotation_table(ang, 0.0, img_size)
    old = os.environ.pop('SID_PM_NO_SAMP_TABLE', None)
    old_rp = os.environ.pop('SID_PM_NO_RP', None)
    if no_table:
        os.environ['SID_PM_NO_SAMP_TABLE'] = '1'
    if no_rp:
        os.environ['SID_PM_NO_RP'] = '1'
    try:
        with _capi.PMContext(0) as ctx:
            ctx.upload_pair(img1, img2)
            ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], img_size, 0.0, ang, rot=rot)
            ctx.run()
            ref, ref_ij = ctx.fetch()
            bad = 0
            for _ in range(REPS - 1):
                ctx.run()
                out, ij = ctx.fetch()
                same = (ij == ref_ij).all(1) & ((out == ref) | (np.isnan(out) & np.isnan(ref))).all(1)
                bad += int((~same).sum())
    finally:
        os.environ.pop('SID_PM_NO_SAMP_TABLE', None)
        os.environ.pop('SID_PM_NO_RP', None)
        if old is not None:
            os.environ['SID_PM_NO_SAMP_TABLE'] = old
        if old_rp is not None:
            os.environ['SID_PM_NO_RP'] = old_rp
    assert bad == 0, '%d point results differed between repetitions' % bad
    sel = np.arange(0, len(ref), 53)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'][sel], g['r1'][sel], g['c2fg'][sel], g['r2fg'][sel],
                                    g['border'][sel], img_size, 0.0, ang, rot=rot, nthreads=8)
    np.testing.assert_array_equal(ref_ij[sel], exp_ij)
    np.testing.assert_array_equal(ref[sel, :4], exp[:, :4])
    np.testing.assert_allclose(ref[sel, 4], exp[:, 4], rtol=1e-5, atol=1e-5)


def _evictor(stop, counter, lock):
    """Host thread that keeps provoking evictions of this process's GPU queues: pages of a hipHostRegister'ed (user-pointer)
    buffer are invalidated (madvise / mprotect), which the kernel driver answers by quiescing the queues - wavefronts in
    flight are saved by the trap handler and restored later, on OTHER compute units of their XCD (tools/ubench/slot_life.hip:
    15 000 workgroups moved in 38 provoked evictions, none to another XCD).  This is the event that broke round 4's pool of blocks
    picked by hardware slot; the free lists of round 5 (PMArgs::ring) must not care."""
    import ctypes, mmap, time
    libc = ctypes.CDLL('libc.so.6', use_errno=True)
    # the HIP runtime THIS process already runs on (torch ships its own copy: opening "libamdhip64.so" by name would map a
    # second runtime into the process)
    path = None
    with open('/proc/self/maps') as fh:
        for line in fh:
            if 'libamdhip64.so' in line:
                path = line.split()[-1]
                break
    if path is None:
        return
    hip = ctypes.CDLL(path)
    size = 8 << 20
    while not stop.is_set():
        buf = mmap.mmap(-1, size)
        view = (ctypes.c_char * size).from_buffer(buf)
        addr = ctypes.addressof(view)
        ctypes.memset(addr, 1, size)
        # (register .. unregister under the lock the main thread holds around its copies to the host: kernels overlap the
        # evictions - the point of the test -, the runtime's own pinning of a pageable copy target does not)
        with lock:
            if hip.hipHostRegister(ctypes.c_void_p(addr), ctypes.c_size_t(size), ctypes.c_uint(0)) == 0:
                libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(size), ctypes.c_int(4))        # MADV_DONTNEED
                ctypes.memset(addr, 2, size)
                libc.mprotect(ctypes.c_void_p(addr), ctypes.c_size_t(size), ctypes.c_int(1))      # PROT_READ
                libc.mprotect(ctypes.c_void_p(addr), ctypes.c_size_t(size), ctypes.c_int(3))      # PROT_READ | PROT_WRITE
                time.sleep(0.002)
                hip.hipHostUnregister(ctypes.c_void_p(addr))
                counter[0] += 1
        del view
        buf.close()
        time.sleep(0.003)


@pytest.mark.timeout(240)
@pytest.mark.parametrize('angles,img_size', [(1, 35), (3, 34)])
def test_recycled_blocks_survive_queue_evictions(workload, c_oracle, angles, img_size):
    """The launches of at most 7 angles take the blocks of global memory that hold their per-placement sums and accumulators
    from per-XCD free lists (round 5).  300 repetitions of a 6 400-point run while a host thread provokes queue evictions:
    every repetition bit-identical to the first, the first equal to the oracle; and the lists are whole afterwards (the next
    set_points would re-initialise them, so a second handle-less check is the repetitions themselves)."""
    import faulthandler, threading
    faulthandler.dump_traceback_later(200, exit=False)                # (a hang shows where every thread stands)
    img1, img2, size = workload
    g = syn.make_grid(size, size, 80, border='mixed')
    ang = list(range(-angles, angles + 1))
    rot = my.rotation_table(ang, 0.0, img_size)
    stop, counter, lock = threading.Event(), [0], threading.Lock()
    with _capi.PMContext(0) as ctx:
        ctx.upload_pair(img1, img2)
        ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], img_size, 0.0, ang, rot=rot)
        ctx.run()
        ref, ref_ij = ctx.fetch()
        th = threading.Thread(target=_evictor, args=(stop, counter, lock), daemon=True)
        th.start()
        bad = 0
        try:
            for _ in range(300):
                ctx.run()
                ctx.sync()                                          # the kernels run - and are evicted - with the lock free
                with lock:
                    out, ij = ctx.fetch()
                same = (ij == ref_ij).all(1) & ((out == ref) | (np.isnan(out) & np.isnan(ref))).all(1)
                bad += int((~same).sum())
        finally:
            stop.set()
            th.join(timeout=10)
            faulthandler.cancel_dump_traceback_later()
    assert bad == 0, '%d point results differed between repetitions (%d evictions provoked)' % (bad, counter[0])
    assert counter[0] > 0, 'no queue eviction could be provoked on this box'
    sel = np.arange(0, len(ref), 53)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'][sel], g['r1'][sel], g['c2fg'][sel], g['r2fg'][sel],
                                    g['border'][sel], img_size, 0.0, ang, rot=rot, nthreads=8)
    np.testing.assert_array_equal(ref_ij[sel], exp_ij)
    np.testing.assert_array_equal(ref[sel, :4], exp[:, :4])
    np.testing.assert_allclose(ref[sel, 4], exp[:, 4], rtol=1e-5, atol=1e-5)
