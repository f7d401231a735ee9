"""BASELINE.json configurations exercised on the GPU box (one MI355X):

* config 1 stand-in : a 50x50 grid on a synthetic 4000x4000 pair through ``SeaIceDrift.get_drift_PM`` (the two
  Sentinel-1 GeoTIFFs of the reference's test are not obtainable) against the C oracle fed with the prelude
  that fixture G4 validates (``tests/test_host_logic.py``);
* config 3 dry run  : ``python bench.py --gpus 2`` exactly as the N = 1 line is invoked - bench.py spawns the ranks
  itself; on a one-GPU box both ranks share the device and the collectives run over gloo;
* config 5          : ``bench.py --mode stream`` with 16 pairs of 10000x10000 px through the two upload slots,
  three pairs checked against the oracle on a subsample;
* regression cases for the handle: pairs of another shape after set_points, pm_dispatch on a reused handle.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn
from sea_ice_drift_amd.domain import ArrayNansat
from sea_ice_drift_amd.seaicedrift import SeaIceDrift

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, universal_newlines=True)
    assert p.returncode == 0, 'bench.py %s failed:\n%s\n%s' % (' '.join(argv), p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'expected ONE JSON line, got %d' % len(lines)
    return json.loads(lines[0])


def test_config1_standin_50x50_grid_on_4000px_pair(c_oracle):
    """seaicedrift.py:62-88 on a same-size synthetic pair: every grid of the public call against the oracle."""
    H = W = 4000
    img1, img2 = syn.make_pair(H, W, seed=4100)
    # two georeferences that differ by a small rotation: alpha0 != 0, fractional pixel coordinates on image 1
    n1 = ArrayNansat.rotated(img1, angle_deg=0.0, scale=1e-3, origin=(10.0, 60.0))
    n2 = ArrayNansat.rotated(img2, angle_deg=-1.5, scale=1e-3, origin=(10.02, 60.01))
    rng = np.random.default_rng(41)
    # "feature tracking" vectors: 400 start points on image 1 and their true positions on image 2 (+ noise)
    x1 = rng.uniform(300, W - 300, 400)
    y1 = rng.uniform(300, H - 300, 400)
    dc, dr = syn.true_displacement(x1, y1)
    lon1, lat1 = n1.transform_points(x1, y1)
    lon2, lat2 = n2.transform_points(x1 + dc + rng.normal(0, 1, 400), y1 + dr + rng.normal(0, 1, 400))
    cg, rg = np.meshgrid(np.linspace(250, W - 250, 50), np.linspace(250, H - 250, 50))
    lons, lats = n2.transform_points(cg.ravel(), rg.ravel())
    lons, lats = lons.reshape(50, 50), lats.reshape(50, 50)
    kw = dict(img_size=34, angles=list(range(-4, 5)), min_border=20, max_border=40)
    u, v, a, r, h, lon_d, lat_d = SeaIceDrift(n1, n2).get_drift_PM(lons, lats, lon1, lat1, lon2, lat2, **kw)
    assert u.shape == (50, 50)
    # expected: the G4-validated prelude -> C oracle -> postlude
    xk1, yk1 = n1.transform_points(lon1, lat1, 1)
    xk2, yk2 = n2.transform_points(lon2, lat2, 1)
    pre = my.pm_prelude(lons, lats, n1, xk1, yk1, n2, xk2, yk2, **kw)
    gpi = pre['gpi']
    assert gpi.sum() > 2000
    exp, _ = c_oracle.pm_batch(img1, img2, pre['c1pm1i'][gpi], pre['r1pm1i'][gpi], pre['c2fg'][gpi], pre['r2fg'][gpi],
                               pre['brd2'][gpi], 34, pre['alpha0'], kw['angles'],
                               rot=my.rotation_table(kw['angles'], pre['alpha0'], 34), nthreads=8)
    eu, ev, ea, er, eh, elon, elat = my.pm_postlude(pre, exp, n2)
    for name, got, want in (('u', u, eu), ('v', v, ev), ('a', a, ea), ('r', r, er), ('lon2', lon_d, elon), ('lat2', lat_d, elat)):
        np.testing.assert_array_equal(got, want, err_msg=name)
    np.testing.assert_allclose(h, eh, rtol=1e-5, atol=1e-5, equal_nan=True)
    # sanity: the sweep found real peaks
    ok = np.isfinite(u)
    assert ok.sum() > 2000 and np.nanmedian(r[ok]) > 0.3


def test_config3_dry_run_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and WORLD_SIZE unset: strong scaling on the same grid."""
    one = run_bench('--gpus', '1', '--steps', '3', '--warmup', '1', '--size', '3000', '--grid', '60', '--no-cpu-baseline')
    two = run_bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--size', '3000', '--grid', '60', '--no-cpu-baseline')
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2
    assert two['scaling'] == 'strong' and two['config']['points_total'] == one['config']['points_total'] == 3600
    assert sum(two['config']['points_per_gpu_all']) == 3600 and len(two['config']['points_per_gpu_all']) == 2
    assert two['config']['points_per_gpu'] == two['config']['points_per_gpu_all'][0]
    assert two['parity_check']['ok'] and one['parity_check']['ok']
    assert two['weak_scaling']['points_total'] == 7200
    for d in (one, two):
        assert d['roofline']['bound'] == 'mfma' and d['value'] > 0
    # the N > 1 line says where rank 0's step goes (kernels / gather / un-permutation / copy to the host)
    bd = two['step_breakdown_ms']
    assert bd['backend'] == 'gloo' and bd['kernel_ms'] > 0 and bd['gather_ms'] >= 0 and bd['d2h_ms'] >= 0
    # ... and every rank's share, with the slowest rank named
    pr = bd['per_rank']
    assert len(pr['kernel_ms']) == 2 and min(pr['kernel_ms']) > 0 and pr['points'] == two['config']['points_per_gpu_all']
    assert bd['slowest_rank'] in (0, 1) and bd['slowest_kernel_ms'] == max(pr['kernel_ms'])
    # the shard cuts were moved by measured feedback before the warm-up (3 rounds: 4 measurements, every one a partition of the grid;
    # the cuts in use are the measured ones with the smallest slowest rank)
    rb = two['config']['rebalance']
    assert rb['rounds'] == 3 and len(rb['history']) == 4 and all(sum(h['points']) == 3600 and min(h['kernel_ms']) > 0 for h in rb['history'])
    assert abs(rb['kept_slowest_kernel_ms'] - min(max(h['kernel_ms']) for h in rb['history'])) < 1e-3
    assert two['config']['points_per_gpu_all'] in [h['points'] for h in rb['history']]
    assert 'rebalance' not in one['config']


def test_config5_dry_run_two_ranks_stream_their_shares():
    """BASELINE config 5's multi-GPU half as a dry run: `python bench.py --mode stream --gpus 2` - two ranks sharing the
    one GPU (gloo for the flag reduction), each streaming its share of the batch through its own two device slots; no
    collective on the data path.  Every rank's pairs are checked against the oracle and every rank's clock is reported."""
    d = run_bench('--mode', 'stream', '--gpus', '2', '--pairs', '6', '--steps', '1', '--warmup', '1', '--size', '3000', '--grid', '60',
                  '--check', '32', timeout=1800)
    assert d['n_gpus'] == 2 and d['config']['pairs_total'] == 6 and d['config']['pairs_per_gpu'] == 3
    assert d['parity_check']['ok'] and d['parity_check']['pairs_checked_per_rank'] == 3
    pr = d['step_breakdown_ms']['per_rank']
    assert pr['pairs'] == [3, 3] and min(pr['batch_ms']) > 0 and d['step_breakdown_ms']['slowest_rank'] in (0, 1)
    assert d['value'] > 0 and d['config']['points_per_pair'] == 3600


def test_rccl_code_path_runs_on_one_gpu():
    """The collectives of the N-GPU path on the RCCL backend itself: a one-rank ``nccl`` group (initialised before any other
    GPU work, no launcher, no re-exec) runs the pair broadcast, the all_reduce(MAX) and the index gather of the set-up and
    the per-step gather of the packed result block on device tensors (dist.PackedGatherer(force_collective=True)) - the
    branch that a dry run on one GPU (gloo, two ranks sharing the device) cannot reach.  Results still equal the oracle."""
    d = run_bench('--gpus', '1', '--force-collective', '--steps', '5', '--warmup', '2', '--size', '3000', '--grid', '60',
                  '--no-cpu-baseline')
    bd = d['step_breakdown_ms']
    assert bd['backend'] == 'nccl'
    # one kernel (sid_pm_unpermute) reads the gathered block and writes the pinned host buffer: no separate copy
    assert bd['kernel_ms'] > 0 and bd['gather_ms'] > 0 and bd['unpermute_ms'] > 0 and bd['d2h_ms'] >= 0
    assert 'sid_pm_unpermute' in bd['results'] and bd['per_rank']['points'] == [3600] and bd['slowest_rank'] == 0
    assert d['parity_check']['ok'] and d['config']['points_total'] == 3600


def test_config4_ft_feeding_pm_on_the_full_size_pair():
    """BASELINE config 4 (ftlib.py:236-281 feeding seaicedrift.py:62-88): get_drift_FT with the GPU detector and matcher ->
    get_drift_PM on the 10000x10000 pair.  >= 99 % of the feature-tracking vectors lie within 3 px of the synthetic
    displacement field, and the PM result on the FT-derived first guess equals the C oracle on a 400-point subsample."""
    d = run_bench('--mode', 'ftpm', '--steps', '2', '--warmup', '1', '--check', '400', timeout=2400)
    assert d['ft_vectors'] > 10000 and d['ft_vectors_within_3px_of_truth'] >= 0.99
    assert d['parity_check']['ok'] and d['parity_check']['points'] == 400
    assert d['valid_grid_points'] > 35000 and d['median_abs_drift_error_px'] < 4.0
    assert d['ft_ms'] > 0 and d['pm_ms'] > 0 and d['value'] > 0


def test_config5_stream_16_pairs_full_size():
    """16 pairs of 10000x10000 px streamed through the two device slots; 3 pairs checked against the oracle."""
    d = run_bench('--mode', 'stream', '--pairs', '16', '--steps', '1', '--warmup', '1', '--check', '32', timeout=2400)
    assert d['config']['pairs_total'] == 16 and d['config']['pairs_per_gpu'] == 16
    assert d['parity_check']['ok'] and d['parity_check']['pairs_checked_per_rank'] == 3
    assert d['value'] > 0 and d['config']['points_per_pair'] == 40000


def test_pair_of_another_shape_after_set_points(pm_ctx, c_oracle):
    """The launch classes are sized for the image-2 shape current at set_points; a pair of another shape selected
    afterwards must be re-classified (no out-of-bounds LDS, correct results), not silently mis-sized."""
    small = syn.make_pair(300, 300, seed=5)
    big = syn.make_pair(700, 700, seed=6)
    angles = list(range(-7, 8))
    rot = my.rotation_table(angles, 0.0, 34)
    # points that are invalid on the small pair (outside image 2) and valid on the big one
    c = np.array([150.0, 400.0, 520.0, 150.0])
    r = np.array([150.0, 380.0, 500.0, 160.0])
    b = np.array([20.0, 50.0, 35.0, 45.0])
    pm_ctx.upload_pair(small[0], small[1], slot=0)
    pm_ctx.select_pair(0)
    pm_ctx.set_points(c, r, c, r, b, 34, 0.0, angles, rot=rot)
    pm_ctx.run()
    got_small, _ = pm_ctx.fetch()
    exp_small, _ = c_oracle.pm_batch(small[0], small[1], c, r, c, r, b, 34, 0.0, angles, rot=rot)
    np.testing.assert_array_equal(got_small[:, :4], exp_small[:, :4])
    assert np.isnan(got_small[1:3]).all()
    pm_ctx.upload_pair(big[0], big[1], slot=1)
    pm_ctx.select_pair(1)
    pm_ctx.run()                                    # no set_points in between
    got_big, _ = pm_ctx.fetch()
    exp_big, _ = c_oracle.pm_batch(big[0], big[1], c, r, c, r, b, 34, 0.0, angles, rot=rot)
    np.testing.assert_array_equal(got_big[:, :4], exp_big[:, :4])
    np.testing.assert_allclose(got_big[:, 4], exp_big[:, 4], rtol=1e-5, atol=1e-5)
    assert np.isfinite(got_big).all()


def test_pm_dispatch_on_a_reused_handle_uses_the_uploaded_pair(c_oracle):
    """A handle whose current pair is a borrowed binding (or the other slot) must match the images handed to
    pm_dispatch, not the old ones."""
    import torch
    a = syn.make_pair(400, 400, seed=17)
    other = syn.make_pair(400, 400, seed=99)
    g = syn.make_grid(400, 400, 5, margin=90)
    angles = [-3, 0, 3]
    exp, _ = c_oracle.pm_batch(a[0], a[1], g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, angles,
                               rot=my.rotation_table(angles, 0.0, 34))
    with _capi.PMContext(0) as ctx:
        t1, t2 = torch.from_numpy(other[0]).cuda(), torch.from_numpy(other[1]).cuda()
        ctx.bind_pair_tensors(t1, t2)               # borrowed binding is current
        got = my.pm_dispatch(a[0], a[1], g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, context=ctx,
                             angles=angles)
        np.testing.assert_array_equal(got[:, :4], exp[:, :4])
        ctx.upload_pair(other[0], other[1], slot=1)
        ctx.select_pair(1)                          # the other slot is current
        got = my.pm_dispatch(a[0], a[1], g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, context=ctx,
                             angles=angles)
        np.testing.assert_array_equal(got[:, :4], exp[:, :4])


def test_two_threads_share_the_device_handle_without_mixing_results(c_oracle):
    """SURVEY.md section 5 "state per call": two threads call pm_dispatch (the shared per-device handle) with two
    DIFFERENT pairs and point sets at the same time, many times over; every call must return its own pair's
    results (= the C oracle), never the other thread's.  Without the per-device lock the calls interleave
    upload_pair / set_points / run on one handle."""
    import threading
    cases = []
    for seed, n_side, angles in ((17, 6, [-3, 0, 3]), (99, 7, list(range(-3, 4)))):
        img = syn.make_pair(400, 400, seed=seed)
        g = syn.make_grid(400, 400, n_side, margin=90)
        exp, _ = c_oracle.pm_batch(img[0], img[1], g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, angles,
                                   rot=my.rotation_table(angles, 0.0, 34), nthreads=4)
        cases.append((img, g, angles, exp))
    errors = []
    start = threading.Barrier(2)

    def worker(k):
        img, g, angles, exp = cases[k]
        try:
            start.wait()
            for _ in range(25):
                got = my.pm_dispatch(img[0], img[1], g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0,
                                     angles=angles)
                assert got.shape == exp.shape
                np.testing.assert_array_equal(got[:, :4], exp[:, :4])
                np.testing.assert_allclose(got[:, 4], exp[:, 4], rtol=1e-5, atol=1e-5, equal_nan=True)
        except BaseException as e:                  # noqa: reported by the main thread
            errors.append((k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    my.release_contexts()                           # the handles come back on the next call
    img, g, angles, exp = cases[0]
    got = my.pm_dispatch(img[0], img[1], g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, angles=angles)
    np.testing.assert_array_equal(got[:, :4], exp[:, :4])


@pytest.mark.parametrize('nang', [15, 3])
def test_full_size_properties_order_subset_and_translation(nang):
    """BASELINE's full size (200x200 grid on a 10000x10000 pair, mixed borders) through properties that need no oracle run of
    that size: (a) the result of a point does not depend on the ORDER of the points (launch classes, XCD-aware order, buckets
    of the four-per-CU build for 3 angles); (b) nor on which OTHER points are in the call; (c) cropping both images and moving
    the coordinates along moves c2, r2 by the same offset and leaves a, r, h alone.  All bit for bit."""
    from sea_ice_drift_amd.pmlib import rotation_table
    size, s = 10000, 34
    angles = list(range(-(nang // 2), nang // 2 + 1))
    img1, img2 = syn.make_pair(size, size)
    g = syn.make_grid(size, size, 200)
    rot = rotation_table(angles, 0.0, s)
    names = ('c1', 'r1', 'c2fg', 'r2fg', 'border')
    rng = np.random.default_rng(2026)
    with _capi.PMContext(0) as ctx:
        ctx.upload_pair(img1, img2)

        def run(sel, off=(0.0, 0.0)):
            v = [g[k][sel] for k in names]
            ctx.set_points(v[0] - off[0], v[1] - off[1], v[2] - off[0], v[3] - off[1], v[4], s, 0.0, angles, rot=rot)
            ctx.run()
            return ctx.fetch()
        base, base_ij = run(slice(None))
        ok = np.isfinite(base[:, 0])
        assert ok.sum() > 39000
        perm = rng.permutation(len(base))
        got, got_ij = run(perm)
        np.testing.assert_array_equal(got, base[perm])
        np.testing.assert_array_equal(got_ij, base_ij[perm])
        sub = np.sort(rng.choice(len(base), 5000, replace=False))
        got, got_ij = run(sub)
        np.testing.assert_array_equal(got, base[sub])
        np.testing.assert_array_equal(got_ij, base_ij[sub])
        # (c) crop 640 columns and 512 rows off the top left of both images (keeps the 16-byte alignment of the rows)
        oc, orr = 640, 512
        inner = np.nonzero((g['c1'] - oc > 200) & (g['r1'] - orr > 200) & (g['c2fg'] - oc > 200) & (g['r2fg'] - orr > 200))[0]
        ctx.upload_pair(np.ascontiguousarray(img1[orr:, oc:]), np.ascontiguousarray(img2[orr:, oc:]))
        got, got_ij = run(inner, off=(float(oc), float(orr)))
        exp = base[inner].copy()
        exp[:, 0] -= oc; exp[:, 1] -= orr
        np.testing.assert_array_equal(got, exp)
        np.testing.assert_array_equal(got_ij, base_ij[inner])
