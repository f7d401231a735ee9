"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on seeded inputs."""
import numpy as np
import pytest

from sea_ice_drift_amd import _capi, synthetic as syn
from oracle import pm_oracle as po

pytestmark = pytest.mark.gpu

ANGLES7 = list(range(-3, 4))
ANGLES15 = list(range(-7, 8))


def rot_for(angles, alpha0, s):
    return np.array([po.rotation_terms(a - alpha0, s) for a in angles])


def assert_parity(got, got_ij, exp, exp_ij, mcc_norm=False):
    """Integer outputs bit-exact; c2, r2, a exact; r exact (float32 spec; within 1e-5 when mcc_norm divides it by a float32
    standard deviation, as in test_hessian_options_incl_gaussian_smoothing); h within 1e-5."""
    np.testing.assert_array_equal(got_ij, exp_ij)
    nan_e = np.isnan(exp[:, 0])
    np.testing.assert_array_equal(np.isnan(got[:, 0]), nan_e)
    g, e = got[~nan_e], exp[~nan_e]
    np.testing.assert_array_equal(g[:, :3], e[:, :3])
    if mcc_norm:
        np.testing.assert_allclose(g[:, 3], e[:, 3], rtol=1e-5, atol=1e-5)
    else:
        np.testing.assert_array_equal(g[:, 3], e[:, 3])
    np.testing.assert_allclose(g[:, 4], e[:, 4], rtol=1e-5, atol=1e-5)


def test_device_double_rsqrt_is_ieee(pm_ctx):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.integers(1, 2**40, 200000).astype(np.float64),
                        rng.random(100000) * 1e12 + 1.0, [1.0, 2.0, 3.0, 4.0, 578.0, 1156.0]])
    y = pm_ctx.debug_rsqrt(x)
    np.testing.assert_array_equal(y, 1.0 / np.sqrt(x))


@pytest.mark.parametrize('s,alpha0,angles', [(34, 0.0, ANGLES7), (35, -3.85, [-3, 0, 3]), (34, 0.0, ANGLES15)])
def test_small_pair_matches_oracle(pm_ctx, c_oracle, s, alpha0, angles):
    img1, img2 = syn.make_pair(600, 600, seed=7)
    img1 = img1.copy()
    img1[300:320, 300:330] = 0                       # invalid patch -> NaN path (pmlib.py:152-154)
    g = syn.make_grid(600, 600, 12, margin=90)
    rot = rot_for(angles, alpha0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, alpha0,
                                    angles, rot=rot, nthreads=8)
    assert np.isnan(exp[:, 0]).sum() >= 1
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, alpha0, angles, rot=rot)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    assert_parity(got, got_ij, exp, exp_ij)


@pytest.mark.parametrize('s,angles', [(34, ANGLES7), (34, [-3, 0, 3]), (35, [-3, 0, 3])])
def test_four_per_cu_build_equals_the_three_per_cu_build_and_the_oracle(pm_ctx, c_oracle, monkeypatch, s, angles):
    """Slot-group layouts: borders 20 .. 23 run the three-wavefront class (four points per CU with the 168-VGPR build; round 4);
    without it (SID_PM_NO_W3) borders 20 and 21 run the 128-VGPR build of round 3 (pm_kernel_rp_occ4.hip), without that
    (SID_PM_NO_OCC4) three per CU; borders 22 .. 30 exercise the short operand table in the other launch classes.  All three
    against the oracle, and bit-identical to each other (the switches are read at every set_points)."""
    img1, img2 = syn.make_pair(900, 900, seed=11)
    g = syn.make_grid(900, 900, 14, margin=120)
    n = len(g['c1'])
    border = np.array([20, 21, 20, 22, 26, 21, 30, 23][:8] * (n // 8 + 1), dtype=np.float64)[:n]
    rot = rot_for(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], border, s, 0.0, angles, rot=rot,
                                    nthreads=8)
    pm_ctx.upload_pair(img1, img2)
    res = []
    for envs in ((), ('SID_PM_NO_W3',), ('SID_PM_NO_W3', 'SID_PM_NO_OCC4')):
        for k in ('SID_PM_NO_W3', 'SID_PM_NO_OCC4'):
            monkeypatch.delenv(k, raising=False)
        for k in envs:
            monkeypatch.setenv(k, '1')
        pm_ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], border, s, 0.0, angles, rot=rot)
        pm_ctx.run()
        got, got_ij = pm_ctx.fetch()
        assert_parity(got, got_ij, exp, exp_ij)
        res.append((got.copy(), got_ij.copy(), pm_ctx.work_info()['launches']))
    for k in ('SID_PM_NO_W3', 'SID_PM_NO_OCC4'):
        monkeypatch.delenv(k, raising=False)
    for r in res[1:]:
        np.testing.assert_array_equal(res[0][0], r[0])
        np.testing.assert_array_equal(res[0][1], r[1])
    assert res[1][2] > res[2][2]                     # borders 20 and 21 were a launch of their own in round 3's classes


def test_sampling_table_flagged_entries_and_fractional_centres(pm_ctx, c_oracle):
    """The offset table serves integral template centres; entries whose coordinate sits on a rounding
    boundary are flagged and recomputed per point (forced here with tc.T = 17.5: every row coordinate is
    k + 1/2), and fractional centres / templates at the image border take the on-the-fly path.  All must
    equal the oracle (get_template, pmlib.py:105-113)."""
    img1, img2 = syn.make_pair(600, 600, seed=21)
    g = syn.make_grid(600, 600, 10, margin=90)
    s, angles = 34, [-2.0, 0.0, 2.0]
    rot = rot_for(angles, 0.0, s)
    rot[1] = [1.0, 0.0, 17.5, 18.0]                  # caller-supplied terms: rr = i - 17.5 + r1 -> floor(rr + .5) on the boundary
    c1, r1 = g['c1'].copy(), g['r1'].copy()
    c1[::3] += 0.3                                   # fractional centres (on-the-fly path)
    r1[::3] -= 0.45
    c1[1], r1[1] = 10.0, 300.0                       # patch reaches over the left image edge -> zero pixels -> NaN
    c1[2], r1[2] = 40.0, 41.0                        # integral centre, patch not inside the image: on-the-fly path
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, g['c2fg'], g['r2fg'], g['border'], s, 0.0, angles, rot=rot,
                                    nthreads=8)
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(c1, r1, g['c2fg'], g['r2fg'], g['border'], s, 0.0, angles, rot=rot)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    assert_parity(got, got_ij, exp, exp_ij)
    assert np.isnan(exp[1, 0]) and (exp_ij[:, 2] == 1).any()           # the edge case fired, the boundary angle won somewhere


def test_debug_point_intermediates(pm_ctx):
    img1, img2 = syn.make_pair(400, 400, seed=11)
    pm_ctx.upload_pair(img1, img2)
    s, alpha0, angles = 34, 0.0, ANGLES7
    rot = rot_for(angles, alpha0, s)
    d = pm_ctx.debug_point(200.0, 180.0, 203.0, 178.0, 23.0, s, alpha0, angles, rot=rot)
    for k, a in enumerate(angles):
        np.testing.assert_array_equal(d['templates'][k], po.get_template(img1, 200.0, 180.0, a - alpha0, s))
    res, (bij, bk, best_result, _) = po.use_mcc(200.0, 180.0, 203.0, 178.0, 23.0, img1, img2, s, alpha0,
                                               full=True, angles=angles)
    assert tuple(d['ij']) == (bij[0], bij[1], bk)
    np.testing.assert_array_equal(d['ccm'], best_result)
    np.testing.assert_array_equal(d['hes'], po.raw_hessian(best_result))
    np.testing.assert_allclose(d['out'], res, rtol=1e-5, atol=1e-5)


def test_one_shot_batch_entry(c_oracle):
    img1, img2 = syn.make_pair(300, 300, seed=3)
    g = syn.make_grid(300, 300, 4, margin=80, border=20)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0,
                                    ANGLES7, rot=rot_for(ANGLES7, 0.0, 34))
    got, got_ij = _capi.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0,
                                 ANGLES7, rot=rot_for(ANGLES7, 0.0, 34))
    assert_parity(got, got_ij, exp, exp_ij)


@pytest.mark.parametrize('s,nang', [(21, 5), (33, 7), (36, 15), (49, 3), (34, 17), (35, 31), (51, 7), (64, 15), (64, 3)])
def test_other_template_sizes_and_angle_counts(pm_ctx, c_oracle, s, nang):
    """Generic-size code path (s not 34/35; up to 64 = the K of one matrix instruction, round 4) and more than 15 angles
    (several template groups)."""
    size = 500 if s < 50 else 700
    img1, img2 = syn.make_pair(size, size, seed=21)
    g = syn.make_grid(size, size, 7, margin=95 if s < 50 else 140, border=20 if 40 < s < 50 else 'mixed')
    half = nang // 2
    angles = [0.5 * k for k in range(-half, nang - half)]
    rot = rot_for(angles, 1.25, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 1.25,
                                    angles, rot=rot, nthreads=8)
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 1.25, angles, rot=rot)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    assert_parity(got, got_ij, exp, exp_ij)


def test_ties_and_degenerate_inputs(pm_ctx, c_oracle):
    """Exact ties (periodic image: many placements with r == 1), a constant template (r == 1
    everywhere -> first placement wins) and a flat window (r == 0 everywhere)."""
    yy, xx = np.mgrid[0:400, 0:400]
    periodic = (1 + ((yy % 8) * 8 + (xx % 8)) * 3).astype(np.uint8)      # period 8 in both directions
    flat1 = periodic.copy()
    flat1[100:200, 100:200] = 77                                        # constant template region
    flat2 = periodic.copy()
    flat2[250:400, 250:400] = 9                                         # flat search window region
    pts = dict(c1=[300.0, 150.0, 60.0], r1=[60.0, 150.0, 300.0], c2fg=[301.0, 150.0, 320.0],
               r2fg=[62.0, 150.0, 320.0], border=[20.0, 22.0, 21.0])
    angles = [-3.0, 0.0, 3.0]
    rot = rot_for(angles, 0.0, 34)
    v = [pts[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    exp, exp_ij = c_oracle.pm_batch(flat1, flat2, *v, 34, 0.0, angles, rot=rot)
    pm_ctx.upload_pair(flat1, flat2)
    pm_ctx.set_points(*v, 34, 0.0, angles, rot=rot)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    np.testing.assert_array_equal(got_ij, exp_ij)
    np.testing.assert_array_equal(got[:, :4], exp[:, :4])
    assert exp[0, 3] == 1.0 and exp[1, 3] == 1.0 and exp[2, 3] == 0.0
    assert tuple(exp_ij[1]) == (0, 0, 0) and tuple(exp_ij[2]) == (0, 0, 0)


def test_pair_streaming_through_two_slots(pm_ctx, c_oracle):
    """BASELINE config 5 in miniature: pairs streamed back to back through the two device slots (upload of
    pair k+1 on the copy stream while pair k is matched) give the results of matching each pair alone."""
    import torch
    pairs = [syn.make_pair(500, 500, seed=40 + k) for k in range(5)]
    pinned = [[torch.from_numpy(im).pin_memory() for im in p] for p in pairs]     # pinned: the copies are asynchronous
    g = syn.make_grid(500, 500, 8, margin=90)
    rot = rot_for(ANGLES7, 0.0, 34)
    pm_ctx.upload_pair(pinned[0][0].numpy(), pinned[0][1].numpy(), slot=0)
    pm_ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, ANGLES7, rot=rot)
    results = []
    for k in range(len(pairs)):
        pm_ctx.select_pair(k % 2)
        if k + 1 < len(pairs):
            pm_ctx.upload_pair(pinned[k + 1][0].numpy(), pinned[k + 1][1].numpy(), slot=(k + 1) % 2, select=False)
        pm_ctx.run()
        results.append(pm_ctx.fetch())
    for k, (img1, img2) in enumerate(pairs):
        exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0,
                                        ANGLES7, rot=rot, nthreads=8)
        assert_parity(results[k][0], results[k][1], exp, exp_ij)
    assert not np.array_equal(results[0][1], results[1][1])            # the pairs really differ


@pytest.mark.parametrize('flags', [2, 3, 7, 6])
def test_hessian_options_incl_gaussian_smoothing(pm_ctx, c_oracle, flags):
    """hes_smth (scipy gaussian_filter sigma 1 before the Hessian, pmlib.py:46-47) alone and combined with
    hes_norm / mcc_norm: flags bit 0 = hes_norm, bit 1 = hes_smth, bit 2 = mcc_norm."""
    img1, img2 = syn.make_pair(600, 600, seed=55)
    g = syn.make_grid(600, 600, 9, margin=90)
    rot = rot_for(ANGLES7, 0.0, 34)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0,
                                    ANGLES7, rot=rot, flags=flags, nthreads=8)
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, ANGLES7, rot=rot, flags=flags)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    np.testing.assert_array_equal(got_ij, exp_ij)
    np.testing.assert_array_equal(got[:, :3], exp[:, :3])
    np.testing.assert_allclose(got[:, 3], exp[:, 3], rtol=1e-5, atol=1e-5)      # r: exact unless mcc_norm
    np.testing.assert_allclose(got[:, 4], exp[:, 4], rtol=1e-5, atol=1e-5)      # h


@pytest.mark.gpu
@pytest.mark.parametrize('s', [34, 35])
def test_shortened_normalisation_is_the_specification(pm_ctx, s):
    """The winner's NCC matrix uses exact_from_sums_fast (pm_kernel_mfma.hip): 2^27 pseudo-random sums, bit for bit
    against the specification's IEEE route; the guard must fire (perfect matches, float32 rounding boundaries)."""
    n, bad, slow = pm_ctx.debug_ncc_selftest(1 << 27, img_size=s, seed=20240917 + s)
    assert n >= 1 << 27
    assert bad == 0
    assert 0 < slow < n // 20                                          # (the generator over-samples flat windows)


@pytest.mark.gpu
def test_shortened_hypot_is_numpys_float32_hypot(pm_ctx):
    """ph_hessian_fast takes sqrt(x^2 + y^2) from v_rsq_f64 + Newton steps with a rounding-boundary guard: 2^27
    pseudo-random float32 pairs (incl. zeros, denormals, huge values, perfect squares) on the device, bit for bit
    against (float)sqrt((double)x*x + (double)y*y) - what np.hypot does for float32 (pmlib.py:55)."""
    n, bad = pm_ctx.debug_hypot_selftest(1 << 27, seed=20261002)
    assert n >= 1 << 27
    assert bad == 0


def test_results_written_straight_into_pinned_host_memory(pm_ctx, c_oracle):
    """bind_results_host: the kernels write into pinned host tensors (zero copy); after run + sync they hold what fetch() returns
    from the device buffers - NaN rows included."""
    img1, img2 = syn.make_pair(700, 700, seed=19)
    img1 = img1.copy()
    img1[350:365, 340:370] = 0                        # NaN rows
    g = syn.make_grid(700, 700, 13, margin=90)
    rot = rot_for(ANGLES15, 0.0, 34)
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], 34, 0.0, ANGLES15, rot=rot)
    pm_ctx.run()
    ref, ref_ij = pm_ctx.fetch()
    assert np.isnan(ref[:, 0]).any()
    out, ij = pm_ctx.bind_results_host()
    out.fill_(-7.0); ij.fill_(-7)
    pm_ctx.run()
    pm_ctx.sync()
    np.testing.assert_array_equal(out.numpy(), ref)
    np.testing.assert_array_equal(ij.numpy(), ref_ij)


@pytest.mark.parametrize('s,angles', [(34, ANGLES15), (34, [-3, 0, 3]), (35, ANGLES7)])
def test_every_launch_class_incl_global_sums_borders_20_to_68(pm_ctx, c_oracle, monkeypatch, s, angles):
    """Round 4: borders 28 .. 47 run launches that keep the per-placement sum of squares in GLOBAL memory (the smaller LDS
    footprint lifts them into the next residency class), the Hessian magnitudes lie over the dead window, and search borders up
    to 68 fit the 160 KB of LDS (58 before).  Every border 20 .. 68 in one call - all launch classes, both forms of the sums -
    against the oracle; then the same points with the global form forced everywhere (SID_PM_ALWAYS_GS) and forbidden wherever an
    LDS instantiation exists (SID_PM_NO_GS): bit-identical results.  A point the device refused would raise in fetch
    (sid_pm_check)."""
    size = 1400
    img1, img2 = syn.make_pair(size, size, seed=23)
    rng = np.random.default_rng(3)
    borders = np.repeat(np.arange(20, 69), 2).astype(np.float64)
    n = len(borders)
    c1 = np.rint(rng.uniform(250, size - 250, n)); r1 = np.rint(rng.uniform(250, size - 250, n))
    dc, dr = syn.true_displacement(c1, r1)
    c2 = c1 + np.rint(dc) + rng.integers(-2, 3, n); r2 = r1 + np.rint(dr) + rng.integers(-2, 3, n)
    rot = rot_for(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot, nthreads=8)
    assert np.isfinite(exp[:, 0]).sum() > n * 0.9
    pm_ctx.upload_pair(img1, img2)
    results = []
    for env in (None, 'SID_PM_ALWAYS_GS', 'SID_PM_NO_GS', 'SID_PM_NO_GSI'):
        # (SID_PM_NO_GSI: sum w' per placement is not kept in the global-memory blocks - the winner multiplies the all-ones operand)
        for k in ('SID_PM_ALWAYS_GS', 'SID_PM_NO_GS', 'SID_PM_NO_GSI'):
            monkeypatch.delenv(k, raising=False)
        if env:
            monkeypatch.setenv(env, '1')
        pm_ctx.set_points(c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot)
        pm_ctx.run()
        got, got_ij = pm_ctx.fetch()
        assert_parity(got, got_ij, exp, exp_ij)
        results.append((got, got_ij))
    for got, got_ij in results[1:]:
        np.testing.assert_array_equal(got_ij, results[0][1])
        np.testing.assert_array_equal(got[:, :4], results[0][0][:, :4])
    for k in ('SID_PM_ALWAYS_GS', 'SID_PM_NO_GS', 'SID_PM_NO_GSI'):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize('s,angles,flags', [(34, ANGLES15, 1), (35, [-3, 0, 3], 1), (34, ANGLES7, 7), (35, [0.5 * k for k in range(-8, 9)], 3)])
def test_borders_69_to_111_keep_their_tables_in_global_memory(pm_ctx, c_oracle, monkeypatch, s, angles, flags):
    """Round 4: beyond border 68 the per-placement tables of a point (sum w'^2, the row sums, the NCC matrix of the winning angle,
    the Hessian magnitudes) no longer fit the 160 KB of LDS next to the window; those points run `pm_kernel_rp<S, 4, 0, 0, true>`,
    which keeps them in the point's block of global memory.  Borders up to 111 (reference kwarg max_border; its default is 50), any angle set (slot-group
    sets run the full-table kernel there), both Hessian routes, several groups of angles."""
    size = 1700
    img1, img2 = syn.make_pair(size, size, seed=29)
    rng = np.random.default_rng(4)
    borders = np.concatenate([np.arange(66, 112, 3 if len(angles) > 7 else 2), [111, 111]]).astype(np.float64)
    n = len(borders)
    c1 = np.rint(rng.uniform(420, size - 420, n)); r1 = np.rint(rng.uniform(420, size - 420, n))
    dc, dr = syn.true_displacement(c1, r1)
    c2 = c1 + np.rint(dc) + rng.integers(-2, 3, n); r2 = r1 + np.rint(dr) + rng.integers(-2, 3, n)
    rot = rot_for(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot, nthreads=8, flags=flags)
    assert np.isfinite(exp[:, 0]).sum() > n * 0.9
    pm_ctx.upload_pair(img1, img2)
    pm_ctx.set_points(c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot, flags=flags)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    assert_parity(got, got_ij, exp, exp_ij, mcc_norm=bool(flags & 4))
    # one more border no longer fits one workgroup's LDS: since round 6 such a point runs the large-window pipeline
    # (csrc/pm_large.hip) in the same batch - same results as the oracle, no error
    b2 = np.array([112.0, 70.0, 130.0])
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1[:3], r1[:3], c2[:3], r2[:3], b2, s, 0.0, angles, rot=rot, nthreads=8, flags=flags)
    pm_ctx.set_points(c1[:3], r1[:3], c2[:3], r2[:3], b2, s, 0.0, angles, rot=rot, flags=flags)
    pm_ctx.run()
    got, got_ij = pm_ctx.fetch()
    assert_parity(got, got_ij, exp, exp_ij, mcc_norm=bool(flags & 4))


@pytest.mark.parametrize('s,angles', [(34, ANGLES15), (35, ANGLES15), (35, [-3, 0, 3]), (34, ANGLES7)])
def test_three_wavefronts_per_point_class_equals_four_and_the_oracle(pm_ctx, c_oracle, monkeypatch, s, angles):
    """Round 4: the smallest windows (borders 20 .. 23, window pitch 104) run FOUR points per CU with three wavefronts each
    (192 threads, sums in global memory: pm_kernel_rp<S, 4, *, 1104>); SID_PM_NO_W3=1 is round 3's classes of 256 threads.  Both equal the oracle, bit for bit each other - also with the on-the-fly sampler (fractional centres) and at the
    image edge, where the patch cannot ride with the window."""
    size = 900
    img1, img2 = syn.make_pair(size, size, seed=41)
    rng = np.random.default_rng(6)
    borders = np.repeat(np.arange(20, 25), 24).astype(np.float64)
    n = len(borders)
    c1 = np.rint(rng.uniform(60, size - 60, n)); r1 = np.rint(rng.uniform(60, size - 60, n))
    c1[::7] += 0.37; r1[::5] -= 0.41                                   # fractional template centres: the general sampler
    c1[:4] = [30.0, size - 31.0, 400.0, 400.0]; r1[:4] = [400.0, 400.0, 30.0, size - 31.0]   # template patch cut by the image edge
    dc, dr = syn.true_displacement(c1, r1)
    c2 = np.clip(c1 + np.rint(dc) + rng.integers(-2, 3, n), 70, size - 70); r2 = np.clip(r1 + np.rint(dr) + rng.integers(-2, 3, n), 70, size - 70)
    rot = rot_for(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot, nthreads=8)
    assert np.isfinite(exp[:, 0]).sum() > n * 0.8
    pm_ctx.upload_pair(img1, img2)
    res = []
    for env in (None, 'SID_PM_NO_W3'):
        monkeypatch.delenv('SID_PM_NO_W3', raising=False)
        if env:
            monkeypatch.setenv(env, '1')
        pm_ctx.set_points(c1, r1, c2, r2, borders, s, 0.0, angles, rot=rot)
        per_cu = _capi.estimate_residency(borders[::24], s, len(angles))
        pm_ctx.run()
        got, got_ij = pm_ctx.fetch()
        assert_parity(got, got_ij, exp, exp_ij)
        res.append((got, got_ij, per_cu))
    monkeypatch.delenv('SID_PM_NO_W3', raising=False)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][0][:, :4], res[1][0][:, :4])
    # the launch classes the host predicts: four per CU up to border 23 (without the class: the full table three per CU)
    cls = [int(v) & 15 for v in res[0][2]]
    assert cls[:3] == [4, 4, 4] and cls[4] == 3 and cls[3] in (3, 4)    # (border 23 at 35 px: one granule too many)
    if len(angles) > 7:
        assert [int(v) & 15 for v in res[1][2]] == [3, 3, 3, 3, 3]
