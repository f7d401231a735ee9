"""GPU parity of what round 6 changed inside the one-workgroup-per-point kernels: recycled blocks of global memory (bitmap free
lists, PMArgs::ring) for EVERY launch that keeps per-placement tables there - against exclusive blocks and against the oracle."""
import numpy as np
import pytest

from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('s,angles,flags', [(34, list(range(-7, 8)), 1), (35, list(range(-7, 8)), 1), (34, list(range(-3, 4)), 1),
                                            (35, [-3, 0, 3], 1), (34, list(range(-7, 8)), 7)])
def test_recycled_blocks_equal_exclusive_blocks_and_the_oracle(pm_ctx, c_oracle, monkeypatch, s, angles, flags):
    """Every launch class (borders 20 .. 50 + two big ones), 15 / 7 / 3 angles: SID_PM_RECYCLE_ALL=1 (shipped), =0 (round 5: only
    the launches that keep accumulators recycle) and SID_PM_NO_RECYCLE=1 (a block per launch position everywhere) give the
    same bits, and those are the oracle's."""
    size = 1100
    img1, img2 = syn.make_pair(size, size, seed=61)
    img1 = img1.copy()
    img1[500:520, 480:530] = 0
    rng = np.random.default_rng(62)
    n = 400
    c1 = np.rint(rng.uniform(200, size - 200, n)); r1 = np.rint(rng.uniform(200, size - 200, n))
    c1[::5] += 0.5
    dc, dr = syn.true_displacement(c1, r1)
    c2 = np.rint(c1 + dc) + rng.integers(-2, 3, n); r2 = np.rint(r1 + dr) + rng.integers(-2, 3, n)
    border = np.array(list(range(20, 51)) * (n // 31 + 1), dtype=np.float64)[:n]
    border[-2:] = [75.0, 90.0]
    c1[-2:] = r1[-2:] = c2[-2:] = r2[-2:] = 550.0
    rot = my.rotation_table(angles, 0.0, s)
    exp, exp_ij = c_oracle.pm_batch(img1, img2, c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, nthreads=8, flags=flags)
    assert np.isnan(exp[:, 0]).any() and np.isfinite(exp[:, 0]).sum() > 0.9 * n
    pm_ctx.upload_pair(img1, img2)
    res = []
    for env in ({'SID_PM_RECYCLE_ALL': '1'}, {'SID_PM_RECYCLE_ALL': '0'}, {'SID_PM_NO_RECYCLE': '1'}):
        for k in ('SID_PM_RECYCLE_ALL', 'SID_PM_NO_RECYCLE'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pm_ctx.set_points(c1, r1, c2, r2, border, s, 0.0, angles, rot=rot, flags=flags)
        for _ in range(3):                                          # (blocks are reused across runs as well)
            pm_ctx.run()
        res.append(pm_ctx.fetch())
    for k in ('SID_PM_RECYCLE_ALL', 'SID_PM_NO_RECYCLE'):
        monkeypatch.delenv(k, raising=False)
    got, ij = res[0]
    np.testing.assert_array_equal(ij, exp_ij)
    nan = np.isnan(exp[:, 0])
    if flags & 4:
        np.testing.assert_array_equal(got[~nan, :3], exp[~nan, :3])
        np.testing.assert_allclose(got[~nan, 3:], exp[~nan, 3:], rtol=1e-5, atol=1e-5)
    else:
        np.testing.assert_array_equal(got[~nan, :4], exp[~nan, :4])
        np.testing.assert_allclose(got[~nan, 4], exp[~nan, 4], rtol=1e-5, atol=1e-5)
    for o, j in res[1:]:
        np.testing.assert_array_equal(o, got)
        np.testing.assert_array_equal(j, ij)
