"""Key-point detector (the ORB half): the NumPy oracle of the package's own specification on the CPU,
and on the GPU the kernels against that oracle bit for bit, then the whole feature-tracking chain on a synthetic
pair with a known displacement field.  Parity with OpenCV's ORB is unpinned by construction (cv2 absent)."""
import numpy as np
import pytest

from oracle import orb_oracle as oo
from sea_ice_drift_amd import orb, synthetic as syn


def test_tables_are_well_formed():
    d = orb.direction_table()
    assert d.shape == (32, 2) and d[0].tolist() == [16384, 0] and d[8].tolist() == [0, 16384]
    p = orb.rotated_pattern()
    assert p.shape == (32, 256, 4) and p.dtype == np.int8
    assert (np.abs(p.astype(int)) <= 14).all()
    assert not ((p[0, :, 0] == p[0, :, 2]) & (p[0, :, 1] == p[0, :, 3])).any()
    # direction 8 is a quarter turn: (x, y) -> (-y, x)
    np.testing.assert_array_equal(p[8, :, 0], -p[0, :, 1])
    np.testing.assert_array_equal(p[8, :, 1], p[0, :, 0])


def test_oracle_finds_planted_corners_and_their_orientation():
    img = np.full((200, 220), 60, dtype=np.int64)
    img[80:, 100:] = 200                                           # one bright quadrant: a corner at (100, 80)
    # (a little seeded noise: on an ideal step the FAST scores form a plateau and the strict 3x3 maximum rejects it)
    img = (img + np.random.default_rng(3).integers(-6, 7, img.shape)).astype(np.uint8)
    xy, meta, resp, desc = oo.detect_and_compute(img, orb.rotated_pattern(), orb.direction_table(), n_levels=1,
                                                 n_features=50)
    assert len(xy) >= 1
    d = np.hypot(xy[:, 0] - 100, xy[:, 1] - 80)
    k = int(np.argmin(d))
    assert d[k] <= 3
    # the bright mass lies towards +x, +y: direction index near 4 (45 degrees)
    assert meta[k, 3] in (3, 4, 5)
    assert desc.shape[1] == 32 and (resp[:-1] >= resp[1:]).all()


def test_oracle_level_split_matches_opencv_rule():
    sc, lr, lc, want = oo.level_geometry(1000, 800, 7, 1.2, 1000)
    assert sum(want) == 1000 and want[0] > want[1] > want[5] and lr[0] == 1000 and lc[1] == 667
    np.testing.assert_allclose(sc[3], float(np.float32(1.2)) ** 3, rtol=1e-15)


@pytest.mark.gpu
def test_kernels_equal_the_oracle():
    img = syn.make_pair(700, 640, seed=91)[0]
    for kw in (dict(n_features=1500, n_levels=4), dict(n_features=300, n_levels=2, fast_threshold=35, patch_size=31)):
        exp_xy, exp_meta, exp_resp, exp_desc = oo.detect_and_compute(img, orb.rotated_pattern(), orb.direction_table(), **kw)
        xy, desc, meta, resp = orb.detect_and_compute(img, full=True, **kw)
        assert len(exp_xy) > 100
        np.testing.assert_array_equal(meta, exp_meta)
        np.testing.assert_array_equal(resp, exp_resp)
        np.testing.assert_array_equal(xy.astype(np.float32), exp_xy)
        np.testing.assert_array_equal(desc, exp_desc)


@pytest.mark.gpu
def test_feature_tracking_end_to_end_with_the_gpu_detector():
    """ftlib.feature_tracking with the package's detector + the GPU matcher: the matched vectors follow the synthetic
    displacement field (ftlib.py:241-285 chain, config 4 of BASELINE.json in miniature)."""
    from sea_ice_drift_amd import ftlib
    from sea_ice_drift_amd.domain import ArrayNansat
    img1, img2 = syn.make_pair(2000, 2000, seed=55, speckle=0.03)
    n1, n2 = ArrayNansat(img1, matrix=((1e-4, 0.0), (0.0, 1e-4))), ArrayNansat(img2, matrix=((1e-4, 0.0), (0.0, 1e-4)))
    x1, y1, x2, y2 = ftlib.feature_tracking(n1, n2, nFeatures=20000, max_drift=1e9, ratio_test=0.75)
    assert len(x1) > 300
    dc, dr = syn.true_displacement(x1, y1)
    err = np.hypot(x2 - x1 - dc, y2 - y1 - dr)
    assert (err < 3.0).mean() > 0.9


@pytest.mark.gpu
def test_detector_calls_from_several_threads_share_nothing():
    """sid_orb_detect takes a workspace (buffers + stream) per call: four threads on four different images, three rounds,
    give what the same calls give one after the other (ftlib.track runs the two images of a pair this way)."""
    from concurrent.futures import ThreadPoolExecutor
    imgs = [syn.make_pair(900 + 64 * k, 1000 - 32 * k, seed=70 + k)[k & 1] for k in range(4)]
    kw = dict(n_features=4000, n_levels=5)
    ref = [orb.detect_and_compute(im, full=True, **kw) for im in imgs]
    for _ in range(3):
        with ThreadPoolExecutor(max_workers=4) as pool:
            got = list(pool.map(lambda im: orb.detect_and_compute(im, full=True, **kw), imgs))
        for g, e in zip(got, ref):
            assert len(e[0]) > 500
            for a, b in zip(g, e):
                np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_selection_and_ordering_on_the_device_equal_the_host_route(monkeypatch):
    """Round 4: the levels run on the device from end to end (radix selection of the level's share by a one-wavefront pick,
    exact (response, y, x) order by counting) with one synchronisation per image; SID_ORB_HOST_SELECT=1 is the previous route
    (counts and histograms to the host, std::sort there).  Same key points in the same order, bit for bit - more features than
    candidates, fewer, a cap by max_out, plateaus of equal responses - and both equal the oracle."""
    rng = np.random.default_rng(8)
    big = syn.make_pair(1500, 1300, seed=33)[1]
    flat = np.full((600, 600), 90, dtype=np.uint8)
    flat[::40, :] = 200; flat[:, ::40] = 200                           # a lattice: thousands of equal responses
    noise = rng.integers(0, 256, (500, 700)).astype(np.uint8)
    for name, img, kw in (('big', big, dict(n_features=30000, n_levels=7)), ('big-few', big, dict(n_features=200, n_levels=3)),
                          ('lattice', flat, dict(n_features=5000, n_levels=4)), ('noise', noise, dict(n_features=100000, n_levels=5, fast_threshold=10))):
        monkeypatch.delenv('SID_ORB_HOST_SELECT', raising=False)
        dev = orb.detect_and_compute(img, full=True, **kw)
        monkeypatch.setenv('SID_ORB_HOST_SELECT', '1')
        host = orb.detect_and_compute(img, full=True, **kw)
        monkeypatch.delenv('SID_ORB_HOST_SELECT')
        assert len(host[0]) > 50, name
        for a, b in zip(dev, host):
            np.testing.assert_array_equal(a, b, err_msg=name)
    exp_xy, exp_meta, exp_resp, exp_desc = oo.detect_and_compute(flat, orb.rotated_pattern(), orb.direction_table(), n_features=5000, n_levels=4)
    xy, desc, meta, resp = orb.detect_and_compute(flat, full=True, n_features=5000, n_levels=4)
    np.testing.assert_array_equal(meta, exp_meta)
    np.testing.assert_array_equal(resp, exp_resp)
    np.testing.assert_array_equal(desc, exp_desc)
