#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own Python.

Runs only in the build container (needs /root/reference; see oracle/ref_harness.py for the
stub-import recipe).  The fixtures hold numbers only - inputs (or the seeds that regenerate
them, with a sha256) and the outputs the reference's functions returned:

  g1_templates.npz   pmlib.get_template              (pmlib.py:89-115)
  g1b_templates_order1.npz  pmlib.get_template(rot_order=1)  (pmlib.py:89-115, scipy order 1)
  g2_hessian.npz     pmlib.get_hessian               (pmlib.py:36-59)
  g3_use_mcc.npz     pmlib.use_mcc / rotate_and_match (pmlib.py:117-212) with
                     template_matcher=<restated TM_CCOEFF_NORMED> injected through the
                     reference's own plug point (cv2 is absent: parity unpinned at that call)
  g3b_use_mcc_order1.npz  pmlib.use_mcc(rot_order=1) on G3's pair and points
  g1c_templates_spline.npz  pmlib.get_template(rot_order = 2..5): scipy's whole-image spline prefilter + order-n interpolation
                     (pmlib.py:89-115) on a smooth image and on G1's noise image, incl. templates cut by the image border
  g3d_use_mcc_spline.npz  pmlib.use_mcc(rot_order = 2, 3, 5) on G3's pair and points
  g3c_large_rotations.npz  pmlib.use_mcc on G3's pair and points with scene rotations alpha0 of 30, 90 and -137.5 degrees
                     (pmlib.py:79-87,151: the sampling angle is angle - alpha0), rot_order 0 and 1, sizes 34 / 35
  g8_rotate_and_match.npz  pmlib.rotate_and_match called directly (pmlib.py:117-174) - the shape of the reference's own test
                     (tests.py:336-337: whole image 2 as the window, img_size=50, 7 angles), rectangular windows, img_size up to
                     128 - and pmlib.use_mcc with search borders of 112 / 160 / 250 px; the NCC matrices as sha256 + a subsample
  g4_pattern_matching.npz  pmlib.pattern_matching    (pmlib.py:326-497) end to end on an
                     affine stand-in for Nansat, incl. the kernel-input vectors it built
  g5_fullsize.npz    sha256 of the 10000x10000 benchmark pair + C-oracle results on a 1 %
                     subsample of the 200x200 grid (regression pin, not reference output)
  g6_uint8_image.npz lib.get_uint8_image             (lib.py:27-59) on seeded float32 images with NaN / inf
                     pixels, percentile-driven and explicit vmin / vmax
  g7_feature_tracking.npz  ftlib.feature_tracking    (ftlib.py:241-285: domain, Lowe, drift and least-squares
                     filters) on synthetic key points / descriptors, brute-force matcher injected

    python tests/golden/make_golden.py [g1 g2 g3 g4 g5]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import c_oracle, pm_oracle as po, ref_harness          # noqa: E402
from sea_ice_drift_amd import synthetic as syn                     # noqa: E402
from sea_ice_drift_amd.domain import ArrayNansat                   # noqa: E402

G1_ANGLES = [-9, -6, -3, 0, 3, 6, 9, 1.234, -3.85, 45, 90, -7.5]
G1_CENTRES = [(200, 150), (200.3, 150.7), (123.5, 321.5), (50.49999, 60.5)]
G3_ANGLE_SETS = [[-3, 0, 3], list(range(-3, 4)), list(range(-7, 8))]


def g1_image():
    return np.random.default_rng(101).integers(1, 256, (400, 400), dtype=np.uint8)


def g3_pair():
    img1, img2 = syn.make_pair(512, 512, seed=303, amplitude=6.0, period=700.0)
    img1 = img1.copy()
    img1[250:262, 100:140] = 0                      # invalid pixels -> NaN path
    return img1, img2


def g3_points():
    rng = np.random.default_rng(304)
    n = 72
    c1 = rng.uniform(90, 420, n)
    r1 = rng.uniform(90, 420, n)
    c1[::3] = np.rint(c1[::3])                      # a mix of integer and fractional centres
    r1[::3] = np.rint(r1[::3])
    dc, dr = syn.true_displacement(c1, r1, amplitude=6.0, period=700.0)
    c2fg = np.rint(c1 + dc) + rng.integers(-3, 4, n)
    r2fg = np.rint(r1 + dr) + rng.integers(-3, 4, n)
    border = rng.integers(20, 51, n).astype(np.float64)
    border[:6] = [20, 50, 21, 49, 35, 36]
    return c1, r1, c2fg, r2fg, border


def make_g1(pmlib):
    img = g1_image()
    out = []
    for s in (34, 35):
        for a in G1_ANGLES:
            for (c, r) in G1_CENTRES:
                out.append(pmlib.get_template(img, c, r, a, s).ravel())
    edge = pmlib.get_template(img, 5, 5, 10, 34)    # sticks out of the image -> zeros
    np.savez_compressed(os.path.join(HERE, 'g1_templates.npz'),
                        img_sha=syn.sha256(img), t34=np.array(out[:48], dtype=np.uint8),
                        t35=np.array(out[48:], dtype=np.uint8), edge=edge)


def make_g1b(pmlib):
    """get_template(..., rot_order=1): scipy's bilinear sampling (double arithmetic, uint8 output rounding) over G1's
    angle / centre set, plus centres whose template reaches the image border (mode='constant', cval=0)."""
    img = g1_image()
    out = []
    for s in (34, 35):
        for a in G1_ANGLES:
            for (c, r) in G1_CENTRES:
                out.append(pmlib.get_template(img, c, r, a, s, rot_order=1).ravel())
    edges = [(5, 5, 10, 34), (390.5, 17.25, -6, 35), (17, 399, 3, 34), (399, 399, 0, 35), (0, 0, 0, 34)]
    d = dict(img_sha=syn.sha256(img), t34=np.array(out[:48], dtype=np.uint8), t35=np.array(out[48:], dtype=np.uint8),
             edge_args=np.array(edges, dtype=np.float64))
    for k, (c, r, a, s) in enumerate(edges):
        d['edge%d' % k] = pmlib.get_template(img, c, r, a, s, rot_order=1)
    np.savez_compressed(os.path.join(HERE, 'g1b_templates_order1.npz'), **d)


def make_g3b(pmlib):
    """use_mcc(..., rot_order=1) on G3's pair and points (the kwarg travels use_mcc -> rotate_and_match -> get_template,
    pmlib.py:204-208,151), restated matcher injected as in G3."""
    img1, img2 = g3_pair()
    c1, r1, c2fg, r2fg, border = g3_points()
    d = dict(pair_sha=syn.sha256(img1, img2))
    for s, alpha0 in ((34, 0.0), (35, -3.85)):
        for k, angles in enumerate(G3_ANGLE_SETS):
            res = np.array([pmlib.use_mcc(c1[i], r1[i], c2fg[i], r2fg[i], border[i], img1, img2, s, alpha0,
                                          angles=angles, rot_order=1, template_matcher=po.match_template)
                            for i in range(len(c1))], dtype=np.float64)
            d['out_s%d_k%d' % (s, k)] = res
    np.savez_compressed(os.path.join(HERE, 'g3b_use_mcc_order1.npz'), **d)


def g1c_image():
    """A smooth positive image (the spline orders overshoot on noise and hit 0 = the reference's NaN rule)."""
    return syn.make_pair(400, 400, seed=111)[0]


G1C_CASES = [(200, 150, 0, 34), (200.3, 150.7, 7.5, 35), (123.5, 321.5, -9, 34), (50.49999, 60.5, 45, 35), (210, 220, 90, 34),
             (300.25, 100.75, -137.5, 35), (5, 5, 10, 34), (390.5, 17.25, -6, 35), (399, 399, 0, 35), (0, 0, 0, 34), (200, 200, 3, 64)]


def make_g1c(pmlib):
    d = dict(img_sha=syn.sha256(g1c_image()), noise_sha=syn.sha256(g1_image()), cases=np.array(G1C_CASES, dtype=np.float64))
    for name, img in (('smooth', g1c_image()), ('noise', g1_image())):
        for order in (2, 3, 4, 5):
            for k, (c, r, a, s) in enumerate(G1C_CASES):
                d['%s_o%d_%d' % (name, order, k)] = pmlib.get_template(img, c, r, a, s, rot_order=order)
    np.savez_compressed(os.path.join(HERE, 'g1c_templates_spline.npz'), **d)


def make_g3d(pmlib):
    """use_mcc(..., rot_order = 2 / 3 / 5) on G3's pair and points (restated matcher injected as in G3)."""
    img1, img2 = g3_pair()
    c1, r1, c2fg, r2fg, border = g3_points()
    d = dict(pair_sha=syn.sha256(img1, img2))
    for s, alpha0, order, k in G3D_CASES:
        angles = G3_ANGLE_SETS[k]
        res = np.array([pmlib.use_mcc(c1[i], r1[i], c2fg[i], r2fg[i], border[i], img1, img2, s, alpha0,
                                      angles=angles, rot_order=order, template_matcher=c_oracle.match_template)
                        for i in range(len(c1))], dtype=np.float64)
        d['out_s%d_o%d_k%d' % (s, order, k)] = res
        d['alpha0_s%d_o%d_k%d' % (s, order, k)] = alpha0
    np.savez_compressed(os.path.join(HERE, 'g3d_use_mcc_spline.npz'), **d)


G3D_CASES = ((34, 0.0, 3, 0), (35, -3.85, 3, 2), (34, 30.0, 2, 1), (35, 0.0, 5, 0), (34, -3.85, 4, 0))   # s, alpha0, rot_order, angle set
G3C_ALPHA0 = [30.0, 90.0, -137.5]
G3C_ANGLE_SETS = [[-3, 0, 3], list(range(-7, 8))]


def make_g3c(pmlib):
    """use_mcc with LARGE effective rotations: the template is sampled at angle - alpha0 (pmlib.py:151) and alpha0
    (pmlib.py:79-87) is tens of degrees for ascending / descending pairs.  G3's pair and points; alpha0 in G3C_ALPHA0, angle sets
    G3C_ANGLE_SETS, template sides 34 / 35, rot_order 0 and 1; restated matcher injected as in G3.  (The synthetic pair is not
    rotated, so the peaks are weak - the fixture pins arithmetic, not geophysics.)"""
    img1, img2 = g3_pair()
    c1, r1, c2fg, r2fg, border = g3_points()
    d = dict(pair_sha=syn.sha256(img1, img2), alpha0=np.array(G3C_ALPHA0))
    for s in (34, 35):
        for ai, alpha0 in enumerate(G3C_ALPHA0):
            for k, angles in enumerate(G3C_ANGLE_SETS):
                for order in (0, 1):
                    res = np.array([pmlib.use_mcc(c1[i], r1[i], c2fg[i], r2fg[i], border[i], img1, img2, s, alpha0,
                                                  angles=angles, rot_order=order, template_matcher=po.match_template)
                                    for i in range(len(c1))], dtype=np.float64)
                    d['out_s%d_a%d_k%d_o%d' % (s, ai, k, order)] = res
    np.savez_compressed(os.path.join(HERE, 'g3c_large_rotations.npz'), **d)


def g2_inputs():
    rng = np.random.default_rng(202)
    mats = {}
    for n in (42, 72, 101, 102):
        # smooth-ish correlation-like surfaces: low-pass noise in [-1, 1]
        m = rng.standard_normal((n, n)).astype(np.float32)
        k = np.ones(5, dtype=np.float32) / 5
        m = np.apply_along_axis(lambda v: np.convolve(v, k, 'same'), 0, m)
        m = np.apply_along_axis(lambda v: np.convolve(v, k, 'same'), 1, m).astype(np.float32)
        mats[n] = (m / np.abs(m).max()).astype(np.float32)
    return mats


def make_g2(pmlib):
    d = {}
    for n, m in g2_inputs().items():
        d['in%d' % n] = m
        d['norm%d' % n] = pmlib.get_hessian(m)
        d['raw%d' % n] = pmlib.get_hessian(m, hes_norm=False)
        d['smth%d' % n] = pmlib.get_hessian(m, hes_norm=True, hes_smth=True)
    np.savez_compressed(os.path.join(HERE, 'g2_hessian.npz'), **d)


def make_g3(pmlib):
    img1, img2 = g3_pair()
    c1, r1, c2fg, r2fg, border = g3_points()
    d = dict(pair_sha=syn.sha256(img1, img2), c1=c1, r1=r1, c2fg=c2fg, r2fg=r2fg, border=border)
    for s, alpha0 in ((34, 0.0), (35, -3.85)):
        for k, angles in enumerate(G3_ANGLE_SETS):
            for mcc_norm in (False, True):
                if mcc_norm and k != 1:
                    continue
                res = np.array([pmlib.use_mcc(c1[i], r1[i], c2fg[i], r2fg[i], border[i], img1, img2, s, alpha0,
                                              angles=angles, mcc_norm=mcc_norm, template_matcher=po.match_template)
                                for i in range(len(c1))], dtype=np.float64)
                d['out_s%d_k%d_m%d' % (s, k, int(mcc_norm))] = res
    # full intermediates of a few points through the reference's rotate_and_match
    full = []
    for i in (0, 1, 7):
        s, alpha0, angles = 34, 0.0, G3_ANGLE_SETS[1]
        hws = int(s / 2.)
        b = border[i]
        win = img2[int(r2fg[i] - hws - b):int(r2fg[i] + hws + b + 1), int(c2fg[i] - hws - b):int(c2fg[i] + hws + b + 1)]
        dc, dr, ba, br, bh, bres, btmpl = pmlib.rotate_and_match(img1, c1[i], r1[i], s, win, alpha0, angles=angles,
                                                                  template_matcher=po.match_template)
        d['full%d_result' % i] = bres
        d['full%d_template' % i] = btmpl
        d['full%d_scalars' % i] = np.array([dc, dr, ba, br, bh], dtype=np.float64)
        full.append(i)
    d['full_points'] = np.array(full)
    np.savez_compressed(os.path.join(HERE, 'g3_use_mcc.npz'), **d)


def g8_pair():
    img1, img2 = syn.make_pair(640, 700, seed=808, amplitude=5.0, period=500.0)
    img1 = img1.copy()
    img1[600:640, 0:40] = 0                          # a zero corner (NaN x 7 for a template that touches it)
    return img1, img2


# rotate_and_match cases: (name, c1, r1, img_size, window of image 2 as (r0, r1, c0, c1) or None = the whole image, alpha0, kwargs)
G8_RAM = [
    ('tests_py', 300, 100, 50, None, 0, dict(angles=[-3, -2, -1, 0, 1, 2, 3])),            # tests.py:336-337
    ('rect', 320.5, 330.25, 35, (100, 420, 150, 633), -3.85, dict(angles=[-3, 0, 3], mcc_norm=True)),
    ('rect_smth', 200, 250, 34, (60, 331, 40, 500), 30.0, dict(angles=list(range(-7, 8)), hes_smth=True)),
    ('even', 350, 300, 34, (200, 400, 250, 450), 0.0, dict(angles=[0], hes_norm=False)),       # a 200 x 200 window (even side)
    ('s64', 330, 320, 64, (150, 500, 150, 520), 2.0, dict(angles=[-2, 0, 2], rot_order=1)),
    ('s100', 330, 320, 100, (100, 560, 120, 600), 0.0, dict(angles=[-3, 0, 3])),
    ('s128', 330, 320, 128, None, 0.0, dict(angles=[0, 4])),
    ('s65', 310.5, 305.5, 65, (100, 400, 100, 431), -90.0, dict(angles=[-1, 0, 1, 2], mcc_norm=True, hes_smth=True)),
    ('nan', 30, 615, 50, (100, 300, 100, 300), 0.0, dict(angles=[-3, 0, 3])),                # the template touches the zero corner
    ('tiny', 330, 320, 50, (100, 152, 100, 151), 0.0, dict(angles=[0, 1])),                  # a 3 x 2 NCC matrix
]
# use_mcc cases beyond the LDS classes: (name, c1, r1, c2fg, r2fg, border, img_size, alpha0, kwargs)
G8_MCC = [
    ('b112', 330.0, 320.0, 333.0, 318.0, 112.0, 34, 0.0, dict(angles=[-3, 0, 3])),
    ('b160', 340.0, 310.0, 338.0, 312.0, 160.0, 35, -3.85, dict(angles=list(range(-3, 4)))),
    ('b250', 350.0, 320.0, 350.0, 320.0, 250.0, 34, 0.0, dict(angles=[-3, 0, 3], mcc_norm=True)),
    ('b30_s100', 330.0, 320.0, 331.0, 322.0, 30.0, 100, 1.5, dict(angles=[-3, 0, 3])),
    ('b20_s65', 330.5, 320.5, 331.0, 322.0, 20.0, 65, 0.0, dict(angles=list(range(-7, 8)))),
]


def ccm_digest(m):
    """sha256 of the float32 matrix bytes + a strided subsample (what a fixture keeps of a large NCC matrix)."""
    import hashlib
    m = np.ascontiguousarray(m, dtype=np.float32)
    return hashlib.sha256(m.tobytes()).hexdigest(), m[::max(1, m.shape[0] // 16), ::max(1, m.shape[1] // 16)].copy()


def make_g8(pmlib):
    """rotate_and_match as a call of its own and use_mcc beyond the LDS launch classes, from the imported reference with the
    restated matcher (its C form - pinned to the NumPy form by tests/test_oracle_golden.py - for speed) injected."""
    img1, img2 = g8_pair()
    d = dict(pair_sha=syn.sha256(img1, img2))
    for name, c1, r1, s, win, alpha0, kw in G8_RAM:
        image2 = img2 if win is None else img2[win[0]:win[1], win[2]:win[3]]
        out = pmlib.rotate_and_match(img1, c1, r1, s, image2, alpha0, template_matcher=c_oracle.match_template, **kw)
        if isinstance(out[5], float):                               # NaN x 7 (pmlib.py:152-154)
            assert all(np.isnan(x) for x in out)
            d['ram_%s_scalars' % name] = np.full(5, np.nan)
            continue
        d['ram_%s_scalars' % name] = np.array(out[:5], dtype=np.float64)
        sha, sub = ccm_digest(out[5])
        d['ram_%s_ccm_sha' % name] = sha
        d['ram_%s_ccm_sub' % name] = sub
        d['ram_%s_ccm_shape' % name] = np.array(out[5].shape)
        d['ram_%s_template' % name] = out[6]
    for name, c1, r1, c2fg, r2fg, b, s, alpha0, kw in G8_MCC:
        d['mcc_%s' % name] = np.array(pmlib.use_mcc(c1, r1, c2fg, r2fg, b, img1, img2, s, alpha0,
                                                    template_matcher=c_oracle.match_template, **kw), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'g8_rotate_and_match.npz'), **d)


def g4_inputs():
    img1, img2 = syn.make_pair(1200, 1200, seed=5)
    n1 = ArrayNansat.rotated(img1, angle_deg=0.0, scale=0.001, origin=(10.0, 70.0))
    n2 = ArrayNansat.rotated(img2, angle_deg=2.0, scale=0.001, origin=(10.003, 70.002))
    rng = np.random.default_rng(3)
    c1 = rng.uniform(100, 1100, 300)
    r1 = rng.uniform(100, 1100, 300)
    lon, lat = n1.transform_points(c1, r1)
    c2, r2 = n2.transform_points(lon, lat, 1)
    dc, dr = syn.true_displacement(c1, r1)
    c2 = c2 + dc + rng.normal(0, 0.5, 300)
    r2 = r2 + dr + rng.normal(0, 0.5, 300)
    lon_g, lat_g = np.meshgrid(np.linspace(10.05, 11.1, 25), np.linspace(70.05, 71.1, 25))
    return n1, n2, c1, r1, c2, r2, lon_g, lat_g


def make_g4(pmlib):
    n1, n2, c1, r1, c2, r2, lon_g, lat_g = g4_inputs()
    rec = {}
    orig = pmlib.prepare_first_guess

    def wrap(*a, **k):
        out = orig(*a, **k)
        rec['fg'] = [x.copy() for x in out]
        return out
    pmlib.prepare_first_guess = wrap
    try:
        kw = dict(img_size=34, angles=list(range(-3, 4)))
        out = pmlib.pattern_matching(lon_g, lat_g, n1, c1, r1, n2, c2, r2, threads=1,
                                     template_matcher=po.match_template, **kw)
    finally:
        pmlib.prepare_first_guess = orig
    sa = pmlib.shared_args
    np.savez_compressed(os.path.join(HERE, 'g4_pattern_matching.npz'),
                        pair_sha=syn.sha256(n1[1], n2[1]), c1=c1, r1=r1, c2=c2, r2=r2, lon_g=lon_g, lat_g=lat_g,
                        fg_c2=rec['fg'][0], fg_r2=rec['fg'][1], fg_border=rec['fg'][2],
                        k_c1=sa[0], k_r1=sa[1], k_c2fg=sa[2], k_r2fg=sa[3], k_border=sa[4], alpha0=sa[8],
                        u=out[0], v=out[1], a=out[2], r=out[3], h=out[4], lon2=out[5], lat2=out[6])


def make_g5():
    img1, img2 = syn.make_pair(10000, 10000)
    g = syn.make_grid(10000, 10000, 200)
    sel = np.arange(37, 40000, 100)
    angles = list(range(-7, 8))
    rot = np.array([po.rotation_terms(a, 34) for a in angles])
    out, ij = c_oracle.pm_batch(img1, img2, g['c1'][sel], g['r1'][sel], g['c2fg'][sel], g['r2fg'][sel],
                                g['border'][sel], 34, 0.0, angles, rot=rot, nthreads=os.cpu_count())
    np.savez_compressed(os.path.join(HERE, 'g5_fullsize.npz'), pair_sha=syn.sha256(img1, img2),
                        grid_sha=syn.sha256(*[g[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]),
                        sel=sel, out=out, ij=ij)


def g6_cases():
    """Inputs of the staging fixture: float32 'sigma0 in dB'-like images with NaN / inf pixels."""
    rng = np.random.default_rng(606)
    cases = []
    for k, shape in enumerate([(64, 80), (301, 257), (1, 37), (240, 260)]):
        img = rng.normal(-22.0, 4.0, shape).astype(np.float32)
        img[rng.random(shape) < 0.07] = np.nan
        if k == 1:
            img[5, 7] = np.inf; img[9, 11] = -np.inf
        if k == 3:
            img[100:140, 120:180] = np.nan                  # a masked block (land)
            img[::17, ::13] = np.float32(-22.0)              # many exact ties
        cases.append(img)
    return cases


G6_PARAMS = [(None, None, 10, 99), (None, None, 1, 99.9), (-30.0, -12.5, 10, 99), (None, -10.0, 5, 99), (-35.0, None, 10, 95)]


def make_g6(reflib):
    """lib.get_uint8_image (lib.py:27-59) run as the reference wrote it (numpy %s)."""
    import contextlib, io
    d = {'numpy_version': np.__version__}
    for ci, img in enumerate(g6_cases()):
        for pi, (vmin, vmax, pmin, pmax) in enumerate(G6_PARAMS):
            if ci == 1 and (vmin is None or vmax is None) and pi == 1:
                pass
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all='ignore'):
                out = reflib.get_uint8_image(img.copy(), vmin, vmax, pmin, pmax)
            d['out_%d_%d' % (ci, pi)] = out
            with np.errstate(all='ignore'):
                d['vmin_%d_%d' % (ci, pi)] = np.float64(np.nanpercentile(img, pmin) if vmin is None else vmin)
                d['vmax_%d_%d' % (ci, pi)] = np.float64(np.nanpercentile(img, pmax) if vmax is None else vmax)
    np.savez_compressed(os.path.join(HERE, 'g6_uint8_image.npz'), **d)


class G7KeyPoint(object):
    """What the reference reads of a cv2.KeyPoint."""
    def __init__(self, x, y):
        self.pt = (float(x), float(y))


class G7Match(object):
    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(d)


class G7BruteForce(object):
    """Stands in for cv2.BFMatcher(cv2.NORM_HAMMING) through the reference's own ``matcher=`` argument
    (ftlib.py:66,95): brute-force Hamming k-nearest neighbours, equal distances by the smaller index."""
    def __init__(self, norm):
        self.norm = norm

    def knnMatch(self, d1, d2, k=2):
        from oracle import ft_oracle
        idx, dist = ft_oracle.knn2(d1, d2)
        return [tuple(G7Match(q, idx[q, j], dist[q, j]) for j in range(k)) for q in range(len(d1))]


class G7Nansat(ArrayNansat):
    """ArrayNansat with nansat's behaviour for the time stamp: ValueError when the data has none."""
    def __init__(self, image, start=None, **kw):
        ArrayNansat.__init__(self, image, **kw)
        self._start = start

    @property
    def time_coverage_start(self):
        if self._start is None:
            raise ValueError('no time_coverage_start')
        return self._start


def g7_inputs(timed):
    """Two 3000x3000 'images' (only their shape and georeference matter), 6000 / 5500 synthetic key points
    with 256-bit descriptors: 3500 true correspondences (few flipped bits, ~1 % gross outliers), the rest noise."""
    import datetime
    rng = np.random.default_rng(707)
    shape = (3000, 3000)
    img = np.ones(shape, dtype=np.uint8)
    t0 = datetime.datetime(2020, 1, 1)
    scale = 4e-4                                           # degrees per pixel (~44 m)
    n1 = G7Nansat(img, start=t0 if timed else None, origin=(10.0, 78.0), matrix=((scale, 0.0), (0.0, -scale)))
    n2 = G7Nansat(img, start=t0 + datetime.timedelta(hours=24) if timed else None,
                  origin=(10.0 + 150 * scale, 78.0 - 90 * scale), matrix=((scale, 0.0), (0.0, -scale)))
    ncommon, na, nb = 3500, 2500, 2000
    xy1 = rng.uniform(40, 2960, (ncommon + na, 2))
    d1 = rng.integers(0, 256, (ncommon + na, 32), dtype=np.uint8)
    # image 2 sees the common points shifted by the georeference offset plus a smooth drift, with noisy descriptors
    drift = np.stack([8 + 6 * np.sin(xy1[:ncommon, 1] / 700.0), -5 + 4 * np.cos(xy1[:ncommon, 0] / 900.0)], axis=1)
    xy2c = xy1[:ncommon] - np.array([150.0, 90.0]) + drift + rng.normal(0, 0.4, (ncommon, 2))
    flips = np.zeros((ncommon, 256), dtype=np.uint8)
    for r in range(ncommon):
        flips[r, rng.permutation(256)[:rng.integers(0, 45)]] = 1
    d2c = d1[:ncommon] ^ np.packbits(flips, axis=1)
    bad = rng.permutation(ncommon)[:35]                    # gross outliers: right descriptor, wrong place
    xy2c[bad] += rng.uniform(-900, 900, (len(bad), 2))
    xy2 = np.concatenate([xy2c, rng.uniform(40, 2960, (nb, 2))])
    d2 = np.concatenate([d2c, rng.integers(0, 256, (nb, 32), dtype=np.uint8)])
    p2 = rng.permutation(len(xy2))
    return n1, n2, xy1, d1, xy2[p2], d2[p2]


def make_g7():
    """ftlib.feature_tracking (ftlib.py:241-285) as the reference wrote it, fed with synthetic key points
    (its find_key_points is OpenCV's ORB) and a brute-force matcher through its own ``matcher=`` argument."""
    ftlib = ref_harness.load_ftlib()
    d = {}
    for timed in (False, True):
        n1, n2, xy1, d1, xy2, d2 = g7_inputs(timed)
        table = {id(n1[1]): (xy1, d1), id(n2[1]): (xy2, d2)}
        calls = []

        def fake_find_key_points(image, **kwargs):
            xy, desc = (xy1, d1) if not calls else (xy2, d2)
            calls.append(1)
            return [G7KeyPoint(x, y) for x, y in xy], desc
        ftlib.find_key_points = fake_find_key_points
        kw = dict(matcher=G7BruteForce, max_drift=25000.0) if not timed else dict(matcher=G7BruteForce, max_speed=0.3)
        x1, y1, x2, y2 = ftlib.feature_tracking(n1, n2, domainMargin=10, ratio_test=0.75, psi=150, **kw)
        tag = 'timed' if timed else 'untimed'
        for name, v in (('x1', x1), ('y1', y1), ('x2', x2), ('y2', y2)):
            d['%s_%s' % (name, tag)] = np.asarray(v, dtype=np.float64)
        assert len(x1) > 2000, len(x1)
    np.savez_compressed(os.path.join(HERE, 'g7_feature_tracking.npz'), **d)


def main():
    which = sys.argv[1:] or ['g1', 'g1b', 'g1c', 'g2', 'g3', 'g3b', 'g3c', 'g3d', 'g4', 'g5', 'g6', 'g7', 'g8']
    if 'g7' in which:
        t = time.time()
        make_g7()
        print('g7 done in %.1f s' % (time.time() - t))
    pmlib, reflib = ref_harness.load()
    if 'g6' in which:
        t = time.time()
        make_g6(reflib)
        print('g6 done in %.1f s' % (time.time() - t))
    c_oracle.build()
    for name, fn in (('g1', make_g1), ('g1b', make_g1b), ('g1c', make_g1c), ('g3d', make_g3d), ('g2', make_g2), ('g3', make_g3), ('g3b', make_g3b), ('g3c', make_g3c), ('g4', make_g4), ('g8', make_g8)):
        if name in which:
            t = time.time()
            fn(pmlib)
            print(name, 'done in %.1f s' % (time.time() - t))
    if 'g5' in which:
        t = time.time()
        make_g5()
        print('g5 done in %.1f s' % (time.time() - t))


if __name__ == '__main__':
    main()
