"""Seeded synthetic SAR-like uint8 image pairs and PM grids.

There is no network and no Sentinel-1 data on the build or GPU machines, so the
benchmark and the parity tests run on a generated pair:

* img1  : band-limited texture (three octaves of box-blurred Gaussian noise),
          mapped to uint8 in [1, 255] (0 is the reference's "invalid" value,
          lib.py:52-57, and would trigger the NaN path of pmlib.py:152-154).
* img2  : img1 resampled (nearest) under the analytic displacement field
          d_col = A sin(2 pi row / L), d_row = -A sin(2 pi col / L), then
          multiplicative speckle, clipped to [1, 255].
* grid  : n x n integer pixel positions, first guess = true displacement + a small
          seeded error, search border per point (fixed or Rayleigh-"mixed",
          mimicking pmlib.py:300-318).

Everything is NumPy/SciPy on the host and deterministic for a given seed.
"""
import hashlib

import numpy as np
from scipy import ndimage as nd

SEED_IMG = 20200123
SEED_SPECKLE = 20200125
SEED_GRID = 20200127
FIELD_AMPLITUDE = 10.0
FIELD_PERIOD = 1200.0


def _texture(rows, cols, rng, strip=2048):
    """Sum of box-blurred white noise at widths 3/9/27, float32, built in row strips."""
    out = np.empty((rows, cols), dtype=np.float32)
    halo = 16
    # noise is drawn for the whole image row-block by row-block so that the result does
    # not depend on the strip size: one generator call per row.
    noise = np.empty((rows + 2 * halo, cols + 2 * halo), dtype=np.float32)
    for r0 in range(0, noise.shape[0], strip):
        r1 = min(r0 + strip, noise.shape[0])
        noise[r0:r1] = rng.standard_normal((r1 - r0, noise.shape[1]), dtype=np.float32)
    acc = np.zeros_like(noise)
    for width, gain in ((3, 1.0), (9, 2.0), (27, 4.0)):
        acc += gain * nd.uniform_filter(noise, size=width, mode='nearest')
    out[:] = acc[halo:halo + rows, halo:halo + cols]
    return out


def make_pair(rows, cols, seed=SEED_IMG, amplitude=FIELD_AMPLITUDE, period=FIELD_PERIOD,
              speckle=0.08, strip=1024):
    """Return (img1, img2) uint8 arrays in [1, 255]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tex = _texture(rows, cols, rng)
    m, s = float(tex.mean(dtype=np.float64)), float(tex.std(dtype=np.float64))
    img1 = np.clip(np.rint(128.0 + 45.0 * (tex - m) / s), 1, 255).astype(np.uint8)
    del tex
    rng2 = np.random.Generator(np.random.PCG64(seed + (SEED_SPECKLE - SEED_IMG)))
    img2 = np.empty_like(img1)
    cc = np.arange(cols, dtype=np.float64)[None, :]
    d_row = -amplitude * np.sin(2 * np.pi * cc / period)            # depends on col only
    for r0 in range(0, rows, strip):
        r1 = min(r0 + strip, rows)
        rr = np.arange(r0, r1, dtype=np.float64)[:, None]
        d_col = amplitude * np.sin(2 * np.pi * rr / period)         # depends on row only
        # a feature at (r, c) of img1 appears at (r + d_row, c + d_col) of img2
        src_r = np.clip(np.rint(rr - d_row), 0, rows - 1).astype(np.int64)
        src_c = np.clip(np.rint(cc - d_col), 0, cols - 1).astype(np.int64)
        block = img1[src_r, src_c].astype(np.float32)
        block *= 1.0 + speckle * rng2.standard_normal(block.shape, dtype=np.float32)
        img2[r0:r1] = np.clip(np.rint(block), 1, 255).astype(np.uint8)
    return img1, img2


def true_displacement(c, r, amplitude=FIELD_AMPLITUDE, period=FIELD_PERIOD):
    """(d_col, d_row) of the synthetic field at pixel (c, r) of image 1."""
    return (amplitude * np.sin(2 * np.pi * np.asarray(r, dtype=np.float64) / period),
            -amplitude * np.sin(2 * np.pi * np.asarray(c, dtype=np.float64) / period))


def make_grid(rows, cols, n_side, border='mixed', seed=SEED_GRID, margin=100, fg_error=3):
    """Kernel inputs for an n_side x n_side (or (n_rows, n_cols)) grid on a shared
    georeference (alpha0 = 0).

    Returns dict(c1, r1, c2fg, r2fg, border) of float64 vectors holding integers, exactly
    the five vectors the reference hands to its Pool (pmlib.py:438,443).
    border: 'mixed' -> clip(floor(Rayleigh(16)), 20, 50); or an int for a fixed border.
    """
    n_rows, n_cols = (n_side, n_side) if np.isscalar(n_side) else n_side
    rng = np.random.Generator(np.random.PCG64(seed))
    cs = np.rint(np.linspace(margin, cols - 1 - margin, n_cols))
    rs = np.rint(np.linspace(margin, rows - 1 - margin, n_rows))
    c1, r1 = np.meshgrid(cs, rs)
    c1, r1 = c1.ravel(), r1.ravel()
    dc, dr = true_displacement(c1, r1)
    err = rng.integers(-fg_error, fg_error + 1, size=(2, c1.size))
    c2fg = c1 + np.rint(dc) + err[0]
    r2fg = r1 + np.rint(dr) + err[1]
    if border == 'mixed':
        b = np.clip(np.floor(rng.rayleigh(16.0, size=c1.size)), 20, 50)
    else:
        b = np.full(c1.size, float(border))
    return dict(c1=c1.astype(np.float64), r1=r1.astype(np.float64), c2fg=c2fg.astype(np.float64),
                r2fg=r2fg.astype(np.float64), border=b.astype(np.float64))


def sha256(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
