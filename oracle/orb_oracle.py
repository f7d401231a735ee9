"""NumPy restatement of the key-point detector's specification (include/sid_orb.h).

TEST INFRASTRUCTURE ONLY (same rule as the rest of ``oracle/``).  Parity status: **unpinned against the
reference** - the reference's ``find_key_points`` (ftlib.py:26-61) is OpenCV's ORB, cv2 is not installed, and its
tests assert only ``> 1000`` key points (tests.py:230).  This file pins the package's own detector: the HIP kernels
(``csrc/orb.hip``) must reproduce it bit for bit (all steps are integer arithmetic).  The direction table and the
comparison pattern are data shared with the product (``sea_ice_drift_amd.orb``), not code.
"""
import numpy as np

RING = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
        (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def level_geometry(rows, cols, n_levels, scale_factor, n_features):
    scale = float(np.float32(scale_factor))
    sc, lr, lc = [], [], []
    s = 1.0
    for _ in range(n_levels):
        sc.append(s)
        lr.append(int(np.floor(rows / s + 0.5)))
        lc.append(int(np.floor(cols / s + 0.5)))
        s *= scale
    factor = 1.0 / scale
    nd = n_features * (1.0 - factor) / (1.0 - factor ** n_levels)
    want, tot = [], 0
    for _ in range(n_levels - 1):
        want.append(int(np.floor(nd + 0.5)))
        tot += want[-1]
        nd *= factor
    want.append(max(n_features - tot, 0))
    return sc, lr, lc, want


def resize(img0, r, c):
    rows0, cols0 = img0.shape
    sx = (cols0 << 16) // c
    sy = (rows0 << 16) // r
    fx = np.clip(np.arange(c, dtype=np.int64) * sx + (sx >> 1) - 32768, 0, (cols0 - 1) << 16)
    fy = np.clip(np.arange(r, dtype=np.int64) * sy + (sy >> 1) - 32768, 0, (rows0 - 1) << 16)
    x0, y0 = fx >> 16, fy >> 16
    wx, wy = (fx >> 8) & 255, (fy >> 8) & 255
    x1, y1 = np.minimum(x0 + 1, cols0 - 1), np.minimum(y0 + 1, rows0 - 1)
    I = img0.astype(np.int64)
    a, b = I[y0][:, x0], I[y0][:, x1]
    cc, d = I[y1][:, x0], I[y1][:, x1]
    top = a * (256 - wx)[None, :] + b * wx[None, :]
    bot = cc * (256 - wx)[None, :] + d * wx[None, :]
    return ((top * (256 - wy)[:, None] + bot * wy[:, None] + 32768) >> 16).astype(np.uint8)


def fast_score(img, edge, t):
    r, c = img.shape
    I = img.astype(np.int32)
    score = np.zeros((r, c), dtype=np.int32)
    ys, xs = slice(edge, r - edge), slice(edge, c - edge)
    centre = I[ys, xs]
    d = [I[edge + dy:r - edge + dy, edge + dx:c - edge + dx] - centre for dx, dy in RING]
    best = np.full(centre.shape, -256, dtype=np.int32)
    for s in range(16):
        arc = np.stack([d[(s + k) & 15] for k in range(9)])
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    score[ys, xs] = np.where(best > t, best, 0)
    return score


def candidates(img, score, edge):
    r, c = img.shape
    s = score
    core = s[edge:r - edge, edge:c - edge]
    keep = core > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                keep &= core > s[edge + dy:r - edge + dy, edge + dx:c - edge + dx]
    ys, xs = np.nonzero(keep)
    ys, xs = ys + edge, xs + edge
    I = img.astype(np.int64)
    resp = np.empty(len(xs), dtype=np.int64)
    for k, (x, y) in enumerate(zip(xs, ys)):
        blk = I[y - 4:y + 5, x - 4:x + 5]
        ix = blk[1:8, 2:9] - blk[1:8, 0:7]
        iy = blk[2:9, 1:8] - blk[0:7, 1:8]
        a, b, cc = (ix * ix).sum(), (iy * iy).sum(), (ix * iy).sum()
        resp[k] = 25 * (a * b - cc * cc) - (a + b) * (a + b)
    return xs, ys, resp


def blur(img):
    I = np.pad(img.astype(np.int64), 2, mode='edge')
    w = (1, 4, 6, 4, 1)
    r, c = img.shape
    h = sum(w[k] * I[:, k:k + c] for k in range(5))
    v = sum(w[k] * h[k:k + r, :] for k in range(5))
    return ((v + 128) >> 8).astype(np.uint8)


def detect_and_compute(image, pattern, dirs, edge_threshold=34, n_features=100000, n_levels=7, patch_size=34,
                       fast_threshold=20, scale_factor=1.2):
    """-> xy float32 [N,2], meta int32 [N,4], response int64 [N], desc uint8 [N,32]; ordered like the kernels'."""
    img0 = np.asarray(image, dtype=np.uint8)
    rows, cols = img0.shape
    sc, lr, lc, want = level_geometry(rows, cols, n_levels, scale_factor, n_features)
    R = patch_size // 2
    dyy, dxx = np.mgrid[-R:R + 1, -R:R + 1]
    disc = (dxx * dxx + dyy * dyy) <= R * R
    out_xy, out_meta, out_resp, out_desc = [], [], [], []
    total = 0
    for l in range(n_levels):
        r, c = lr[l], lc[l]
        if r <= 2 * edge_threshold or c <= 2 * edge_threshold or want[l] <= 0 or total >= n_features:
            continue
        lvl = img0 if l == 0 else resize(img0, r, c)
        xs, ys, resp = candidates(lvl, fast_score(lvl, edge_threshold, fast_threshold), edge_threshold)
        if len(xs) == 0:
            continue
        order = np.lexsort((xs, ys, -resp))
        n = min(len(xs), want[l], n_features - total)
        order = order[:n]
        xs, ys, resp = xs[order], ys[order], resp[order]
        bl = blur(lvl).astype(np.int32)
        I = lvl.astype(np.int64)
        for x, y, rs in zip(xs, ys, resp):
            patch = I[y - R:y + R + 1, x - R:x + R + 1]
            m10 = int((dxx * patch)[disc].sum())
            m01 = int((dyy * patch)[disc].sum())
            v = m10 * dirs[:, 0].astype(np.int64) + m01 * dirs[:, 1].astype(np.int64)
            b = int(np.argmax(v))                                  # first maximum = smallest direction index
            pt = pattern[b].astype(np.int64)
            bits = bl[y + pt[:, 1], x + pt[:, 0]] < bl[y + pt[:, 3], x + pt[:, 2]]
            out_desc.append(np.packbits(bits.astype(np.uint8), bitorder='little'))
            out_xy.append((np.float32(float(x) * sc[l]), np.float32(float(y) * sc[l])))
            out_meta.append((x, y, l, b))
            out_resp.append(rs)
        total += n
    if not out_xy:
        return (np.zeros((0, 2), np.float32), np.zeros((0, 4), np.int32), np.zeros(0, np.int64), np.zeros((0, 32), np.uint8))
    return (np.array(out_xy, dtype=np.float32), np.array(out_meta, dtype=np.int32), np.array(out_resp, dtype=np.int64),
            np.array(out_desc, dtype=np.uint8))
