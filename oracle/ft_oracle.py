"""ORACLE (test infrastructure only - never imported by the product): NumPy restatement of the reference's
descriptor matcher, ``/root/reference/sea_ice_drift/ftlib.py:92-116``.

``bf.knnMatch(descriptors1, descriptors2, k=2)`` with ``cv2.NORM_HAMMING`` (ftlib.py:95-96) is a brute-force
scan: distance = number of differing bits of the two 32-byte strings; the two smallest per query, nearest
first.  OpenCV (un-vendored, unpinned: README.md:32) is absent from this build, so the tie order among equal
distances is NOT pinned by the reference ("parity unpinned" for ties): this restatement - like the kernel -
keeps the smaller train index first.  The Lowe filter is ftlib.py:101-116 verbatim in array form.
"""
import numpy as np


def knn2(desc1, desc2, chunk=256):
    d1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 32)
    n1, n2 = len(d1), len(d2)
    idx = np.full((n1, 2), -1, dtype=np.int32)
    dist = np.full((n1, 2), -1, dtype=np.int32)
    if n2 == 0:
        return idx, dist
    w2 = d2.view(np.uint64).reshape(n2, 4)
    for a in range(0, n1, chunk):
        w1 = d1[a:a + chunk].view(np.uint64).reshape(-1, 4)
        dm = np.bitwise_count(w1[:, None, :] ^ w2[None, :, :]).sum(axis=2).astype(np.int64)      # [q, n2]
        key = dm * (1 << 32) + np.arange(n2, dtype=np.int64)[None, :]                             # distance, then index
        kk = min(2, n2)
        part = np.sort(np.partition(key, kk - 1, axis=1)[:, :kk], axis=1)
        idx[a:a + chunk, :kk] = (part & 0xffffffff).astype(np.int32)
        dist[a:a + chunk, :kk] = (part >> 32).astype(np.int32)
    return idx, dist


def knn2_loops(desc1, desc2):
    """Pure-Python cross-check for tiny inputs."""
    out_i, out_d = [], []
    for q in np.asarray(desc1, dtype=np.uint8).reshape(-1, 32):
        cand = sorted((int(np.unpackbits(q ^ t).sum()), j) for j, t in enumerate(np.asarray(desc2, dtype=np.uint8).reshape(-1, 32)))
        cand = (cand + [(-1, -1), (-1, -1)])[:2]
        out_d.append([c[0] for c in cand]); out_i.append([c[1] for c in cand])
    return np.array(out_i, dtype=np.int32).reshape(-1, 2), np.array(out_d, dtype=np.int32).reshape(-1, 2)


def filter_matches(idx, dist, ratio_test, pts1, pts2):
    """ftlib.py:101-116."""
    good = [q for q in range(len(idx)) if float(dist[q, 0]) < ratio_test * float(dist[q, 1])]
    x1 = np.array([pts1[q][0] for q in good]); y1 = np.array([pts1[q][1] for q in good])
    x2 = np.array([pts2[idx[q, 0]][0] for q in good]); y2 = np.array([pts2[idx[q, 0]][1] for q in good])
    return x1, y1, x2, y2
