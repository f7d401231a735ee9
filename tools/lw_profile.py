import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from sea_ice_drift_amd import _capi, pmlib as my, synthetic as syn
size, s, K = 4000, 50, 7
img1, img2 = syn.make_pair(size, size, seed=11)
angles = list(np.linspace(-3, 3, K)); rot = my.rotation_table(angles, 0.0, s)
with _capi.PMContext(0) as ctx:
    ctx.upload_pair(img1, img2)
    for _ in range(3):
        ctx.rotate_and_match(size * 0.6, size * 0.2, s, 0.0, angles, rot=rot, window=(0, 0, size, size), want_ccm=False, want_template=False, flags=7)
