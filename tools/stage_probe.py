#!/usr/bin/env python3
"""Staging passes timed one by one (wall clock, warm): begin (first pass), order_stats with 1 / 2 / 4 ranks, the map."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sea_ice_drift_amd import _capi
n = 10000
rng = np.random.default_rng(5)
img = rng.normal(-22.0, 4.0, (n, n)).astype(np.float32)
img[rng.random((n, n), dtype=np.float32) < 0.05] = np.nan
t = torch.from_numpy(img).cuda()
ws = _capi.StageWorkspace(0)
st = torch.cuda.current_stream().cuda_stream
def best(fn, reps=7):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return round(b * 1e3, 4)
nv = ws.begin(t.data_ptr(), n, n, n, st)
res = {'begin_ms': best(lambda: ws.begin(t.data_ptr(), n, n, n, st))}
for ranks in ([nv // 2], [nv // 10, nv // 10 + 1], [nv // 10, nv // 10 + 1, nv * 99 // 100, nv * 99 // 100 + 1], [nv // 10, nv // 5, nv // 2, nv * 99 // 100]):
    res['order_stats_%d_ranks_%s_ms' % (len(ranks), 'spread' if len(ranks) == 4 and ranks[1] - ranks[0] > 1 else 'adjacent')] = best(lambda: ws.order_stats(ranks))
fr = [0.10, 0.99]
rk4 = [int(0.10 * (nv - 1)), int(0.10 * (nv - 1)) + 1, int(0.99 * (nv - 1)), int(0.99 * (nv - 1)) + 1]
res['begin_hint_ms'] = best(lambda: ws.begin(t.data_ptr(), n, n, n, st, fractions=fr))
res['order_stats_4_ranks_after_hint_ms'] = best(lambda: ws.order_stats(rk4))
def both():
    ws.begin(t.data_ptr(), n, n, n, st, fractions=fr); ws.order_stats(rk4)
res['begin_hint_plus_order_stats_ms'] = best(both)
def both_plain():
    ws.begin(t.data_ptr(), n, n, n, st); ws.order_stats(rk4)
res['begin_plus_order_stats_plain_ms'] = best(both_plain)
import io, contextlib
from sea_ice_drift_amd import lib
def whole():
    with contextlib.redirect_stdout(io.StringIO()):
        lib.get_uint8_image(t, None, None, 10, 99)
res['get_uint8_image_ms'] = best(whole)
os.environ['SID_STAGE_NO_HINT'] = '1'
res['get_uint8_image_no_hint_ms'] = best(whole)
del os.environ['SID_STAGE_NO_HINT']
out = torch.empty((n, n), dtype=torch.uint8, device='cuda')
res['scale_ms'] = best(lambda: _capi.stage_scale_u8(t.data_ptr(), n, n, n, np.float32(-30.0), np.float32(20.0), out.data_ptr(), n, st))
print(json.dumps(res))
