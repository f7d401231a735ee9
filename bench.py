#!/usr/bin/env python3
"""bench.py - PM grid-points/sec of the HIP pattern-matching path on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N=1 directly; N>1 under
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): one synthetic 10000x10000 uint8 pair, 200x200 grid,
34 px template, 15 trial angles [-7..7], mixed search border 20..50 px - the five kernel
input vectors and both images are resident in HBM before the timed region starts.

A "step" = one pass of the hot path over the whole grid: kernel launches for every point,
the gather of the (N,5)+(N,3) results to rank 0 (RCCL when N>1) and their copy to the host
(the reference's seam ends with the results in the parent process, pmlib.py:444,462).
N>1 is weak scaling: every rank owns a 200x200-point share of a (200*N)x200 grid on the
same pair (points dealt by search-window size, sea_ice_drift_amd/dist.py).

Prints ONE JSON line on rank 0 (see the task contract), with a "roofline" object measured
live from HIP events around the kernel launches and a "cpu_baseline" object = the C oracle
(oracle/pm_oracle.c, OpenMP over points) timed on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks used for the roofline fractions (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
# dense int8 MFMA = 2x the bf16 dense peak (the guide: 'I8 ~2x bf16 rate'; 4 855 TOP/s measured here
# with tools/ubench/valu_rates.hip); integer ops, quoted in the contract's 'TFLOP/s' unit field
MFMA_I8_PEAK_TOPS = 5000.0


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=10000, help='image side in pixels')
    ap.add_argument('--grid', type=int, default=200, help='grid points per side (per GPU)')
    ap.add_argument('--border', default='mixed', help="'mixed' or a fixed border in pixels")
    ap.add_argument('--angles', type=int, default=7, help='trial angles are -A..A in 1 degree steps')
    ap.add_argument('--img-size', type=int, default=34)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=0, help='points in the CPU sample (0 = auto, ~15 s)')
    ap.add_argument('--check', type=int, default=64, help='points verified against the oracle after timing')
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from sea_ice_drift_amd import _capi, synthetic as syn
    from sea_ice_drift_amd.dist import ResultGatherer, shard_indices, shard_size

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`'
                             % (args.gpus, args.gpus))
        raise SystemExit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    border = args.border if args.border == 'mixed' else int(args.border)
    angles = list(range(-args.angles, args.angles + 1))
    s = args.img_size
    H = W = args.size

    # ---- inputs: rank 0 generates the pair, the others receive it over xGMI ----
    t_gen = time.time()
    if rank == 0:
        img1, img2 = syn.make_pair(H, W)
        t1 = torch.from_numpy(img1).to(dev)
        t2 = torch.from_numpy(img2).to(dev)
    else:
        img1 = img2 = None
        t1 = torch.empty((H, W), dtype=torch.uint8, device=dev)
        t2 = torch.empty((H, W), dtype=torch.uint8, device=dev)
    if world > 1:
        dist.broadcast(t1, 0)
        dist.broadcast(t2, 0)
    t_gen = time.time() - t_gen

    n_rows = args.grid * world if args.scaling == 'weak' else args.grid
    g = syn.make_grid(H, W, (n_rows, args.grid), border=border)
    n_total = g['c1'].size
    idx = shard_indices(g['border'], world, rank)
    m = shard_size(n_total, world)
    from sea_ice_drift_amd.pmlib import rotation_table     # rotation terms via NumPy, as the reference
    rot = rotation_table(angles, 0.0, s)

    ctx = _capi.PMContext(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.bind_pair_tensors(t1, t2)
    ctx.set_points(g['c1'][idx], g['r1'][idx], g['c2fg'][idx], g['r2fg'][idx], g['border'][idx], s, 0.0, angles,
                   rot=rot)
    out_t = torch.full((m, 5), float('nan'), dtype=torch.float64, device=dev)
    ij_t = torch.full((m, 3), -1, dtype=torch.int32, device=dev)
    ctx.bind_results_tensors(out_t[:len(idx)], ij_t[:len(idx)])
    info = ctx.work_info()
    gather = ResultGatherer(n_total, idx, dev)
    host_out = torch.empty((n_total, 5), dtype=torch.float64).pin_memory() if rank == 0 else None
    host_ij = torch.empty((n_total, 3), dtype=torch.int32).pin_memory() if rank == 0 else None

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(k=None):
        if k is not None:
            ev[k][0].record()
        ctx.run()
        if k is not None:
            ev[k][1].record()
        o, j = gather.gather(out_t, ij_t)
        if rank == 0:
            host_out.copy_(o, non_blocking=True)
            host_ij.copy_(j, non_blocking=True)
            torch.cuda.current_stream().synchronize()      # results are on the host: end of the seam

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float('nan')

    if rank == 0:
        res = host_out.numpy().copy()
        res_ij = host_ij.numpy().copy()
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total / (elapsed / args.steps)
        launches = max(info['launches'], 1)
        kern_s = kern_ms * 1e-3
        hbm_achieved = info['hbm_bytes'] / kern_s / 1e9
        mfma_achieved = 2.0 * info['macs'] / kern_s / 1e12
        # HBM traffic per launch from the committed PMC profile (rocprofv3 cannot run inside this process);
        # only quoted when the profile was collected on this very workload
        traffic, traffic_note = None, 'no PMC profile for this workload under profiles/'
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic.json')) as fh:
                tj = json.load(fh)
            if tj.get('workload') == {'size': args.size, 'grid': args.grid, 'angles': args.angles,
                                      'border': args.border if args.border == 'mixed' else int(args.border),
                                      'img_size': s}:
                traffic = tj['hbm_bytes_per_launch'] * tj['launches_per_step'] / launches
                traffic_note = 'bytes per launch, PMC FETCH_SIZE x2 + WRITE_SIZE (profiles/traffic.json): ' + tj['note']
        except (OSError, ValueError, KeyError):
            pass
        line = {
            'metric': 'PM grid-points/sec (10000x10000 px pair, 34px template)',
            'value': value, 'unit': 'grid-points/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': '%dx%d grid per GPU on one synthetic %dx%d uint8 pair, template %d px, '
                                   '%d angles [%d..%d], border %s, flags hes_norm'
                                   % (args.grid, args.grid, H, W, s, len(angles), angles[0], angles[-1], border),
                       'points_total': int(n_total), 'points_per_gpu': int(len(idx)),
                       'parallelism': 'points sharded over %d GPU(s), RCCL gather to rank 0' % world},
            'roofline': {
                # the sweep runs on v_mfma_i32_16x16x64_i8: the matrix cores are the roofline that bounds it
                'bound': 'mfma', 'achieved': mfma_achieved, 'peak': MFMA_I8_PEAK_TOPS, 'unit': 'TFLOP/s',
                'frac': mfma_achieved / MFMA_I8_PEAK_TOPS, 'traffic': traffic, 'traffic_note': traffic_note,
                # one template, two band heights: <S,4> for the one- and three-per-CU classes, <S,8> for two per CU
                'kernel': 'sid::pm_kernel_mfma<%d,4> + <%d,8>' % ((s, s) if s in (34, 35) else (0, 0)), 'launches_per_step': launches,
                'kernel_ms_per_step': kern_ms, 'avg_launch_ms': kern_ms / launches,
                'algorithmic_macs_per_step': info['macs'], 'algorithmic_bytes_per_step': info['hbm_bytes'],
                'note': 'achieved = 2 x algorithmic MACs (sum K*Rh*Rw*s*s, integer ops) / kernel time measured with HIP '
                        'events on the launch stream; the MFMA tiles carry 64 window columns for a 34-column template '
                        '(47 % padding) and a 16th all-ones template slot, so the matrix pipe itself is busier than '
                        'this figure (see DESIGN.md); VALU work around the MFMAs is the practical limiter',
                'hbm': {'achieved': hbm_achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_achieved / HBM_PEAK_GBS,
                        'note': '~10^4 MAC per HBM byte: not HBM-bound; algorithmic bytes = both images once at most'},
            },
            'setup_s': {'generate_and_upload_pair': t_gen},
        }
        # ---- parity spot check against the oracle (checker only; not timed) ----
        from oracle import c_oracle
        c_oracle.build()
        nthreads = host_cores()
        if args.check > 0:
            sel = np.random.default_rng(0).choice(n_total, size=min(args.check, n_total), replace=False)
            exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'][sel], g['r1'][sel], g['c2fg'][sel], g['r2fg'][sel],
                                            g['border'][sel], s, 0.0, angles, rot=rot, nthreads=nthreads)
            ok = (np.array_equal(res_ij[sel], exp_ij) and np.array_equal(res[sel, :4], exp[:, :4], equal_nan=True)
                  and np.allclose(res[sel, 4], exp[:, 4], rtol=1e-5, atol=1e-5, equal_nan=True))
            line['parity_check'] = {'points': int(len(sel)), 'ok': bool(ok)}
            if not ok:
                print(json.dumps(line))
                raise SystemExit('PARITY FAILURE against the CPU oracle - the number above is invalid')
        if not args.no_cpu_baseline:
            # bounded sample: every stride-th point of the same grid, sized for ~15 s of wall time
            cal = np.arange(0, n_total, max(1, n_total // (8 * nthreads)))[:8 * nthreads]
            tc = time.perf_counter()
            c_oracle.pm_batch(img1, img2, g['c1'][cal], g['r1'][cal], g['c2fg'][cal], g['r2fg'][cal],
                              g['border'][cal], s, 0.0, angles, rot=rot, nthreads=nthreads)
            rate = len(cal) / (time.perf_counter() - tc)
            n_s = args.cpu_sample or int(min(n_total, max(len(cal), rate * 15.0)))
            smp = np.linspace(0, n_total - 1, n_s).astype(np.int64)
            tc = time.perf_counter()
            c_oracle.pm_batch(img1, img2, g['c1'][smp], g['r1'][smp], g['c2fg'][smp], g['r2fg'][smp],
                              g['border'][smp], s, 0.0, angles, rot=rot, nthreads=nthreads)
            dt = time.perf_counter() - tc
            line['cpu_baseline'] = {
                'value': n_s / dt, 'unit': 'grid-points/s', 'cores': nthreads, 'kind': 'port',
                'sample': '%d evenly spaced points of the same %d-point grid, same pair/angles/borders, '
                          'oracle/pm_oracle.c (restated CPU pmlib, exact-integer NCC; not cv2) with OpenMP '
                          'over points, %.1f s' % (n_s, n_total, dt)}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
