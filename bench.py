#!/usr/bin/env python3
"""bench.py - PM grid-points/sec of the HIP pattern-matching path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode grid|stream]

N = 1 runs in this process.  N > 1 works both ways: launched by ``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or plainly
as ``python bench.py --gpus N`` - the parent then starts that same launcher as a child process before it
touches torch or the GPU (a process that has initialised the GPU is never replaced) and exits with its code.

--mode grid (default; BASELINE.json configs[1] at N = 1, configs[2] at N > 1)
    one synthetic 10000x10000 uint8 pair, 200x200 grid, 34 px template, 15 trial angles [-7..7], mixed
    search border 20..50 px; the five kernel-input vectors and both images are resident in HBM before the
    timed region.  A step = one pass of the hot path over the whole grid: kernel launches for every point,
    the gather of the packed (N,5) float64 + (N,3) int32 results to rank 0 (one RCCL gather when N > 1) and
    their copy to the host (the reference's seam ends with the results in the parent, pmlib.py:444,462).  With one
    rank the kernels write their 52 B per point straight into the pinned host buffer (zero copy; the step ends with the
    stream synchronised and the results readable on the host - the parity check reads exactly that buffer).
    N > 1 is STRONG scaling by default - the same 200x200 grid cut into runs of equal estimated cost (by search border)
    (sea_ice_drift_amd/dist.py), the cuts then moved by --rebalance (3) rounds of MEASURED feedback during set-up, before the
    warm-up: every rank times its own kernels, one all_gather, dist.rebalance_cuts; reported under config.rebalance - and
    the weak figure ((200*N)x200 grid, 40 000 points per GPU) is measured after it and reported under "weak_scaling";
    ``--scaling weak`` makes the weak workload the headline.

--mode ftpm (BASELINE.json configs[3])
    the public chain on the same pair: SeaIceDrift.get_drift_FT (key-point detector and Hamming matcher on the GPU, the
    reference's filters on the host) feeding SeaIceDrift.get_drift_PM on the 200x200 grid.  A step = both calls, host
    arrays in, host grids out (the pair upload included); value = valid grid points / step time; ft_ms / pm_ms, the number
    of feature-tracking vectors and the share of them within 3 px of the synthetic displacement field are reported, and
    --check points of the PM result are compared with the C oracle fed with the same FT-derived first guess.

--mode stream (BASELINE.json configs[4])
    a batch of --pairs (16) synthetic 10000x10000 pairs in pinned host memory, dealt round-robin to the ranks;
    every rank streams its pairs through the two device slots of its handle (sid_pm_upload_pair on the copy
    stream while the kernels of the previous pair run) and fetches each result block.  A step = the whole
    batch; value = points of all pairs / time, uploads included.  No collective on the data path.

Prints ONE JSON line on rank 0 with a "roofline" object measured live from HIP events around the kernel
launches and a "cpu_baseline" object (B2 = the C oracle with OpenMP over points, plus B1 = the
reference-shaped Python/Pool loop under "b1") timed on bounded samples of the same workload.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks used for the roofline fractions (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
# dense int8 MFMA = 2x the bf16 dense peak (the guide: 'I8 ~2x bf16 rate'; 4 855 TOP/s measured here
# with tools/ubench/valu_rates.hip); integer ops, quoted in the contract's 'TFLOP/s' unit field
MFMA_I8_PEAK_TOPS = 5000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default 20; 3 in stream mode)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed steps (default 3; 1 in stream mode)')
    ap.add_argument('--mode', choices=('grid', 'stream', 'ftpm'), default='grid')
    ap.add_argument('--pairs', type=int, default=16, help='stream mode: pairs in the batch (all ranks together)')
    ap.add_argument('--size', type=int, default=10000, help='image side in pixels')
    ap.add_argument('--grid', type=int, default=200, help='grid points per side')
    ap.add_argument('--border', default='mixed', help="'mixed' or a fixed border in pixels")
    ap.add_argument('--angles', type=int, default=7, help='trial angles are -A..A in 1 degree steps')
    ap.add_argument('--img-size', type=int, default=34)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default=None,
                    help='N > 1, grid mode: strong (default) = the same grid sharded; weak = grid rows x N')
    ap.add_argument('--no-weak', action='store_true', help='skip the secondary weak-scaling measurement')
    ap.add_argument('--rebalance', type=int, default=3,
                    help='N > 1, strong scaling: rounds of measured feedback on the shard cuts BEFORE the warm-up (every rank times '
                         'its kernels over 3 steps, one all_gather, dist.rebalance_cuts; the cuts with the smallest slowest rank are kept); 0 = the estimated cuts')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-seam', action='store_true', help="skip the seam-inclusive timings (sid_pm_batch host-in / host-out, set_points) reported under 'seam_inclusive'")
    ap.add_argument('--no-also-defaults', action='store_true',
                    help="N = 1, grid mode: skip the second measurement on the reference's DEFAULT configuration (img_size 35, "
                         "angles [-3, 0, 3]; pmlib.py:118,329) that is reported under 'reference_defaults'")
    ap.add_argument('--force-collective', action='store_true',
                    help='N = 1: initialise a one-rank nccl (RCCL) group and run the broadcast, the all_reduce and the gathers '
                         'of the N-GPU path on it (exercises the RCCL code path on a one-GPU box)')
    ap.add_argument('--cpu-sample', type=int, default=0,
                    help='points in the B2 CPU sample (0 = auto: the whole grid when that takes <= ~30 s, else ~12 s worth)')
    ap.add_argument('--check', type=int, default=64,
                    help='points verified against the oracle after timing when the CPU baseline (which checks the whole grid) is off')
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = {'stream': 3, 'ftpm': 5}.get(args.mode, 20)
    if args.warmup is None:
        args.warmup = {'stream': 1, 'ftpm': 2}.get(args.mode, 3)
    if args.scaling is None:
        args.scaling = 'strong'
    return args


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: run torch.distributed.run as a CHILD (this process has
    not imported torch and never touches the GPU) and hand its exit code on."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, host_cores() // args.gpus)))
    return subprocess.call(cmd, env=env)


def lib_md5():
    from sea_ice_drift_amd import _capi
    with open(_capi.LIB_PATH, 'rb') as fh:
        return hashlib.md5(fh.read()).hexdigest()


class GridRun(object):
    """The resident state of one grid workload on this rank + its step function."""

    def __init__(self, args, dev, world, rank, local_rank, t1, t2, n_rows, angles, rot):
        import torch
        from sea_ice_drift_amd import _capi, synthetic as syn
        self.torch = torch
        s = args.img_size
        H = W = args.size
        border = args.border if args.border == 'mixed' else int(args.border)
        self.g = g = syn.make_grid(H, W, (n_rows, args.grid), border=border)
        self.n_total = g['c1'].size
        from sea_ice_drift_amd.dist import shard_cuts_by_cost
        self.args, self.dev, self.s, self.angles, self.rot = args, dev, s, angles, rot
        self.rank, self.world = rank, world
        self.order, self.cuts, self.cost = shard_cuts_by_cost(g['border'], world, s, len(angles))   # (world 1: all points)
        self.ctx = _capi.PMContext(local_rank)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.ctx.bind_pair_tensors(t1, t2)
        self.reshard(self.cuts)

    def reshard(self, cuts):
        """This rank's run of the border-ordered points under `cuts` (every rank calls this with the same cuts: the gatherer's
        set-up is a collective)."""
        from sea_ice_drift_amd.dist import PackedGatherer, indices_of_cut
        g = self.g
        self.cuts = cuts
        self.idx = idx = indices_of_cut(self.order, cuts, self.rank)
        self.ctx.set_points(g['c1'][idx], g['r1'][idx], g['c2fg'][idx], g['r2fg'][idx], g['border'][idx], self.s, 0.0,
                            self.angles, rot=self.rot)
        self.gather = PackedGatherer(self.n_total, idx, self.dev, force_collective=self.args.force_collective,
                                     timing=self.world > 1 or self.args.force_collective)
        out_t, ij_t = self.gather.local_views()
        self.ctx.bind_results_tensors(out_t, ij_t)
        self.info = self.ctx.work_info()

    def kernel_ms(self, steps=3):
        """This rank's kernels alone (HIP events on the launch stream around `steps` runs, after one untimed run)."""
        torch = self.torch
        self.ctx.run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            self.ctx.run()
        b.record()
        b.synchronize()
        self.ctx.check()
        return a.elapsed_time(b) / steps

    def rebalance(self, rounds):
        """Measured feedback on the cuts (N > 1): the estimate prices a launch as cost + tail and leaves the slowest of eight
        shards ~4 % above the mean; ``dist.rebalance_with_feedback`` - a public helper of the library's
        N-GPU path, not a benchmark-only trick - has every rank time its own kernels, makes the times known everywhere with one
        all_gather per round (a rank whose kernels failed still takes part, then every rank raises), moves the cuts and keeps
        the ones with the smallest slowest rank.  Set-up work, before the warm-up."""
        from sea_ice_drift_amd.dist import rebalance_with_feedback
        res = rebalance_with_feedback(self.kernel_ms, self.reshard, self.cost, self.cuts, rounds, self.dev)
        res.pop('cuts')
        return res

    def step(self, ev=None):
        if ev is not None:
            ev[0].record()
        self.ctx.run()
        if ev is not None:
            ev[1].record()
        self.gather.gather_to_host()            # rank 0: results on the host = end of the seam
        self.ctx.check()                        # (stream synchronised above) no valid point was refused by its launch

    def poisoned_step(self):
        """One more, untimed step whose result buffers were overwritten (NaN / -1) first: what the parity check reads is
        the output of THIS step - a step that launched or gathered nothing cannot pass on the values of an earlier one."""
        self.gather.poison()
        self.step()

    def results(self):
        return self.gather.host_results()

    def close(self):
        self.ctx.close()


def timed_steps(torch, dist, world, run, steps, warmup, discard_warmup=None):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        run.step()
    fence()
    if discard_warmup is not None:
        discard_warmup()
    t0 = time.perf_counter()
    for k in range(steps):
        run.step(ev[k])
    fence()
    elapsed = time.perf_counter() - t0
    timed_steps.local_elapsed = elapsed                    # this rank's own clock (per-rank breakdown of the stream mode)
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(steps, 1)
    return elapsed, kern_ms


def traffic_from_profile(args, launches):
    """HBM bytes per launch from the committed PMC profile (rocprofv3 cannot run inside this process).  Quoted
    only for the workload it was collected on, and with a statement whether it was this very build."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as fh:
            tj = json.load(fh)
    except (OSError, ValueError):
        return None, 'no PMC profile under profiles/'
    want = {'size': args.size, 'grid': args.grid, 'angles': args.angles,
            'border': args.border if args.border == 'mixed' else int(args.border), 'img_size': args.img_size}
    if tj.get('workload') != want:
        return None, 'profiles/traffic.json was collected on another workload'
    same = tj.get('so_md5') == lib_md5()
    note = ('bytes per launch, PMC FETCH_SIZE x2 + WRITE_SIZE (profiles/traffic.json, %s): %s'
            % ('collected on THIS build of libsid_pm.so (md5 match)' if same else
               'collected on an EARLIER build (library md5 differs: %s vs %s)' % (tj.get('so_md5'), lib_md5()),
               tj.get('note', '')))
    return tj['hbm_bytes_per_launch'] * tj['launches_per_step'] / max(launches, 1), note


def cpu_baselines(args, img1, img2, g, n_total, angles, rot, s):
    """B2 (C oracle, OpenMP over points) and B1 (reference-shaped Python/Pool loop) on bounded samples.

    Returns (cpu_baseline object, oracle results of the B2 sample): the B2 run evaluates every point of the grid
    whenever that takes at most ~30 s on this box's cores (otherwise an evenly spaced ~12 s sample), and its
    results - with the peak / runner-up gap of every point - are handed to parity_block, so that the timed GPU
    output is compared with the oracle on all of them."""
    import numpy as np
    from oracle import b1_baseline, c_oracle
    nthreads = host_cores()
    pick = lambda sel: [g[k][sel] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    cal = np.arange(0, n_total, max(1, n_total // (8 * nthreads)))[:8 * nthreads]
    tc = time.perf_counter()
    c_oracle.pm_batch(img1, img2, *pick(cal), s, 0.0, angles, rot=rot, nthreads=nthreads)
    rate = len(cal) / (time.perf_counter() - tc)
    if args.cpu_sample:
        n_s = min(args.cpu_sample, n_total)
    else:
        n_s = n_total if n_total <= rate * 30.0 else int(max(len(cal), rate * 12.0))
    smp = np.arange(n_total) if n_s >= n_total else np.linspace(0, n_total - 1, n_s).astype(np.int64)
    tc = time.perf_counter()
    exp, exp_ij, gap = c_oracle.pm_batch(img1, img2, *pick(smp), s, 0.0, angles, rot=rot, nthreads=nthreads, want_gap=True)
    dt = time.perf_counter() - tc
    out = {'value': len(smp) / dt, 'unit': 'grid-points/s', 'cores': nthreads, 'kind': 'port',
           'sample': 'B2: %s of the same %d-point grid, same pair/angles/borders, '
                     'oracle/pm_oracle.c (restated CPU pmlib, exact-integer NCC; not cv2) with OpenMP over points, '
                     '%.1f s' % ('ALL points' if len(smp) == n_total else '%d evenly spaced points' % len(smp), n_total, dt)}
    # B1: ~8 s; per-point Python tasks under a fork pool, the reference's own structure (pmlib.py:436-448)
    cal1 = np.linspace(0, n_total - 1, 2 * nthreads).astype(np.int64)
    tc = time.perf_counter()
    b1_baseline.run(img1, img2, *pick(cal1), s, 0.0, angles, processes=nthreads)
    rate1 = len(cal1) / (time.perf_counter() - tc)
    n1 = int(min(n_total, max(len(cal1), rate1 * 8.0)))
    smp1 = np.linspace(0, n_total - 1, n1).astype(np.int64)
    tc = time.perf_counter()
    res1 = b1_baseline.run(img1, img2, *pick(smp1), s, 0.0, angles, processes=nthreads)
    dt1 = time.perf_counter() - tc
    out['b1'] = {'value': n1 / dt1, 'unit': 'grid-points/s', 'cores': nthreads, 'kind': 'port',
                 'sample': 'B1: %d points, one Python task per point under multiprocessing.Pool(%d) as pmlib.py:436-448; '
                           'scipy affine_transform templates, float32-FFT correlation standing in for cv2 (absent), '
                           'NumPy Hessian; %.1f s' % (n1, nthreads, dt1)}
    # Differential run of a float32-FFT matcher - OpenCV's route for TM_CCOEFF_NORMED, the one call the oracle cannot pin
    # (cv2 is absent) - against the exact-integer oracle: B1's results on its sample and on EVERY point whose peak lies
    # within 1e-4 of the runner-up (the ones a noisy matcher could flip).  A flip = peak position or angle differs.
    if len(smp) == n_total:
        close = np.flatnonzero(np.isfinite(gap) & (gap > 0) & (gap < 1e-4))
        res_c = b1_baseline.run(img1, img2, *pick(close), s, 0.0, angles, processes=nthreads) if close.size else np.zeros((0, 5))

        def flips(sel, got):
            want = exp[sel]
            both = np.isfinite(want[:, 0]) & np.isfinite(got[:, 0])
            moved = both & ((got[:, 0] != want[:, 0]) | (got[:, 1] != want[:, 1]) | (got[:, 2] != want[:, 2]))
            dr = np.abs(got[both, 3] - want[both, 3])
            return {'points': int(len(sel)), 'peak_or_angle_differs': int(moved.sum()),
                    'nan_disagreements': int((np.isfinite(want[:, 0]) != np.isfinite(got[:, 0])).sum()),
                    'max_abs_r_difference': float(dr.max()) if dr.size else None}
        out['fft_f32_differential'] = {
            'matcher': 'oracle/b1_baseline._ccoeff_normed_fft: float32 FFT correlation, float64 normalisation (OpenCV-like), '
                       'scipy affine_transform templates, NumPy argmax - against oracle/pm_oracle.c (exact integers)',
            'b1_sample': flips(smp1, res1), 'points_with_gap_below_1e-4': flips(close, res_c)}
    return out, (smp, exp, exp_ij, gap)


def parity_block(args, img1, img2, g, n_total, res, res_ij, angles, rot, s, oracle_run=None):
    """The timed GPU output against the C oracle + the cv2-flip exposure census (checker only, not timed).
    oracle_run = (indices, out, ij, gap) of the B2 baseline run (every point of the grid when affordable);
    without it (--no-cpu-baseline) a random subsample of --check points is evaluated here."""
    import numpy as np
    from oracle import c_oracle
    c_oracle.build()
    if oracle_run is not None:
        sel, exp, exp_ij, gap = oracle_run
    else:
        sel = np.random.default_rng(0).choice(n_total, size=min(args.check, n_total), replace=False)
        exp, exp_ij, gap = c_oracle.pm_batch(img1, img2, g['c1'][sel], g['r1'][sel], g['c2fg'][sel], g['r2fg'][sel],
                                             g['border'][sel], s, 0.0, angles, rot=rot, nthreads=host_cores(), want_gap=True)
    bad_ij = ~np.all(res_ij[sel] == exp_ij, axis=1)
    a, b = res[sel, :4], exp[:, :4]
    bad_v = ~np.all((a == b) | (np.isnan(a) & np.isnan(b)), axis=1)
    bad_h = ~np.isclose(res[sel, 4], exp[:, 4], rtol=1e-5, atol=1e-5, equal_nan=True)
    ok = not (bad_ij.any() or bad_v.any() or bad_h.any())
    fin = np.isfinite(gap)
    hd = np.abs(res[sel, 4] - exp[:, 4])
    return {'points': int(len(sel)), 'of_grid_points': int(n_total), 'ok': bool(ok),
            'mismatches': {'peak_or_angle_index': int(bad_ij.sum()), 'c2_r2_a_r_bits': int(bad_v.sum()), 'h_beyond_1e-5': int(bad_h.sum())},
            'nan_points': int(np.isnan(exp[:, 0]).sum()),
            'max_abs_h_difference': float(np.nanmax(hd)) if np.isfinite(hd).any() else None,
            'rule': 'peak row/col/angle index, c2, r2, a, r: bit-exact; h: rtol = atol = 1e-5',
            # a float32/DFT matcher such as cv2's carries ~1e-6 noise: peaks this close to the runner-up could flip there
            'cv2_flip_exposure': {'gap_below_1e-6': int(((gap > 0) & (gap < 1e-6) & fin).sum()),
                                  'gap_below_1e-5': int(((gap > 0) & (gap < 1e-5) & fin).sum()),
                                  'gap_below_1e-4': int(((gap > 0) & (gap < 1e-4) & fin).sum()),
                                  'exact_ties': int(((gap == 0) & fin).sum()), 'of_points': int(fin.sum()),
                                  'min_positive_gap': float(gap[(gap > 0) & fin].min()) if ((gap > 0) & fin).any() else None}}


def seam_inclusive(args, img1, img2, g, s, angles, rot, local_rank, res, res_ij):
    """SURVEY.md section 8(d) defines the metric 'from entering the dispatch'; the headline `value` starts with the pair, the point
    records and the sampling tables resident in HBM (the contract's rule for `value`).  Here is what lies before that, timed on
    the same workload, host buffers in / host buffers out - reported beside `value`, never as it:
      sid_pm_batch_ms            the one-shot C call (include/sid_pm.h): create a handle, upload the pair (2 x 100 MB over PCIe from
                                 pageable memory), set_points (validate, classify, one upload), the kernels, fetch, destroy
      dispatch_resident_pair_ms  handle and pair resident (what pattern_matching pays per call once its upload is done):
                                 set_points + run + fetch
      set_points_ms              set_points alone (host classification of the points + uploads of records and tables)"""
    import numpy as np
    from sea_ice_drift_amd import _capi
    v = [g[k] for k in ('c1', 'r1', 'c2fg', 'r2fg', 'border')]
    n = len(v[0])

    def med(fn, reps=3):
        fn()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            t.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(t)), out
    batch_ms, (out, ij) = med(lambda: _capi.pm_batch(img1, img2, *v, s, 0.0, angles, rot=rot))
    same = bool(np.array_equal(out, res, equal_nan=True) and np.array_equal(ij, res_ij))
    with _capi.PMContext(local_rank) as ctx:
        ctx.upload_pair(img1, img2)
        ctx.sync()
        sp_ms, _ = med(lambda: ctx.set_points(*v, s, 0.0, angles, rot=rot))

        def dispatch():
            ctx.set_points(*v, s, 0.0, angles, rot=rot)
            ctx.run()
            return ctx.fetch()
        disp_ms, _ = med(dispatch)
    return {'sid_pm_batch_ms': batch_ms, 'sid_pm_batch_points_per_s': n / (batch_ms * 1e-3),
            'dispatch_resident_pair_ms': disp_ms, 'dispatch_resident_pair_points_per_s': n / (disp_ms * 1e-3),
            'set_points_ms': sp_ms, 'equals_the_resident_results': same,
            'what': 'host buffers in / host buffers out, median of 3 calls after one untimed call; sid_pm_batch = create + upload of the '
                    '%d MB pair from pageable memory + set_points + kernels + fetch + destroy (SURVEY.md 8(d): from entering the dispatch); '
                    'never `value`' % (2 * img1.size // 1000000)}


def rank_devices(torch, dist, dev):
    """One record per rank of the process group - host, device index, name, architecture, compute units (256 = 8 XCDs x 32), uuid,
    PCI address - gathered with all_gather_object at set-up: evidence in the bench line of how many GPUs the collective really spans."""
    p = torch.cuda.get_device_properties(dev)
    mine = {'rank': dist.get_rank() if dist.is_initialized() else 0, 'host': socket.gethostname(), 'device_index': int(dev.index or 0),
            'name': p.name, 'arch': getattr(p, 'gcnArchName', None), 'compute_units': int(p.multi_processor_count),
            'xcds': int(p.multi_processor_count) // 32 if int(p.multi_processor_count) % 32 == 0 else None,
            'uuid': str(getattr(p, 'uuid', '')) or None,
            'pci_bus_id': ('%04x:%02x:%02x' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, getattr(p, 'pci_device_id', 0))) if hasattr(p, 'pci_bus_id') else None,
            'hbm_gb': round(p.total_memory / 1e9, 1)}
    if not (dist.is_available() and dist.is_initialized()):
        return [mine]
    rows = [None] * dist.get_world_size()
    dist.all_gather_object(rows, mine)
    return rows


def grid_mode(args, torch, dist, dev, world, rank, local_rank):
    import numpy as np
    from sea_ice_drift_amd import synthetic as syn
    from sea_ice_drift_amd.pmlib import rotation_table
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
    if world > 1 or args.force_collective:
        dist.broadcast(t1, 0)
        dist.broadcast(t2, 0)
    t_gen = time.time() - t_gen
    rot = rotation_table(angles, 0.0, s)          # rotation terms via NumPy, as the reference

    headline_rows = args.grid * world if (args.scaling == 'weak' and world > 1) else args.grid
    run = GridRun(args, dev, world, rank, local_rank, t1, t2, headline_rows, angles, rot)
    rebalanced = run.rebalance(args.rebalance) if (world > 1 and args.scaling == 'strong' and args.rebalance > 0) else None
    run.gather.timings()                                   # (reset: construction and warm-up are not the timed steps)
    elapsed, kern_ms = timed_steps(torch, dist, world, run, args.steps, args.warmup, discard_warmup=run.gather.timings)
    exchange = run.gather.timings()
    # every rank's share of a step on rank 0 (one all_gather): the slowest rank of an N-GPU run is then identifiable
    from sea_ice_drift_amd.dist import per_rank_breakdown
    per_rank = per_rank_breakdown([kern_ms, exchange['gather_ms'], exchange['unpermute_ms'], exchange['d2h_ms'], len(run.idx),
                                   run.info['launches']], dev) if (world > 1 or args.force_collective) else None
    devices_seen = rank_devices(torch, dist, dev) if (world > 1 or args.force_collective) else None
    run.poisoned_step()                                    # untimed; the parity check below reads this step's output
    res, res_ij = run.results() if rank == 0 else (None, None)
    n_total, info, g = run.n_total, run.info, run.g
    zero_copy, device_unpermute = getattr(run.gather, 'zero_copy', False), getattr(run.gather, 'device_unpermute', False)
    import numpy as _np
    n_local, points_all = len(run.idx), [int(v) for v in _np.diff(run.cuts)]
    run.close()

    weak = None
    if world > 1 and args.scaling == 'strong' and not args.no_weak:
        wrun = GridRun(args, dev, world, rank, local_rank, t1, t2, args.grid * world, angles, rot)
        ws = max(3, args.steps // 4)
        w_el, _ = timed_steps(torch, dist, world, wrun, ws, 1)
        weak = {'value': wrun.n_total / (w_el / ws), 'unit': 'grid-points/s', 'ms_per_step': w_el / ws * 1e3,
                'steps': ws, 'points_total': int(wrun.n_total),
                'workload': '(%d*%d)x%d grid on the same pair: %d points per GPU' % (args.grid, world, args.grid,
                                                                                  wrun.n_total // world)}
        wrun.close()

    if rank != 0:
        return None
    ms_per_step = elapsed / args.steps * 1e3
    launches = max(info['launches'], 1)
    kern_s = kern_ms * 1e-3
    # rank 0's own kernels ran on its shard: the work of that shard over its kernel time
    mfma_achieved = 2.0 * info['macs'] / kern_s / 1e12
    hbm_achieved = info['hbm_bytes'] / kern_s / 1e9
    traffic, traffic_note = traffic_from_profile(args, launches) if world == 1 else (None, 'single-GPU profile only')
    border = args.border if args.border == 'mixed' else int(args.border)
    line = {
        'metric': 'PM grid-points/sec (10000x10000 px pair, 34px template)',
        'value': n_total / (elapsed / args.steps), 'unit': 'grid-points/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': args.scaling if world > 1 else 'strong', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': '%dx%d grid on one synthetic %dx%d uint8 pair, template %d px, %d angles [%d..%d], '
                               'border %s, flags hes_norm' % (headline_rows, args.grid, H, W, s, len(angles), angles[0],
                                                             angles[-1], border),
                   'points_total': int(n_total), 'points_per_gpu': int(n_local),
                   # (shards of equal estimated cost are unequal in length: rank 0 holds the largest windows)
                   'points_per_gpu_all': points_all,
                   'results': ('written by the kernels into pinned host memory (zero copy); step = launches + stream synchronise'
                               if zero_copy else 'device block -> one gather -> one kernel that un-permutes into pinned host memory'
                               if device_unpermute else 'device block -> (gather, un-permutation) -> one copy to pinned host memory'),
                   'parity_reads': 'the output of one more, untimed step run after the result buffers were overwritten with NaN / -1',
                   'parallelism': ('single GPU, no collective' if world == 1 else
                                   'points cut into %d runs of equal estimated cost (neighbouring borders per GPU)%s, one RCCL gather of the packed '
                                   'result blocks to rank 0' % (world, ', cuts moved by %d rounds of measured feedback before the warm-up' % args.rebalance
                                                                if rebalanced else ''))},
        'roofline': {
            # the sweep runs on v_mfma_i32_16x16x64_i8: the matrix cores are the roofline that bounds it
            'bound': 'mfma', 'achieved': mfma_achieved, 'peak': MFMA_I8_PEAK_TOPS, 'unit': 'TFLOP/s',
            'frac': mfma_achieved / MFMA_I8_PEAK_TOPS, 'traffic': traffic, 'traffic_note': traffic_note,
            'kernel': ('sid::pm_kernel_rp<%d,...> (row-pair sweep; one launch per residency class: 4 / 3 / 3 / 2 / 1 workgroups per CU)' % s
                       if s in (34, 35) and not os.environ.get('SID_PM_NO_RP') else
                       'sid::pm_kernel_mfma<%d,...> (classic sweep: one instantiation per band height / pairing)' % (s if s in (34, 35) else 0)),
            'launches_per_step': launches, 'kernel_ms_per_step': kern_ms, 'avg_launch_ms': kern_ms / launches,
            'algorithmic_macs_per_step': info['macs'], 'algorithmic_bytes_per_step': info['hbm_bytes'],
            'note': 'rank 0 share: achieved = 2 x algorithmic MACs (sum K*Rh*Rw*s*s, integer ops) / kernel time from HIP '
                    'events on the launch stream; padding of the MFMA tiles (template columns, the all-ones slot) is '
                    'not counted as work (DESIGN.md section 6)',
            'hbm': {'achieved': hbm_achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_achieved / HBM_PEAK_GBS,
                    'note': '~10^4 MAC per HBM byte: not HBM-bound; algorithmic bytes = both images once at most'},
        },
        'setup_s': {'generate_and_upload_pair': t_gen},
    }
    if weak is not None:
        line['weak_scaling'] = weak
    if rebalanced is not None:
        line['config']['rebalance'] = rebalanced
    if per_rank is not None and exchange['steps'] > 0:
        # where a step of the N-GPU path goes: rank 0's own kernels, then the exchange step (HIP events on the launch
        # stream; the gather also holds the wait for the slowest rank), the un-permutation into pinned host memory - and the
        # same numbers of EVERY rank, with the slowest rank's kernels named (max / argmax)
        kcol = per_rank[:, 0]
        line['step_breakdown_ms'] = {'kernel_ms': kern_ms, 'gather_ms': exchange['gather_ms'], 'unpermute_ms': exchange['unpermute_ms'],
                                     'd2h_ms': exchange['d2h_ms'], 'backend': dist.get_backend() if dist.is_initialized() else None,
                                     'results': ('one kernel (sid_pm_unpermute) reads the gathered blocks and writes pinned host memory'
                                                 if getattr(run.gather, 'device_unpermute', False) else 'index_select x2 + one copy to the host'),
                                     'collectives': 'broadcast x2 (pair), all_reduce(MAX) + gather (indices) at set-up, one gather of '
                                                    'the packed block per step',
                                     # what the process group itself reports: the ranks it holds and the device each one runs on
                                     # (distinct uuids / PCI addresses = N GPUs; equal ones = a dry run of N ranks sharing a GPU)
                                     'world_size_seen_by_the_process_group': dist.get_world_size() if dist.is_initialized() else 1,
                                     'devices': devices_seen,
                                     'distinct_devices': len({(d.get('uuid'), d.get('pci_bus_id'), d.get('host')) for d in devices_seen}) if devices_seen else None,
                                     'per_rank': {'kernel_ms': [float(v) for v in kcol], 'gather_ms': [float(v) for v in per_rank[:, 1]],
                                                  'points': [int(v) for v in per_rank[:, 4]], 'launches': [int(v) for v in per_rank[:, 5]]},
                                     'slowest_rank': int(kcol.argmax()), 'slowest_kernel_ms': float(kcol.max()),
                                     'fastest_kernel_ms': float(kcol.min())}
    # The reference's DEFAULT configuration is timed HERE, straight after the headline's steps and before the CPU baselines:
    # behind ~12 s of 16-core host work with the GPU idle its steps read 0.5 ms above their own kernel time (round 4's
    # driver run); its oracle comparison follows after the baselines.
    also_defaults = world == 1 and not args.no_also_defaults and not args.force_collective and args.img_size == 34 and args.angles == 7
    defaults_run = defaults_timed(args, torch, dist, dev, local_rank, t1, t2) if also_defaults else None
    if world == 1 and not args.force_collective and not args.no_seam:
        line['seam_inclusive'] = seam_inclusive(args, img1, img2, g, s, angles, rot, local_rank, res, res_ij)
    oracle_run = None
    if not args.no_cpu_baseline:
        line['cpu_baseline'], oracle_run = cpu_baselines(args, img1, img2, g, n_total, angles, rot, s)
    if args.check > 0 or oracle_run is not None:
        line['parity_check'] = parity_block(args, img1, img2, g, n_total, res, res_ij, angles, rot, s, oracle_run)
        if 'fft_f32_differential' in line.get('cpu_baseline', {}):
            # how far a float32-FFT matcher (OpenCV's route, unpinnable here) parts from the exact-integer specification
            line['parity_check']['fft_f32_flips'] = line['cpu_baseline'].pop('fft_f32_differential')
        if not line['parity_check']['ok']:
            print(json.dumps(line))
            raise SystemExit('PARITY FAILURE against the CPU oracle - the number above is invalid')
    if defaults_run is not None:
        line['reference_defaults'] = defaults_check(args, defaults_run, img1, img2)
        if not line['reference_defaults']['parity_check']['ok']:
            print(json.dumps(line))
            raise SystemExit("PARITY FAILURE against the CPU oracle on the reference's default configuration")
    return line


def defaults_timed(args, torch, dist, dev, local_rank, t1, t2):
    """The reference's DEFAULT configuration on the same pair and grid (pmlib.py:118 angles = [-3, 0, 3]; pmlib.py:329
    img_size = 35), timed like the headline (same steps, same warm-up, HIP events for the kernel time)."""
    import copy
    from sea_ice_drift_amd.pmlib import rotation_table
    a2 = copy.copy(args)
    a2.img_size, a2.force_collective = 35, False
    angles = [-3, 0, 3]
    rot = rotation_table(angles, 0.0, a2.img_size)
    run = GridRun(a2, dev, 1, 0, local_rank, t1, t2, a2.grid, angles, rot)
    elapsed, kern_ms = timed_steps(torch, dist, 1, run, args.steps, args.warmup)
    run.poisoned_step()
    res, res_ij = run.results()
    out = {'a2': a2, 'angles': angles, 'rot': rot, 'elapsed': elapsed, 'kern_ms': kern_ms, 'res': res, 'res_ij': res_ij,
           'g': run.g, 'n_total': run.n_total, 'info': run.info}
    run.close()
    return out


def defaults_check(args, d, img1, img2):
    """Every point of the default configuration's timed output against the C oracle; the block of the bench line."""
    import numpy as np
    from oracle import c_oracle
    a2, angles, rot, res, res_ij, g, n_total, info = d['a2'], d['angles'], d['rot'], d['res'], d['res_ij'], d['g'], d['n_total'], d['info']
    elapsed, kern_ms = d['elapsed'], d['kern_ms']
    c_oracle.build()
    tc = time.perf_counter()
    exp, exp_ij = c_oracle.pm_batch(img1, img2, g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], a2.img_size, 0.0, angles,
                                    rot=rot, nthreads=host_cores())
    dt = time.perf_counter() - tc
    bad_ij = ~np.all(res_ij == exp_ij, axis=1)
    a, b = res[:, :4], exp[:, :4]
    bad_v = ~np.all((a == b) | (np.isnan(a) & np.isnan(b)), axis=1)
    bad_h = ~np.isclose(res[:, 4], exp[:, 4], rtol=1e-5, atol=1e-5, equal_nan=True)
    ok = not (bad_ij.any() or bad_v.any() or bad_h.any())
    ms = elapsed / args.steps * 1e3
    return {'workload': '%dx%d grid on the same pair, template 35 px (pmlib.py:329), angles [-3, 0, 3] (pmlib.py:118), border %s'
                        % (a2.grid, a2.grid, args.border),
            'value': n_total / (elapsed / args.steps), 'unit': 'grid-points/s', 'ms_per_step': ms,
            'kernel_ms_per_step': kern_ms, 'step_minus_kernel_ms': ms - kern_ms, 'launches_per_step': info['launches'],
            'timed': 'straight after the headline steps, before the CPU baselines',
            'roofline_frac_mfma': 2.0 * info['macs'] / (kern_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
            'cpu_oracle_points_per_s': n_total / dt,
            'parity_check': {'points': int(n_total), 'ok': bool(ok),
                             'mismatches': {'peak_or_angle_index': int(bad_ij.sum()), 'c2_r2_a_r_bits': int(bad_v.sum()),
                                            'h_beyond_1e-5': int(bad_h.sum())}}}


def stream_mode(args, torch, dist, dev, world, rank, local_rank):
    """BASELINE config 5: every rank streams its share of the batch through its two device slots."""
    import numpy as np
    from sea_ice_drift_amd import _capi, synthetic as syn
    from sea_ice_drift_amd.pmlib import rotation_table
    angles = list(range(-args.angles, args.angles + 1))
    s = args.img_size
    H = W = args.size
    border = args.border if args.border == 'mixed' else int(args.border)
    mine = list(range(rank, args.pairs, world))
    t_gen = time.time()
    base1, base2 = syn.make_pair(H, W)
    # pair p = the base pair rolled by 37 p rows and 53 p columns (both images alike: the drift field moves along);
    # distinct pixels per pair, generated in seconds, pinned so that the uploads are truly asynchronous
    pairs = {}
    for p in mine:
        a = torch.from_numpy(np.roll(base1, (37 * p, 53 * p), axis=(0, 1))).pin_memory()
        b = torch.from_numpy(np.roll(base2, (37 * p, 53 * p), axis=(0, 1))).pin_memory()
        pairs[p] = (a, b)
    t_gen = time.time() - t_gen
    g = syn.make_grid(H, W, args.grid, border=border)
    n_pts = g['c1'].size
    rot = rotation_table(angles, 0.0, s)
    ctx = _capi.PMContext(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    first = pairs[mine[0]] if mine else (torch.from_numpy(base1), torch.from_numpy(base2))
    ctx.upload_pair(first[0].numpy(), first[1].numpy(), slot=0)
    ctx.set_points(g['c1'], g['r1'], g['c2fg'], g['r2fg'], g['border'], s, 0.0, angles, rot=rot)
    info = ctx.work_info()
    results = {}
    # results: the kernels write straight into pinned host buffers, one per upload slot (zero copy; a pair's results are
    # complete when its launches are - no copy, no allocation per pair); the pairs checked afterwards keep a copy
    zero_copy = os.environ.get('SID_PM_NO_ZERO_COPY') is None
    keep = set(mine[:3])
    if zero_copy:
        host_out = [torch.empty((n_pts, 5), dtype=torch.float64, pin_memory=True) for _ in range(2)]
        host_ij = [torch.empty((n_pts, 3), dtype=torch.int32, pin_memory=True) for _ in range(2)]

    class Run(object):
        def step(self, ev=None):
            # pair k matches from slot k % 2 while pair k+1 uploads into the other slot
            if mine:
                a, b = pairs[mine[0]]
                ctx.upload_pair(a.numpy(), b.numpy(), slot=0)
            for k, p in enumerate(mine):
                if k + 1 < len(mine):
                    a, b = pairs[mine[k + 1]]
                    ctx.upload_pair(a.numpy(), b.numpy(), slot=(k + 1) % 2, select=False)
                ctx.select_pair(k % 2)
                if ev is not None and k == 0:
                    ev[0].record()
                if zero_copy:
                    ctx.bind_results_tensors(host_out[k % 2], host_ij[k % 2])
                ctx.run()
                if ev is not None and k == 0:
                    ev[1].record()
                if zero_copy:
                    ctx.sync()                                   # results of pair k readable on the host
                    if p in keep:
                        results[p] = (host_out[k % 2].numpy().copy(), host_ij[k % 2].numpy().copy())
                else:
                    results[p] = ctx.fetch()

    run = Run()
    elapsed, kern_first_ms = timed_steps(torch, dist, world, run, args.steps, args.warmup)
    # the kernels of ONE pair in steady state: both slots uploaded, nothing for the launches to wait for (the events around
    # the first pair of a step also hold that pair's wait for its own upload - reported as kernel_ms_first_pair_incl_upload_wait)
    ctx.sync()
    torch.cuda.synchronize()
    ka, kb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.run()
    ka.record()
    for _ in range(3):
        ctx.run()
    kb.record()
    kb.synchronize()
    ctx.check()
    kern_ms = ka.elapsed_time(kb) / 3.0
    from sea_ice_drift_amd.dist import per_rank_breakdown
    per_rank = per_rank_breakdown([timed_steps.local_elapsed / args.steps * 1e3, kern_first_ms, len(mine)], dev) if world > 1 else None
    line = None
    # parity: up to three pairs of this rank against the oracle on a subsample (checker only)
    ok, checked = True, 0
    if args.check > 0 and mine:
        from oracle import c_oracle
        c_oracle.build()
        sel = np.random.default_rng(1).choice(n_pts, size=min(args.check, n_pts), replace=False)
        for p in mine[:3]:
            a, b = pairs[p]
            exp, exp_ij = c_oracle.pm_batch(a.numpy(), b.numpy(), g['c1'][sel], g['r1'][sel], g['c2fg'][sel],
                                            g['r2fg'][sel], g['border'][sel], s, 0.0, angles, rot=rot, nthreads=host_cores())
            got, got_ij = results[p]
            ok = ok and (np.array_equal(got_ij[sel], exp_ij) and np.array_equal(got[sel, :4], exp[:, :4], equal_nan=True)
                         and np.allclose(got[sel, 4], exp[:, 4], rtol=1e-5, atol=1e-5, equal_nan=True))
            checked += 1
    if world > 1:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    ctx.close()
    if rank == 0:
        total_pts = n_pts * args.pairs
        line = {
            'metric': 'PM grid-points/sec (10000x10000 px pair, 34px template)',
            'value': total_pts / (elapsed / args.steps), 'unit': 'grid-points/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': 'batch of %d synthetic %dx%d uint8 pairs streamed from pinned host memory, %dx%d grid '
                                   'per pair, template %d px, %d angles, border %s; uploads overlap the kernels of the '
                                   'previous pair (two device slots per GPU)' % (args.pairs, H, W, args.grid, args.grid, s,
                                                                                len(angles), border),
                       'pairs_total': args.pairs, 'pairs_per_gpu': len(mine), 'points_per_pair': int(n_pts),
                       'ms_per_pair': elapsed / args.steps * 1e3 / max(len(mine), 1),
                       'parallelism': 'pairs dealt round-robin to %d GPU(s); no collective on the data path' % world},
            'roofline': {'bound': 'mfma', 'achieved': 2.0 * info['macs'] / (kern_ms * 1e-3) / 1e12, 'peak': MFMA_I8_PEAK_TOPS,
                         'unit': 'TFLOP/s', 'frac': 2.0 * info['macs'] / (kern_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                         'traffic': None, 'kernel_ms_per_pair': kern_ms, 'kernel_ms_first_pair_incl_upload_wait': kern_first_ms,
                         'note': 'kernels of one pair in steady state (HIP events on the launch stream around 3 runs on a resident pair after the '
                                 'timed steps: nothing to wait for); the step time also holds the 200 MB H2D upload of every pair, overlapped - '
                                 'ms_per_pair above is the streaming rate, bound by the uploads when they take longer than the kernels'},
            'parity_check': {'pairs_checked_per_rank': checked, 'ok': ok},
            'setup_s': {'generate_and_pin_pairs': t_gen},
        }
        if per_rank is not None:                               # every rank's own clock: the slowest one is named
            line['step_breakdown_ms'] = {'per_rank': {'batch_ms': [float(v) for v in per_rank[:, 0]],
                                                      'kernel_ms_first_pair': [float(v) for v in per_rank[:, 1]],
                                                      'pairs': [int(v) for v in per_rank[:, 2]]},
                                         'slowest_rank': int(per_rank[:, 0].argmax()), 'slowest_batch_ms': float(per_rank[:, 0].max()),
                                         'fastest_batch_ms': float(per_rank[:, 0].min())}
        if not ok:
            print(json.dumps(line))
            raise SystemExit('PARITY FAILURE against the CPU oracle - the number above is invalid')
    return line


def ftpm_mode(args, torch, dist, dev, world, rank, local_rank):
    """BASELINE config 4: feature tracking feeding pattern matching through the public class (one GPU)."""
    import numpy as np
    from sea_ice_drift_amd import pmlib as my, synthetic as syn
    from sea_ice_drift_amd.domain import ArrayNansat
    from sea_ice_drift_amd.seaicedrift import SeaIceDrift
    if world != 1:
        raise SystemExit('--mode ftpm is the one-GPU configuration (BASELINE.json configs[3])')
    size, grid, s = args.size, args.grid, args.img_size
    angles = list(range(-args.angles, args.angles + 1))
    t_gen = time.time()
    img1, img2 = syn.make_pair(size, size, speckle=0.03)
    t_gen = time.time() - t_gen
    scale = 4e-4
    n1 = ArrayNansat(img1, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
    n2 = ArrayNansat(img2, origin=(10.0, 80.0), matrix=((scale, 0.0), (0.0, -scale)))
    cg, rg = np.meshgrid(np.rint(np.linspace(100, size - 101, grid)), np.rint(np.linspace(100, size - 101, grid)))
    lon, lat = n1.transform_points(cg.ravel(), rg.ravel(), 0)
    lon, lat = lon.reshape(cg.shape), lat.reshape(cg.shape)
    sid = SeaIceDrift(n1, n2)
    ft_kw = dict(max_drift=3000.0 * size / 10000.0 + 600.0, nFeatures=100000)
    pm_kw = dict(img_size=s, angles=angles)
    t_ft, t_pm = [], []
    for k in range(args.warmup + args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        uft, vft, lon1ft, lat1ft, lon2ft, lat2ft = sid.get_drift_FT(**ft_kw)
        t1 = time.perf_counter()
        u, v, a, r, h, lon2, lat2 = sid.get_drift_PM(lon, lat, lon1ft, lat1ft, lon2ft, lat2ft, **pm_kw)
        t2 = time.perf_counter()
        if k >= args.warmup:
            t_ft.append(t1 - t0)
            t_pm.append(t2 - t1)
    ft_s, pm_s = float(np.mean(t_ft)), float(np.mean(t_pm))
    ok = np.isfinite(u)
    # feature-tracking vectors against the synthetic displacement field
    x1, y1 = n1.transform_points(lon1ft, lat1ft, 1)
    x2, y2 = n2.transform_points(lon2ft, lat2ft, 1)
    fdc, fdr = syn.true_displacement(x1, y1)
    fterr = np.hypot(x2 - x1 - fdc, y2 - y1 - fdr)
    tdc, tdr = syn.true_displacement(cg, rg)
    err = np.hypot(u[ok] / scale - tdc[ok], -v[ok] / scale - tdr[ok])
    line = {
        'metric': 'PM grid-points/sec (10000x10000 px pair, 34px template)', 'value': float(ok.sum()) / (ft_s + pm_s),
        'unit': 'grid-points/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': (ft_s + pm_s) * 1e3,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': 'FT+PM end to end through SeaIceDrift.get_drift_FT -> get_drift_PM: %dx%d synthetic uint8 pair (host '
                               'arrays in, host grids out), %dx%d grid, template %d px, %d angles; first guess and search border from '
                               'the feature-tracking vectors' % (size, size, grid, grid, s, len(angles)),
                   'parallelism': 'single GPU, no collective'},
        'ft_ms': ft_s * 1e3, 'pm_ms': pm_s * 1e3, 'ft_vectors': int(len(uft)),
        'ft_vectors_within_3px_of_truth': float((fterr < 3.0).mean()) if len(fterr) else 0.0,
        'valid_grid_points': int(ok.sum()), 'median_abs_drift_error_px': float(np.median(err)) if ok.any() else None,
        'detector': 'sea_ice_drift_amd.orb (own ORB-family specification behind the reference interface; OpenCV parity unpinned)',
        'setup_s': {'generate_pair': t_gen},
    }
    if args.check > 0:
        # PM on the FT-derived first guess against the C oracle: the same prelude (first guess, search border, validity
        # mask) feeds both; compared are a, r (bit-exact), h (1e-5) and the destination lon/lat (bit-exact)
        from oracle import c_oracle
        c_oracle.build()
        xk1, yk1 = n1.transform_points(lon1ft, lat1ft, 1)
        xk2, yk2 = n2.transform_points(lon2ft, lat2ft, 1)
        pre = my.pm_prelude(lon, lat, n1, xk1, yk1, n2, xk2, yk2, **pm_kw)
        gpi = pre['gpi']
        pos = np.flatnonzero(gpi.ravel())
        sel = np.sort(np.random.default_rng(3).choice(pos.size, size=min(args.check, pos.size), replace=False))
        exp, _ = c_oracle.pm_batch(img1, img2, pre['c1pm1i'][gpi][sel], pre['r1pm1i'][gpi][sel], pre['c2fg'][gpi][sel],
                                   pre['r2fg'][gpi][sel], pre['brd2'][gpi][sel], s, pre['alpha0'], angles,
                                   rot=my.rotation_table(angles, pre['alpha0'], s), nthreads=host_cores())
        flat = pos[sel]
        dci, dri = (pre['c2pm1'] - pre['c2pm1i'])[gpi][sel], (pre['r2pm1'] - pre['r2pm1i'])[gpi][sel]
        elon, elat = n2.transform_points(exp[:, 0] + dci, exp[:, 1] + dri, 0)
        same = lambda got, want: bool(np.all((got == want) | (np.isnan(got) & np.isnan(want))))
        okp = (same(a.ravel()[flat], exp[:, 2]) and same(r.ravel()[flat], exp[:, 3]) and same(lon2.ravel()[flat], elon)
               and same(lat2.ravel()[flat], elat) and bool(np.allclose(h.ravel()[flat], exp[:, 4], rtol=1e-5, atol=1e-5, equal_nan=True)))
        line['parity_check'] = {'points': int(len(sel)), 'ok': okp, 'valid_points_total': int(gpi.sum()),
                                'rule': 'a, r, lon2, lat2 bit-exact, h to 1e-5, against oracle/pm_oracle.c on the FT-derived first guess'}
        if not okp:
            print(json.dumps(line))
            raise SystemExit('PARITY FAILURE against the CPU oracle - the number above is invalid')
    return line


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'RANK' not in os.environ:
        raise SystemExit(spawn_ranks(args, argv))        # before any torch / GPU call in this process
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    # stdout carries ONE JSON line: whatever libraries print there - gloo's connection lines at init_process_group, RCCL's
    # version banner (flushed when the process ends), progress lines of host code - goes to stderr for the whole life of
    # the process; the line itself is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)')
    # N ranks on fewer devices (functional dry runs on a 1-GPU box) share devices round-robin
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    if world > 1 or args.force_collective:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:                                     # --force-collective: a one-rank RCCL group, no launcher
            os.environ.setdefault('MASTER_PORT', str(free_port()))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        shared = torch.cuda.device_count() < world
        # RCCL cannot put two ranks on one device: the dry run on a smaller box uses gloo for the collectives
        dist.init_process_group('gloo' if shared else 'nccl', **({} if shared else {'device_id': dev}))
    fn = {'stream': stream_mode, 'ftpm': ftpm_mode}.get(args.mode, grid_mode)
    line = fn(args, torch, dist, dev, world, rank, local_dev)
    if rank == 0 and line is not None:
        if world > 1 and dist.get_backend() == 'gloo':
            line['config']['parallelism'] += ' [DRY RUN: %d ranks share %d device(s), gloo collectives]' % (
                world, torch.cuda.device_count())
        os.write(real_stdout, (json.dumps(line) + '\n').encode())
    if world > 1 or args.force_collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
