#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) result: kernel stats (as --stats prints them), the
per-dispatch rows of the PM kernel, and PMC counter values per dispatch.

    python tools/rocpd_summary.py gpurun_out/prof/kt/r01_results.db > profiles/r01_kernel_stats.txt
    python tools/rocpd_summary.py results.db pm_kernel --window WARMUP STEPS     # + stats over the TIMED steps only

--window W S: the dispatches of the filtered kernel are taken as steps of equal length (their number / (W + S + 1): warm-up steps,
timed steps, the one poisoned step the parity check reads); the stats of the S timed steps are printed as well - per launch
position of a step and their sum, which is what bench.py's `kernel_ms_per_step` (HIP events over the same steps) must agree with.
The all-dispatch table above it holds the warm-up (clocks still ramping) and cannot be summed to a step.
"""
import sqlite3
import sys


def window_stats(db, kernel_filter, warmup, steps):
    try:
        rows = list(db.execute("select name, duration, start from kernels where name like ? order by start", ('%' + kernel_filter + '%',)))
    except sqlite3.Error:                                          # (a rocpd schema without `start`: dispatch order, no spans)
        rows = [(r[0], r[1], None) for r in db.execute("select name, duration from kernels where name like ? order by dispatch_id", ('%' + kernel_filter + '%',))]
    total_steps = warmup + steps + 1
    if not rows or len(rows) % total_steps:
        print('\n# TIMED WINDOW: %d dispatches are not %d equal steps - skipped' % (len(rows), total_steps))
        return
    per = len(rows) // total_steps
    win = rows[warmup * per:(warmup + steps) * per]
    print('\n# TIMED WINDOW: steps %d .. %d of %d (%d launches per step; warm-up and the poisoned step left out); durations in ns' %
          (warmup, warmup + steps - 1, total_steps, per))
    print('%-8s %-70s %8s %12s %12s %12s' % ('launch', 'name', 'calls', 'avg_ns', 'min_ns', 'max_ns'))
    tot = 0.0
    for k in range(per):
        d = [r[1] for r in win[k::per]]
        tot += sum(d) / len(d)
        print('%-8d %-70s %8d %12.0f %12d %12d' % (k, win[k][0][:70], len(d), sum(d) / len(d), min(d), max(d)))
    print('# sum of the average launch durations of a step: %.1f ns = %.4f ms' % (tot, tot * 1e-6))
    if win[0][2] is None:
        return
    span = [(win[(i + 1) * per - 1][2] + win[(i + 1) * per - 1][1]) - win[i * per][2] for i in range(steps)]
    print('# first launch start -> last launch end of a step, average: %.4f ms (min %.4f, max %.4f)' %
          (sum(span) / len(span) * 1e-6, min(span) * 1e-6, max(span) * 1e-6))


def main(path, kernel_filter='pm_kernel', *rest):
    db = sqlite3.connect(path)
    print('# source: %s' % path)
    print('# KERNEL STATS (rocprofv3 --kernel-trace --stats; durations in ns)')
    print('%-70s %8s %14s %12s %8s' % ('name', 'calls', 'total_ns', 'avg_ns', 'pct'))
    for name, calls, total, avg, pct in db.execute('select name,total_calls,total_duration*1000,average*1000,'
                                                   'percentage from top_kernels'):
        print('%-70s %8d %14.0f %12.0f %8.3f' % (name[:70], calls, total, avg, pct))
    rows = list(db.execute("select dispatch_id, grid_x * max(grid_y, 1) * max(grid_z, 1), workgroup_x, lds_size, vgpr_count, sgpr_count, "
                           "scratch_size, duration from kernels where name like ? order by dispatch_id",
                           ('%' + kernel_filter + '%',)))
    if rows:
        print('\n# DISPATCHES of *%s* (grid = threads; duration ns)' % kernel_filter)
        print('%8s %10s %6s %8s %6s %6s %8s %12s' % ('dispatch', 'grid', 'wg', 'lds', 'vgpr', 'sgpr', 'scratch',
                                                     'duration'))
        for r in rows:
            print('%8d %10d %6d %8d %6d %6d %8d %12d' % r)
    pmc = list(db.execute("select counter_name, dispatch_id, grid_size, lds_block_size, value, duration "
                          "from counters_collection where kernel_name like ? order by counter_name, dispatch_id",
                          ('%' + kernel_filter + '%',)))
    if pmc:
        print('\n# PMC per dispatch of *%s*' % kernel_filter)
        print('%-16s %8s %10s %8s %16s %12s' % ('counter', 'dispatch', 'grid', 'lds', 'value', 'duration_ns'))
        for r in pmc:
            print('%-16s %8d %10d %8d %16.3f %12d' % r)
    if len(rest) >= 3 and rest[0] == '--window':
        window_stats(db, kernel_filter, int(rest[1]), int(rest[2]))


if __name__ == '__main__':
    main(*sys.argv[1:])
