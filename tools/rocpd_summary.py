#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) result: kernel stats (as --stats prints them), the
per-dispatch rows of the PM kernel, and PMC counter values per dispatch.

    python tools/rocpd_summary.py gpurun_out/prof/kt/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, kernel_filter='pm_kernel'):
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


if __name__ == '__main__':
    main(*sys.argv[1:])
