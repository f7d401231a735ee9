#!/usr/bin/env python3
"""Sum rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (rocpd .db, KB units) per kernel.

    python tools/pmc_traffic.py fetch.db write.db [kernel substring] [steps]
"""
import json, sqlite3, sys


def total(path, counter, kern):
    db = sqlite3.connect(path)
    rows = list(db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                           "and kernel_name like ? group by kernel_name", (counter, '%' + kern + '%')))
    return rows


if __name__ == '__main__':
    fdb, wdb = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else ''
    out = {}
    for name, n, v in total(fdb, 'FETCH_SIZE', kern): out.setdefault(name, {}).update(dispatches=n, fetch_kb=v)
    for name, n, v in total(wdb, 'WRITE_SIZE', kern): out.setdefault(name, {}).update(dispatches=n, write_kb=v)
    print(json.dumps(out, indent=1))
