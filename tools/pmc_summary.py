#!/usr/bin/env python3
"""Sums the hardware counters of a rocprofv3 --pmc run (rocpd SQLite) per kernel name."""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('pmc_events')")]
    rows = list(cur.execute("select * from pmc_events"))
    idx = {c: i for i, c in enumerate(cols)}
    namecol = "name" if "name" in idx else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in idx else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in idx else [c for c in cols if "value" in c][0]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    dur = defaultdict(dict)
    for r in rows:
        agg[r[idx[namecol]]][r[idx[ccol]]] += float(r[idx[vcol]])
        if "dispatch_id" in idx:
            cnt[r[idx[namecol]]].add(r[idx["dispatch_id"]])
            if "duration" in idx:
                dur[r[idx[namecol]]][r[idx["dispatch_id"]]] = float(r[idx["duration"]])
    for k, d in agg.items():
        n = max(1, len(cnt[k]))
        avg_us = sum(dur[k].values()) / max(1, len(dur[k])) / 1e3
        print(f"{k[:110]}  ({n} dispatches, {avg_us:.1f} us each under the profiler)")
        for c, v in sorted(d.items()):
            print(f"    {c:32s} {v/n:16.1f} per dispatch")


if __name__ == "__main__":
    main(sys.argv[1])
