#!/usr/bin/env python3
"""Per-kernel memory-side traffic from two rocprofv3 --pmc runs (FETCH_SIZE in one, WRITE_SIZE in the other).

    python tools/pmc_traffic.py fetch.db write.db

FETCH_SIZE and WRITE_SIZE are reported in kilobytes. On gfx950 FETCH_SIZE counts 128-byte read requests as 64
bytes for wide coalesced loads, so reads are doubled; writes are exact for 16-byte-per-lane stores and dword atomics.
"""
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('pmc_events')")]
    idx = {c: i for i, c in enumerate(cols)}
    namecol = "name" if "name" in idx else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in idx else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in idx else [c for c in cols if "value" in c][0]
    tot, n = defaultdict(float), defaultdict(set)
    for r in cur.execute("select * from pmc_events"):
        if r[idx[ccol]] != counter:
            continue
        tot[r[idx[namecol]]] += float(r[idx[vcol]])
        n[r[idx[namecol]]].add(r[idx["dispatch_id"]])
    return {k: (tot[k], len(n[k])) for k in tot}


def main(fetch_db, write_db):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        fk, fn = f.get(k, (0.0, 0))
        wk, wn = w.get(k, (0.0, 0))
        n = max(fn, wn, 1)
        rd, wr = 2.0 * fk * 1024 / n, wk * 1024 / n
        rows.append((rd + wr, k, n, rd, wr))
    print("# per dispatch: bytes read (FETCH_SIZE KB x 1024 x 2, the gfx950 correction), bytes written (WRITE_SIZE KB x 1024)")
    print(f"{'launches':>8} {'read_MB':>10} {'write_MB':>10} {'total_MB':>10}  kernel")
    for tot, k, n, rd, wr in sorted(rows, reverse=True):
        print(f"{n:8d} {rd / 1e6:10.2f} {wr / 1e6:10.2f} {tot / 1e6:10.2f}  {k[:120]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
