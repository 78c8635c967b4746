#!/usr/bin/env python3
"""Per-kernel memory-side traffic from two rocprofv3 --pmc runs (FETCH_SIZE in one, WRITE_SIZE in the other).

    python tools/pmc_traffic.py fetch.db write.db

FETCH_SIZE and WRITE_SIZE are reported in kilobytes. On gfx950 FETCH_SIZE is the L2's memory-side read REQUESTS x 64
bytes, whatever their size: a fully coalesced 16-byte-per-lane stream is made of 128-byte requests and reads exactly
half of its bytes (reverse step at 2^26 elements: 402.7 MB reported for 805.3 MB read), row-strided loads are made of
64-byte requests and read about right (1x1 conv with resident weights: 505.7 MB for a 537 MB input), the halo conv's
64-byte pieces of pixel rows lie in between (424.9 MB for a 537 MB input that it must read at least once: x 1.27 at
least). tools/pmc_calibrate.sh measures these three on tensors larger than the on-die cache; the factors below come
from it, every other kernel keeps the x 2 of coalesced streams (an upper bound for strided access). Writes are exact
for 16-byte-per-lane stores and dword atomics.
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


READ_FACTOR = (("conv3x3_halo_kernel", 1.27), ("conv1x1_resident_kernel", 1.06))


def read_factor(kernel):
    return next((f for key, f in READ_FACTOR if key in kernel), 2.0)


def main(fetch_db, write_db):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        fk, fn = f.get(k, (0.0, 0))
        wk, wn = w.get(k, (0.0, 0))
        n = max(fn, wn, 1)
        rd, wr = read_factor(k) * fk * 1024 / n, wk * 1024 / n
        rows.append((rd + wr, k, n, rd, wr, read_factor(k)))
    print("# per dispatch: bytes read (FETCH_SIZE KB x 1024 x the kernel's calibrated factor, column 'x'), bytes written (WRITE_SIZE KB x 1024)")
    print(f"{'launches':>8} {'read_MB':>10} {'write_MB':>10} {'total_MB':>10} {'x':>5}  kernel")
    for tot, k, n, rd, wr, fac in sorted(rows, reverse=True):
        print(f"{n:8d} {rd / 1e6:10.2f} {wr / 1e6:10.2f} {tot / 1e6:10.2f} {fac:5.2f}  {k[:120]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
